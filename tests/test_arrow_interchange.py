"""Host containers <-> pyarrow: the buffers the C ABI takes are Arrow's own (CPU only)."""
import numpy as np
import pytest

from naive_query_engine_amd import Column, DType, Field, RecordBatch

pa = pytest.importorskip("pyarrow")


def test_from_arrow_and_back_preserves_values_and_nulls():
    t = pa.record_batch({
        "id": pa.array([1, None, 3, 4], type=pa.int64()),
        "u": pa.array([1, 2, 3, 2**64 - 1], type=pa.uint64()),
        "score": pa.array([1.5, 2.5, None, float("nan")], type=pa.float64()),
        "flag": pa.array([True, False, None, True], type=pa.bool_()),
        "name": pa.array(["vee", None, "", "日本語"], type=pa.string()),
    })
    b = RecordBatch.from_arrow(t)
    assert [f.dtype for f in b.fields] == [DType.INT64, DType.UINT64, DType.FLOAT64, DType.BOOLEAN, DType.UTF8]
    assert b.columns[0].to_list() == [1, None, 3, 4]
    assert b.columns[1].to_list() == [1, 2, 3, 2**64 - 1]
    assert b.columns[3].to_list() == [True, False, None, True]
    assert b.columns[4].to_list() == ["vee", None, "", "日本語"]
    back = b.to_arrow()
    assert back.schema.names == t.schema.names
    for a, e in zip(back.columns, t.columns):
        assert a.to_pylist()[:3] == e.to_pylist()[:3]
    # sliced input (non-zero offset) is normalised
    s = RecordBatch.from_arrow(t.slice(1, 2))
    assert s.columns[0].to_list() == [None, 3] and s.columns[4].to_list() == [None, ""]


def test_round_trip_of_host_batch():
    b = RecordBatch([Field("a", DType.INT64), Field("s", DType.UTF8)], [Column.from_numpy(np.arange(70, dtype=np.int64)), Column.from_list(["x"] * 69 + [None], DType.UTF8)])
    r = RecordBatch.from_arrow(b.to_arrow())
    assert r.columns[0].to_list() == list(range(70)) and r.columns[1].to_list() == ["x"] * 69 + [None]
