"""GPU: the Arrow C Data Interface at the C ABI (nqe_table_import_arrow / nqe_table_export_arrow) through
pyarrow's own `_export_to_c` / `_import_from_c`: real RecordBatches go to HBM and results come back as RecordBatches that own
their buffers through the interface's release callbacks."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
pc = pytest.importorskip("pyarrow.compute")

from naive_query_engine_amd import AggregateFunc, ErrorCode, Operator, Status  # noqa: E402
from naive_query_engine_amd.expression import binop, col, lit_i64  # noqa: E402
from tests.helpers import fields  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def make_batch(n, seed=0, null_frac=0.2):
    rng = np.random.default_rng(seed)

    def mask():
        return rng.random(n) < null_frac

    words = ["", "a", "bob", "véé", "日本語", "x" * 40]
    return pa.RecordBatch.from_arrays(
        [pa.array(rng.integers(-2**62, 2**62, n), pa.int64(), mask=mask()),
         pa.array(rng.integers(0, 2**63, n).astype(np.uint64), pa.uint64(), mask=mask()),
         pa.array(rng.normal(size=n), pa.float64(), mask=mask()),
         pa.array(rng.random(n) < 0.5, pa.bool_(), mask=mask()),
         pa.array([words[int(i)] for i in rng.integers(0, len(words), n)], pa.utf8(), mask=mask()),
         pa.array(np.arange(n, dtype=np.int64))],  # no nulls: no validity buffer
        names=["i", "u", "f", "b", "s", "id"])


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1000])
def test_round_trip_all_types_with_nulls(ctx, n):
    rb = make_batch(n, seed=n)
    t = ctx.table_from_arrow(rb)
    assert t.num_rows == n and t.num_columns == 6
    back = t.to_arrow(names=rb.schema.names)
    assert back.schema.names == rb.schema.names
    assert back.equals(rb), (back.to_pydict(), rb.to_pydict())
    assert t.to_arrow().schema.names == [f"c{i}" for i in range(6)]


@pytest.mark.parametrize("off,ln", [(0, 10), (3, 50), (8, 64), (13, 900), (999, 1), (500, 0)])
def test_sliced_batches_honour_offsets(ctx, off, ln):
    """a slice carries offset != 0 on every child: bit offsets for validity / Boolean, element offsets for values and Utf8 offsets"""
    rb = make_batch(1000, seed=7).slice(off, ln)
    back = ctx.table_from_arrow(rb).to_arrow(names=rb.schema.names)
    assert back.equals(rb)


def test_operators_between_import_and_export(ctx):
    rb = make_batch(5000, seed=3, null_frac=0.1)
    t = ctx.table_from_arrow(rb)
    f = fields(*rb.schema.names)
    out = ctx.selection(t, binop(col(5), Operator.Lt, lit_i64(2500)).flatten(f)).to_arrow(names=rb.schema.names)
    assert out.equals(rb.filter(pc.less(rb.column(5), 2500)))
    agg = ctx.aggregate(t, [(AggregateFunc.Count, 2), (AggregateFunc.Sum, 2)], group_nodes=binop(col(5), Operator.Modulos, lit_i64(7)).flatten(f)).to_arrow(names=["n", "s"])
    exp = pa.table({"k": pc.subtract(rb.column(5), pc.multiply(pc.divide(rb.column(5), 7), 7)), "f": rb.column(2)}).group_by("k").aggregate([("f", "count"), ("f", "sum")]).sort_by("k")
    assert agg.column(0).to_pylist() == exp.column("f_count").to_pylist()
    assert np.allclose(agg.column(1).to_numpy(), exp.column("f_sum").to_numpy(), rtol=1e-9)


def test_unsupported_type_is_rejected_and_nothing_leaks(ctx):
    rb = pa.RecordBatch.from_arrays([pa.array([1, 2, 3], pa.int32())], names=["x"])
    with pytest.raises(ErrorCode) as e:
        ctx.table_from_arrow(rb)
    assert e.value.status == Status.NotSupported
    d = pa.RecordBatch.from_arrays([pa.array(["a", "b", "a"]).dictionary_encode()], names=["d"])
    with pytest.raises(ErrorCode):
        ctx.table_from_arrow(d)


def test_exported_batch_outlives_the_table(ctx):
    rb = make_batch(300, seed=9)
    t = ctx.table_from_arrow(rb)
    out = t.to_arrow(names=rb.schema.names)
    del t
    ctx.trim()
    assert out.equals(rb)  # host copies owned through the release callbacks
