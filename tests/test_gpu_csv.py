"""GPU CSV ingest: nqe_csv_infer_schema + nqe_csv_read through the C ABI against the oracle's
sequential restatement of CsvTable::try_create (csv.rs:53-86) and the reference's own fixtures/tests
(csv.rs:115-170: test_infer_schema, test_read_from_csv)."""
import os

import numpy as np
import pytest

from naive_query_engine_amd import DType, ErrorCode, Status
from oracle import oracle as orc
from tests.helpers import assert_column_equal

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from naive_query_engine_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def read_both(ctx, data: bytes, **kw):
    names, nullable, exp = orc.csv_read(data, **kw)
    gn, gd, gnull = ctx.csv_infer_schema(data, kw.get("has_header", True), kw.get("delimiter", ","), kw.get("max_read_records", 3), kw.get("batch_size", 1_000_000))
    assert gn == names and gnull == nullable and gd == [c.dtype for c in exp]
    t = ctx.csv_read(data, gd, kw.get("has_header", True), kw.get("delimiter", ","), kw.get("batch_size", 1_000_000))
    got = t.to_host()
    assert len(got) == len(exp)
    for i, (g, e) in enumerate(zip(got, exp)):
        assert_column_equal(g, e, what=f"column {names[i]}")
    return names, got


def test_reference_fixtures_and_csv_rs_tests(ctx, csv_tables):
    """csv.rs test_infer_schema / test_read_from_csv on data/test_data.csv, plus the other README tables"""
    data = open(os.path.join(GOLDEN, "test_data.csv"), "rb").read()
    names, got = read_both(ctx, data)
    assert names == ["id", "name", "age", "score"]
    assert [c.dtype for c in got] == [DType.INT64, DType.UTF8, DType.INT64, DType.FLOAT64]
    assert got[0].to_list() == [1, 2, 4, 5, 6, 7, 8, 9]                       # csv.rs:150-153
    assert got[1].to_list()[:3] == ["veeupup", "alex", "lynne"]
    assert got[3].to_list()[:3] == [60.0, 90.1, 99.99]
    for name in ("employee", "rank", "department"):
        data = open(os.path.join(GOLDEN, f"{name}.csv"), "rb").read()
        names, got = read_both(ctx, data)
        ref = csv_tables[name]
        assert names == [f.name for f in ref.fields]
        for g, e in zip(got, ref.columns):
            assert_column_equal(g, e, what=name)


def test_quoting_terminators_and_empty_lines(ctx):
    cases = [
        b'a,b,c\r\n1,"x, ""y""",2.5\r\n\r\n-7,"multi\nline",\n3,plain"q,1e3\n4,"ab"cd"e",.5',
        b'a,b\n1,2\n3,4',                                  # no trailing terminator
        b'a,b\n1,2\n3,4\n\n\n',                            # trailing empty lines
        b'a\r1\r2\r3\r',                                   # bare CR terminators
        b'x,y\n"",1\n"""",2\n"a""b""",3\n',                # empty quoted, lone escaped quote
        b'k,v\n1,\n2,\n3,5\n',                             # trailing empty numeric field -> NULL
        b'k;v\n1;a,b\n2;"c;d"\n',                          # other delimiter
        b'h1,h2,h3\n,,\n,,\n',                             # nothing to infer from: Utf8, nullable
    ]
    for i, data in enumerate(cases):
        kw = {"max_read_records": -1}
        if i == 6:
            kw["delimiter"] = ";"
        read_both(ctx, data, **kw)
    read_both(ctx, b'1,2.5,x\n3,4.5,y\n', has_header=False)   # column_1.. names


def test_numbers_inference_and_nulls(ctx):
    rows = ["i,f,mixed,b,big", "1,1.5,1,true,9223372036854775807", "-2,0.25,2.5,FALSE,-9223372036854775808", "3,100.0,3,True,0",
            ",,,,", "4,1e3,7,false,12", "5,.5,8,true,13", "6,-inf,9,false,14", "7,NaN,10,true,15",
            "8,4.9e-324,11,true,16", "9,1.7976931348623159e308,12,true,17", "10,0.1000000000000000055511151231257827021181583404541015625,13,false,18",
            "11,9007199254740993,14,true,19", "+12,+123456789012345678901234567890.5,15,false,+20"]   # a leading + parses (lexical) but is not inferred
    data = ("\n".join(rows) + "\n").encode()
    names, got = read_both(ctx, data)
    assert [c.dtype for c in got] == [DType.INT64, DType.FLOAT64, DType.FLOAT64, DType.BOOLEAN, DType.INT64]
    assert got[0].to_list()[3] is None and got[3].to_list()[3] is None
    # batch_size keeps only the first batch (quirk Q1); max_read_records decides the types
    read_both(ctx, data, batch_size=5)
    read_both(ctx, data, batch_size=0)
    read_both(ctx, b"a,b\n1,x\n2,y\n3,z\n4.5,w\n", max_read_records=-1)  # all rows sampled: Int64 + Float64 -> Float64

def test_errors(ctx):
    def both_fail(data, **kw):
        with pytest.raises(ErrorCode) as a:
            orc.csv_read(data, **kw)
        with pytest.raises(ErrorCode) as b:
            names, dts, _ = ctx.csv_infer_schema(data, kw.get("has_header", True), ",", kw.get("max_read_records", 3))
            ctx.csv_read(data, dts, kw.get("has_header", True))
        assert a.value.status == b.value.status, data
        return a.value.status

    assert both_fail(b"a,b\n1,2\n3,4\n5,6\n7.5,8\n") == Status.ArrowError            # Int64 column, later row is a float
    assert both_fail(b"a,b\n1,2\n3,4\n5,6\n99999999999999999999,8\n") == Status.ArrowError   # overflow
    assert both_fail(b"a,b\n1,2\n3,4\n5,6\n7\n") == Status.ArrowError                 # too few fields
    assert both_fail(b"a,b\n1,2\n3,4\n5,6\n7,8,9\n") == Status.ArrowError             # too many fields
    assert both_fail(b"a,b\ntrue,1\nfalse,2\nmaybe,3\n", max_read_records=2) == Status.ArrowError
    assert both_fail(b"") == Status.ArrowError
    assert both_fail(b"d\n2020-01-01\n2020-01-02\n") == Status.NotSupported


def test_large_random_file_matches_oracle(ctx):
    """100k rows, every feature mixed; record boundaries cross the 64-byte chunks and 16 KB workgroup tiles everywhere"""
    rng = np.random.default_rng(99)
    n = 100_000
    words = ["", "a", "bob", 'say "hi"', "x,y", "line\nbreak", "crlf\r\nin", "日本語", "trailing\"quote", "p" * 90]
    lines = ["id,price,qty,name,flag,note"]
    for i in range(n):
        price = "" if rng.random() < 0.03 else repr(float(rng.random() * 1e6 - 5e5)) if rng.random() < 0.7 else f"{rng.integers(-10**9, 10**9)}e{rng.integers(-30, 30)}"
        qty = "" if rng.random() < 0.02 else str(int(rng.integers(-10**12, 10**12)))
        def q(s):
            s = str(s)
            return '"' + s.replace('"', '""') + '"' if (any(ch in s for ch in ',"\r\n') or rng.random() < 0.1) else s
        name = q(words[int(rng.integers(0, len(words)))] + (str(i % 97) if i % 5 else ""))
        flag = ["true", "false", "TRUE", "False", ""][int(rng.integers(0, 5))]
        note = q(words[int(rng.integers(0, len(words)))])
        lines.append(f"{i},{price},{qty},{name},{flag},{note}")
        if rng.random() < 0.01:
            lines.append("")   # empty line
    term = ["\n", "\r\n"]
    data = "".join(l + term[int(rng.integers(0, 2))] for l in lines).encode()
    names, got = read_both(ctx, data)
    assert got[0].length == n and got[0].to_list()[:3] == [0, 1, 2]
    # the same image already resident in HBM (staged through a 1-column UInt64 table: no other H2D copy in the ABI)
    from naive_query_engine_amd import Column

    padded = data + b"\0" * ((-len(data)) % 8)
    holder = ctx.table_from_host([Column.from_numpy(np.frombuffer(padded, dtype=np.uint64).copy())])
    _, dts, _ = ctx.csv_infer_schema(data)
    t = ctx.csv_read(None, dts, device_ptr=int(holder.column_info(0).values), nbytes=len(data))
    _, _, exp = orc.csv_read(data)
    for g, e in zip(t.to_host(), exp):
        assert_column_equal(g, e, what="device-resident image")


def test_csv_table_plan_mirror(ctx):
    from naive_query_engine_amd import physical_plan as pp

    t = pp.CsvTable.try_create(os.path.join(GOLDEN, "test_data.csv"), pp.CsvConfig(), ctx)
    assert [f.name for f in t.schema()] == ["id", "name", "age", "score"]
    assert [f.nullable for f in t.schema()] == [False] * 4                      # csv.rs:129-131
    b = pp.ScanPlan.create(t, None).execute()
    assert len(b) == 1 and b[0].num_rows == 8
    assert t.source_name() == "CsvTable"


def test_device_number_parser_vs_strtod():
    """the same corpus as tests/test_csv_host.py, parsed by GPU threads (the first version of the digit scanner was
    mis-compiled for the device only, which this catches)"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_csv_parse_device")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", os.path.join(root, "tests", "cpp", "test_csv_parse_device.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert " 0 failures" in out.stdout
