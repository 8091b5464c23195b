"""CPU: the CSV number parser (csrc/csv_parse.hpp, the code the device kernel runs) against glibc strtod/strtoll on
1.4 M random, halfway and edge-case strings; the oracle's sequential CSV reader against the reference's fixtures."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_number_parser_matches_strtod_on_host():
    exe = os.path.join(ROOT, "tests", "cpp", "test_csv_parse")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_csv_parse.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 failures" in out.stdout


def test_oracle_csv_reader_on_reference_fixtures(csv_tables):
    """csv.rs:115-170 (test_infer_schema, test_read_from_csv) through the oracle's restatement; all four README tables
    must equal the independently written Python reader used for the golden vectors"""
    from naive_query_engine_amd import DType
    from oracle import oracle as orc
    from tests.helpers import assert_column_equal

    for name, ref in csv_tables.items():
        data = open(os.path.join(ROOT, "tests", "golden", f"{name}.csv"), "rb").read()
        names, nullable, cols = orc.csv_read(data)
        assert names == [f.name for f in ref.fields]
        assert nullable == [False] * len(names)
        for g, e in zip(cols, ref.columns):
            assert_column_equal(g, e, what=name)
    names, _, cols = orc.csv_read(open(os.path.join(ROOT, "tests", "golden", "test_data.csv"), "rb").read())
    assert [c.dtype for c in cols] == [DType.INT64, DType.UTF8, DType.INT64, DType.FLOAT64]  # csv.rs:120-125
    assert cols[0].to_list() == [1, 2, 4, 5, 6, 7, 8, 9]


def test_oracle_csv_reader_edge_semantics():
    from naive_query_engine_amd import ErrorCode
    from oracle import oracle as orc

    names, nullable, cols = orc.csv_read(b'a,b,c\r\n1,"x, ""y""",2.5\r\n\r\n-7,"multi\nline",\n3,plain"q,1e3\n4,"ab"cd"e",.5', max_read_records=-1)
    assert names == ["a", "b", "c"] and nullable == [False, False, True]
    assert cols[0].to_list() == [1, -7, 3, 4]
    assert cols[1].to_list() == ['x, "y"', "multi\nline", 'plain"q', 'abcd"e"']
    assert cols[2].to_list() == ["2.5", "", "1e3", ".5"]          # 1e3 / .5 do not match arrow 13's DECIMAL_RE: mixed -> Utf8
    _, _, cols = orc.csv_read(b"k,v\n1,\n2,\n3,5\n", max_read_records=-1)
    assert cols[1].to_list() == [None, None, 5]
    _, _, cols = orc.csv_read(b"a\n1\n2\n3\n4\n", batch_size=2)
    assert cols[0].to_list() == [1, 2]                              # quirk Q1: first batch only
    with pytest.raises(ErrorCode):
        orc.csv_read(b"a,b\n1,2\n3\n")
    with pytest.raises(ErrorCode):
        orc.csv_read(b"a\n1\n2\n3\n4.5\n")                          # Int64 inferred from 3 rows, row 4 is not an integer


def test_c_abi_schema_inference_without_gpu(csv_tables):
    """nqe_csv_infer_schema is host-side code of the product library: it runs here (no GPU, no context) and must agree
    with the oracle's independent inference on the reference's fixtures and on quoting / mixed-type corner cases"""
    import ctypes as C

    from naive_query_engine_amd import DType, capi
    from naive_query_engine_amd.arrow_host import NqeCsvOptions
    from oracle import oracle as orc

    L = capi.lib()

    def infer(data, has_header=True, delimiter=",", max_read_records=3):
        opt = NqeCsvOptions(int(has_header), ord(delimiter), max_read_records, 1_000_000)
        nc, need = C.c_int32(), C.c_int64()
        dts, nul = (C.c_int32 * 64)(), (C.c_int32 * 64)()
        names = C.create_string_buffer(4096)
        st = L.nqe_csv_infer_schema(None, data, len(data), C.byref(opt), 64, C.byref(nc), dts, nul, names, len(names), C.byref(need))
        if st != 0:
            return st
        return (names.raw[: need.value].split(b"\0")[: nc.value], [DType(dts[i]) for i in range(nc.value)], [bool(nul[i]) for i in range(nc.value)])

    cases = [open(os.path.join(ROOT, "tests", "golden", f"{n}.csv"), "rb").read() for n in csv_tables]
    cases += [b'a,b,c\r\n1,"x, ""y""",2.5\r\n\r\n-7,"multi\nline",\n3,plain"q,1e3\n', b"k;v\n1;a,b\n2;\"c;d\"\n", b"h1,h2\n,\n,\n", b"x,y\ntrue,1\nFALSE,2.5\n", b"1,2.5,x\n3,4.5,y\n"]
    for i, data in enumerate(cases):
        kw = dict(has_header=i != len(cases) - 1, delimiter=";" if i == 5 else ",", max_read_records=-1 if i >= 4 else 3)
        names, nullable, cols = orc.csv_read(data, **kw)
        got = infer(data, **kw)
        assert got == ([n.encode() for n in names], [c.dtype for c in cols], nullable), (i, got)
    assert infer(b"") == 1                                   # ArrowError: empty file
    assert infer(b"d\n2020-01-01\n2020-01-02\n") == 11       # NotSupported: Date32 column
