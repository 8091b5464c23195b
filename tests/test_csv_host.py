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
