"""ctypes binding of the CPU oracle (oracle/nqe_oracle.cpp). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  A "batches" value here is the reference's
`Vec<RecordBatch>`: a list of batches, each a list of host `Column`s.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

from naive_query_engine_amd.arrow_host import (Column, ErrorCode, NqeAggregate, NqeColumn, NqeExprNode,
                                               nodes_array)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnqe_oracle.so")

Batches = List[List[Column]]


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (oracle/Makefile)."""
    src = os.path.join(_HERE, "nqe_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "nqe.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnqe_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.orc_last_error.restype = C.c_char_p
        L.orc_batches_new.restype = vp
        L.orc_batches_new.argtypes = [C.POINTER(C.c_int32), C.c_int32]
        L.orc_batches_free.argtypes = [vp]
        L.orc_batches_push.argtypes = [vp, C.POINTER(NqeColumn), C.c_int32]
        L.orc_batches_count.argtypes = [vp]
        L.orc_batches_num_columns.argtypes = [vp]
        L.orc_batch_num_rows.argtypes = [vp, C.c_int32]
        L.orc_batch_num_rows.restype = C.c_int64
        L.orc_batch_column.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(NqeColumn)]
        L.orc_scan.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
        L.orc_expr_evaluate.argtypes = [vp, C.c_int32, C.POINTER(NqeExprNode), C.c_int32, C.POINTER(vp)]
        L.orc_selection.argtypes = [vp, C.POINTER(NqeExprNode), C.c_int32, C.POINTER(vp)]
        L.orc_projection.argtypes = [vp, C.POINTER(NqeExprNode), C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
        L.orc_limit.argtypes = [vp, C.c_int64, C.POINTER(vp)]
        L.orc_offset.argtypes = [vp, C.c_int64, C.POINTER(vp)]
        L.orc_aggregate.argtypes = [vp, C.POINTER(NqeExprNode), C.c_int32, C.POINTER(NqeExprNode), C.c_int32,
                                    C.POINTER(NqeAggregate), C.c_int32, C.c_int32, C.POINTER(vp)]
        L.orc_hash_join.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.orc_hash_join_n.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.orc_xxhash64_word.restype = C.c_uint64
        L.orc_xxhash64_word.argtypes = [C.c_uint64]
        L.orc_xxhash64.restype = C.c_uint64
        L.orc_xxhash64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.orc_synth_fill.argtypes = [C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, vp]
        L.orc_headline_parallel.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, vp]
        L.orc_grouped_parallel.argtypes = [vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, vp]
        L.orc_grouped_parallel_masked.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp]
        L.orc_synth_fill_mt.argtypes = [C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, vp, C.c_int32]
        L.orc_csv_read.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(vp), C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.c_int32]
        L.orc_csv_read.restype = C.c_int
        _lib = L
    return _lib


def _check(st: int):
    if st != 0:
        raise ErrorCode(st, lib().orc_last_error().decode())


class _Handle:
    """Owns an orc_batches*."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.orc_batches_free(self.ptr)
            self.ptr = None

    def to_python(self) -> Batches:
        L = lib()
        out = []
        nc = L.orc_batches_num_columns(self.ptr)
        for b in range(L.orc_batches_count(self.ptr)):
            cols = []
            for c in range(nc):
                d = NqeColumn()
                _check(L.orc_batch_column(self.ptr, b, c, C.byref(d)))
                cols.append(Column.from_nqe_host(d))
            out.append(cols)
        return out


def upload(batches: Batches, dtypes: Optional[Sequence[int]] = None) -> _Handle:
    L = lib()
    if dtypes is None:
        dtypes = [int(c.dtype) for c in batches[0]] if batches else []
    arr = (C.c_int32 * max(1, len(dtypes)))(*dtypes)
    h = _Handle(L.orc_batches_new(arr, len(dtypes)))
    for cols in batches:
        keep: list = []
        carr = (NqeColumn * max(1, len(cols)))()
        for i, c in enumerate(cols):
            carr[i] = c.as_nqe(keep)
        _check(L.orc_batches_push(h.ptr, carr, len(cols)))
    return h


def _as_handle(x) -> _Handle:
    return x if isinstance(x, _Handle) else upload(x)


def _nodes(nodes):
    nodes = list(nodes or [])
    return nodes_array(nodes), len(nodes)


def scan(table, projection: Optional[Sequence[int]] = None) -> Batches:
    h = _as_handle(table)
    out = C.c_void_p()
    if projection is None:
        _check(lib().orc_scan(h.ptr, None, -1, C.byref(out)))
    else:
        arr = (C.c_int32 * max(1, len(projection)))(*projection)
        _check(lib().orc_scan(h.ptr, arr, len(projection), C.byref(out)))
    return _Handle(out.value).to_python()


def expr_evaluate(table, nodes, batch: int = 0) -> Column:
    h = _as_handle(table)
    arr, n = _nodes(nodes)
    out = C.c_void_p()
    _check(lib().orc_expr_evaluate(h.ptr, batch, arr, n, C.byref(out)))
    return _Handle(out.value).to_python()[0][0]


def selection(table, pred_nodes, raw: bool = False):
    h = _as_handle(table)
    arr, n = _nodes(pred_nodes)
    out = C.c_void_p()
    _check(lib().orc_selection(h.ptr, arr, n, C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def projection(table, exprs: Sequence[Sequence[NqeExprNode]], raw: bool = False):
    """exprs = [] models the empty-schema pass-through (projection.rs:47-48)."""
    h = _as_handle(table)
    flat, offs = [], [0]
    for e in exprs:
        flat.extend(e)
        offs.append(len(flat))
    arr, _ = _nodes(flat)
    oarr = (C.c_int32 * len(offs))(*offs)
    out = C.c_void_p()
    _check(lib().orc_projection(h.ptr, arr, oarr, len(exprs), C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def limit(table, n: int, raw: bool = False):
    h = _as_handle(table)
    out = C.c_void_p()
    _check(lib().orc_limit(h.ptr, n, C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def offset(table, n: int, raw: bool = False):
    h = _as_handle(table)
    out = C.c_void_p()
    _check(lib().orc_offset(h.ptr, n, C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def aggregate(table, aggs: Sequence[tuple], group_nodes=None, pred_nodes=None, executions: int = 1,
              raw: bool = False):
    """aggs: [(AggregateFunc, column_index), ...]"""
    h = _as_handle(table)
    parr, pn = _nodes(pred_nodes)
    garr, gn = _nodes(group_nodes)
    aarr = (NqeAggregate * max(1, len(aggs)))()
    for i, (f, c) in enumerate(aggs):
        aarr[i].func = int(f)
        aarr[i].column = int(c)
    out = C.c_void_p()
    _check(lib().orc_aggregate(h.ptr, parr, pn, garr, gn, aarr, len(aggs), executions, C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def hash_join(left, right, left_key: int, right_key: int, raw: bool = False, executions: int = 1):
    """executions > 1: the same HashJoin object executed that many times (its hash table is never cleared, quirk Q11)"""
    hl, hr = _as_handle(left), _as_handle(right)
    out = C.c_void_p()
    _check(lib().orc_hash_join_n(hl.ptr, hr.ptr, left_key, right_key, executions, C.byref(out)))
    r = _Handle(out.value)
    return r if raw else r.to_python()


def xxhash64_word(w: int) -> int:
    return int(lib().orc_xxhash64_word(w & 0xFFFFFFFFFFFFFFFF))


def xxhash64(data: bytes, seed: int = 0) -> int:
    return int(lib().orc_xxhash64(data, len(data), seed))


def csv_read(data: bytes, has_header: bool = True, delimiter: str = ",", max_read_records: int = 3, batch_size: int = 1_000_000):
    """CsvTable::try_create on a file image → (names, nullable flags, columns of the first batch)"""
    out = C.c_void_p()
    names = C.create_string_buffer(1 << 16)
    nullable = (C.c_int32 * 256)()
    _check(lib().orc_csv_read(data, len(data), int(has_header), ord(delimiter), max_read_records, batch_size, C.byref(out), names, len(names), nullable, 256))
    h = _Handle(out.value)
    cols = h.to_python()[0]
    raw = names.raw
    nm = raw.split(b"\0")[: len(cols)]
    return [x.decode() for x in nm], [bool(nullable[i]) for i in range(len(cols))], cols


def synth_fill(kind: int, seed: int, first_row: int, n: int, modulus: int = 1, base: int = 0):
    import numpy as np

    out = np.empty(n, dtype=np.uint64)
    _check(lib().orc_synth_fill(kind, seed, first_row, n, modulus, base, out.ctypes.data if n else None))
    return out


def headline_parallel(ids, v, limit: int, modulus: int, threads: int):
    """the OPTIMISED multi-core CPU form of the headline query (orc_headline_parallel: per-thread direct-mapped tables, merged) — not the
    reference's algorithm; bench.py's optional second CPU number.  → float64[modulus, 4] = count, sum, min, max per key"""
    import numpy as np

    ids = np.ascontiguousarray(ids, dtype=np.int64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros((modulus, 4), dtype=np.float64)
    _check(lib().orc_headline_parallel(ids.ctypes.data, v.ctypes.data, ids.size, limit, modulus, threads, out.ctypes.data))
    return out



def grouped_parallel(ids, v, limit, modulus: int, threads: int, v_valid=None, id_valid=None, with_rows: bool = False):
    """`select count(v), sum(v), min(v), max(v) from t [where id < limit] group by id % modulus` on `threads` threads
    (orc_grouped_parallel: per-thread direct-mapped tables over row ranges, merged) — the form the full-size parity checks use; it is
    itself checked against the reference-faithful single-threaded `aggregate` in tests/test_oracle_golden.py.  `v`: float64, int64 or
    uint64 (accumulated `as f64`); `limit` None: no predicate.  `v_valid` / `id_valid`: boolean row masks (False = NULL) or None.
    → float64[modulus, 4] = count, sum, min, max per key (count 0: no non-NULL value); with_rows: a fifth column, the rows of the
    group whatever their value's validity (rows 0: the key is no group)"""
    import numpy as np

    from naive_query_engine_amd import DType

    ids = np.ascontiguousarray(ids, dtype=np.int64)
    v = np.ascontiguousarray(v)
    dt = {np.dtype(np.float64): DType.FLOAT64, np.dtype(np.int64): DType.INT64, np.dtype(np.uint64): DType.UINT64}[v.dtype]
    assert v.size == ids.size
    masks = []
    for m in (id_valid, v_valid):
        if m is not None:
            m = np.ascontiguousarray(m, dtype=np.uint8)
            assert m.size == ids.size
        masks.append(m)
    stride = 5 if with_rows else 4
    out = np.zeros((modulus, stride), dtype=np.float64)
    _check(lib().orc_grouped_parallel_masked(ids.ctypes.data, None if masks[0] is None else masks[0].ctypes.data, v.ctypes.data,
                                             None if masks[1] is None else masks[1].ctypes.data, int(dt), ids.size, 0 if limit is None else 1,
                                             0 if limit is None else int(limit), modulus, threads, stride, out.ctypes.data))
    return out


def merge_grouped(parts):
    """fold float64[modulus, 4 or 5] partials of `grouped_parallel` over disjoint row ranges (counts, sums and rows add; min / max of the extremes)"""
    import numpy as np

    acc = None
    for p in parts:
        if acc is None:
            acc = p.copy()
        else:
            acc[:, 0] += p[:, 0]
            acc[:, 1] += p[:, 1]
            acc[:, 2] = np.minimum(acc[:, 2], p[:, 2])
            acc[:, 3] = np.maximum(acc[:, 3], p[:, 3])
            if acc.shape[1] > 4:
                acc[:, 4] += p[:, 4]
    return acc


def grouped_columns_parallel(ids, cols, limit, modulus: int, threads: int, valid=None, id_valid=None):
    """`grouped_parallel` once per value column of a query with several: cols = {column index: array}, valid = {column index: bool
    mask} (missing: no NULLs) → {column index: float64[modulus, 5]} (count, sum, min, max of the column's non-NULL values per key, and
    the group's rows).  Partials of disjoint row ranges fold with `merge_grouped_columns`; `finalize_grouped` turns them into the
    query's output columns."""
    valid = valid or {}
    return {c: grouped_parallel(ids, a, limit, modulus, threads, v_valid=valid.get(c), id_valid=id_valid, with_rows=True) for c, a in cols.items()}


def merge_grouped_columns(parts):
    return {c: merge_grouped([p[c] for p in parts]) for c in parts[0]}


def finalize_grouped(state, aggs):
    """{column: float64[modulus, 5]} + the aggregate list [(func, column)] → (keys of the groups in ascending order, one float64 array
    per aggregate) with the reference's finals: count of non-NULL values (count.rs:76), sum (sum.rs:115), avg = sum / cnt — NaN for a
    group without values (avg.rs:121), min / max from f64::MAX / f64::MIN (max.rs:30, min.rs)"""
    import numpy as np

    from naive_query_engine_amd import AggregateFunc as A

    any_col = next(iter(state.values()))
    live = np.nonzero(any_col[:, 4] > 0)[0]
    out = []
    for func, c in aggs:
        s = state[c][live]
        if func == A.Count:
            out.append(s[:, 0])
        elif func == A.Sum:
            out.append(s[:, 1])
        elif func == A.Avg:
            with np.errstate(invalid="ignore", divide="ignore"):
                out.append(s[:, 1] / s[:, 0])
        elif func == A.Min:
            out.append(s[:, 2])
        else:
            out.append(s[:, 3])
    return live, out


def synth_fill_mt(kind: int, seed: int, first_row: int, n: int, modulus: int = 1, base: int = 0, threads: int = 16):
    """`synth_fill` on several threads → uint64[n]"""
    import numpy as np

    out = np.empty(n, dtype=np.uint64)
    _check(lib().orc_synth_fill_mt(kind, seed, first_row, n, modulus, base, out.ctypes.data, threads))
    return out
