"""ctypes binding of include/nqe.h (libnqe_hip.so, built in-tree by __graft_entry__.build()).

There is NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Everything that computes runs in the HIP library; this module only marshals
Arrow-layout buffers and expression encodings across the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from .arrow_host import (DEVICE, Column, DType, ErrorCode, NqeAggregate, NqeColumn, NqeCsvOptions, NqeExprNode, Status, bitmap_bytes,
                         nodes_array)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NQE_LIB_PATH: another build of the same library (tools/build_variant.sh) for A/B runs on one GPU box
LIB_PATH = os.environ.get("NQE_LIB_PATH") or os.path.join(_HERE, "libnqe_hip.so")

# every symbol include/nqe.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "nqe_abi_version", "nqe_ctx_create", "nqe_ctx_destroy", "nqe_ctx_synchronize", "nqe_ctx_memory_stats", "nqe_ctx_trim", "nqe_ctx_reserve", "nqe_last_error",
    "nqe_last_global_error", "nqe_ctx_timing_enable", "nqe_ctx_timing_query", "nqe_ctx_timing_reset", "nqe_ctx_timing_report", "nqe_ctx_jit_wait",
    "nqe_table_create", "nqe_table_create_flags", "nqe_table_release", "nqe_table_num_rows", "nqe_table_num_columns", "nqe_table_column",
    "nqe_table_download_column", "nqe_table_project", "nqe_table_slice", "nqe_table_concat", "nqe_table_pack_words",
    "nqe_table_unpack_words", "nqe_csv_infer_schema", "nqe_csv_read", "nqe_expr_evaluate",
    "nqe_filter", "nqe_selection_execute", "nqe_projection_execute", "nqe_selection_projection_execute",
    "nqe_aggregate_execute", "nqe_aggregate_partial", "nqe_aggregate_merge", "nqe_aggregate_merge_packed", "nqe_hash_join_execute",
    "nqe_hash_join_build", "nqe_hash_join_probe", "nqe_join_table_release", "nqe_take", "nqe_synth_fill",
    "nqe_device_alloc", "nqe_device_free",
    "nqe_comm_get_unique_id", "nqe_comm_rccl_version", "nqe_comm_create", "nqe_comm_create_custom", "nqe_comm_create_p2p", "nqe_comm_destroy", "nqe_comm_rank",
    "nqe_comm_world", "nqe_table_all_gather", "nqe_sharded_aggregate_execute", "nqe_sharded_hash_join_probe",
    "nqe_sharded_selection_projection_execute", "nqe_table_import_arrow", "nqe_table_export_arrow",
]

COMM_ID_BYTES = 128    # NQE_COMM_ID_BYTES
EXCHANGE_ROWS = 4096   # NQE_EXCHANGE_ROWS

# nqe_transport (include/nqe.h): the exchange's vtable
_AG_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_AGV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p)
_GRP_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p)
_DESTROY_FN = C.CFUNCTYPE(None, C.c_void_p)


class ArrowSchemaStruct(C.Structure):
    """struct ArrowSchema of the Arrow C Data Interface (72 bytes)"""
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArrayStruct(C.Structure):
    """struct ArrowArray of the Arrow C Data Interface (80 bytes)"""
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


_P2P_SEND_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p)


class NqeP2P(C.Structure):
    """nqe_p2p (include/nqe.h): point-to-point primitives the library builds its collectives from (send and recv share a signature)"""
    _fields_ = [("user", C.c_void_p), ("send", _P2P_SEND_FN), ("recv", _P2P_SEND_FN), ("group_begin", _GRP_FN), ("group_end", _GRP_FN),
                ("destroy", _DESTROY_FN)]


class NqeTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", _AG_FN), ("all_gather_v", _AGV_FN), ("group_begin", _GRP_FN), ("group_end", _GRP_FN),
                ("destroy", _DESTROY_FN)]


_lib = None


def lib():
    """Loads libnqe_hip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). naive_query_engine_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    pvp = C.POINTER(vp)
    nodes = C.POINTER(NqeExprNode)
    sig = {
        "nqe_abi_version": (C.c_uint32, []),
        "nqe_ctx_create": (i32, [i32, vp, pvp]),
        "nqe_ctx_destroy": (i32, [vp]),
        "nqe_ctx_synchronize": (i32, [vp]),
        "nqe_last_error": (C.c_char_p, [vp]),
        "nqe_last_global_error": (C.c_char_p, []),
        "nqe_ctx_memory_stats": (i32, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "nqe_ctx_trim": (i32, [vp]),
        "nqe_ctx_reserve": (i32, [vp, C.c_size_t]),
        "nqe_ctx_timing_enable": (i32, [vp, i32]),
        "nqe_ctx_timing_query": (i32, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
        "nqe_ctx_timing_reset": (i32, [vp]),
        "nqe_ctx_timing_report": (i32, [vp, C.c_char_p, i64, C.POINTER(i64)]),
        "nqe_ctx_jit_wait": (i32, [vp]),
        "nqe_table_create": (i32, [vp, C.POINTER(NqeColumn), i32, pvp]),
        "nqe_table_create_flags": (i32, [vp, C.POINTER(NqeColumn), i32, C.c_uint32, pvp]),
        "nqe_table_release": (i32, [vp]),
        "nqe_table_num_rows": (i64, [vp]),
        "nqe_table_num_columns": (i32, [vp]),
        "nqe_table_column": (i32, [vp, i32, C.POINTER(NqeColumn)]),
        "nqe_table_download_column": (i32, [vp, i32, vp, vp, vp]),
        "nqe_table_project": (i32, [vp, vp, C.POINTER(i32), i32, pvp]),
        "nqe_table_slice": (i32, [vp, vp, i64, i64, pvp]),
        "nqe_table_concat": (i32, [vp, pvp, i32, pvp]),
        "nqe_table_pack_words": (i32, [vp, pvp, i32, i64, vp]),
        "nqe_table_unpack_words": (i32, [vp, vp, i32, i32, i64, C.POINTER(i64), C.POINTER(i32), pvp]),
        "nqe_csv_infer_schema": (i32, [vp, C.c_char_p, i64, C.POINTER(NqeCsvOptions), i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p, i64,
                                       C.POINTER(i64)]),
        "nqe_csv_read": (i32, [vp, vp, i32, i64, C.POINTER(NqeCsvOptions), C.POINTER(i32), i32, pvp]),
        "nqe_expr_evaluate": (i32, [vp, vp, nodes, i32, pvp]),
        "nqe_filter": (i32, [vp, vp, vp, i32, pvp]),
        "nqe_selection_execute": (i32, [vp, vp, nodes, i32, pvp]),
        "nqe_projection_execute": (i32, [vp, vp, nodes, C.POINTER(i32), i32, pvp]),
        "nqe_selection_projection_execute": (i32, [vp, vp, nodes, i32, nodes, C.POINTER(i32), i32, pvp]),
        "nqe_aggregate_execute": (i32, [vp, vp, nodes, i32, nodes, i32, C.POINTER(NqeAggregate), i32, pvp, pvp]),
        "nqe_aggregate_partial": (i32, [vp, vp, nodes, i32, nodes, i32, C.POINTER(NqeAggregate), i32, pvp, pvp]),
        "nqe_aggregate_merge": (i32, [vp, pvp, pvp, i32, C.POINTER(NqeAggregate), i32, pvp, pvp]),
        "nqe_aggregate_merge_packed": (i32, [vp, vp, i32, i64, i32, i32, C.POINTER(NqeAggregate), i32, pvp, pvp]),
        "nqe_hash_join_execute": (i32, [vp, vp, vp, i32, i32, pvp]),
        "nqe_hash_join_build": (i32, [vp, vp, i32, pvp]),
        "nqe_hash_join_probe": (i32, [vp, vp, vp, i32, pvp]),
        "nqe_join_table_release": (i32, [vp]),
        "nqe_take": (i32, [vp, vp, vp, i32, pvp]),
        "nqe_synth_fill": (i32, [vp, i32, u64, i64, i64, u64, i64, vp]),
        "nqe_device_alloc": (i32, [vp, C.c_size_t, pvp]),
        "nqe_device_free": (i32, [vp, vp]),
        "nqe_comm_get_unique_id": (i32, [vp]),
        "nqe_comm_rccl_version": (i32, [C.POINTER(i32)]),
        "nqe_comm_create": (i32, [vp, vp, i32, i32, pvp]),
        "nqe_comm_create_custom": (i32, [vp, C.POINTER(NqeTransport), i32, i32, pvp]),
        "nqe_comm_create_p2p": (i32, [vp, C.POINTER(NqeP2P), i32, i32, pvp]),
        "nqe_comm_destroy": (i32, [vp]),
        "nqe_comm_rank": (i32, [vp]),
        "nqe_comm_world": (i32, [vp]),
        "nqe_table_all_gather": (i32, [vp, vp, pvp]),
        "nqe_sharded_aggregate_execute": (i32, [vp, vp, nodes, i32, nodes, i32, C.POINTER(NqeAggregate), i32, pvp, pvp]),
        "nqe_sharded_hash_join_probe": (i32, [vp, vp, vp, i32, i32, pvp]),
        "nqe_sharded_selection_projection_execute": (i32, [vp, vp, nodes, i32, nodes, C.POINTER(i32), i32, i32, pvp]),
        "nqe_table_import_arrow": (i32, [vp, vp, vp, pvp]),
        "nqe_table_export_arrow": (i32, [vp, C.POINTER(C.c_char_p), vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


TABLE_IMMUTABLE = 1  # NQE_TABLE_IMMUTABLE


class Context:
    """nqe_ctx: one device + one HIP stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        L = lib()
        h = C.c_void_p()
        st = L.nqe_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0:
            raise ErrorCode(st, L.nqe_last_global_error().decode())
        self.handle = h
        self.device = device
        self.stream = stream  # the hipStream_t the context launches on when it was given one (None: a private stream)

    def check(self, st: int):
        if st != 0:
            raise ErrorCode(st, lib().nqe_last_error(self.handle).decode())

    def synchronize(self):
        self.check(lib().nqe_ctx_synchronize(self.handle))

    def close(self):
        if self.handle:
            lib().nqe_ctx_destroy(self.handle)
            self.handle = None

    # ---- timing (bench.py)
    def timing_enable(self, on: bool = True):
        self.check(lib().nqe_ctx_timing_enable(self.handle, 1 if on else 0))

    def timing_reset(self):
        self.check(lib().nqe_ctx_timing_reset(self.handle))

    def memory_stats(self):
        """(bytes held by live handles, bytes cached in the block pool)"""
        live, pooled = C.c_int64(), C.c_int64()
        self.check(lib().nqe_ctx_memory_stats(self.handle, C.byref(live), C.byref(pooled)))
        return live.value, pooled.value

    def reserve(self, nbytes: int) -> None:
        """nqe_ctx_reserve: take `nbytes` from the driver now; later allocations of the context are served from that block"""
        self.check(lib().nqe_ctx_reserve(self.handle, nbytes))

    def trim(self) -> None:
        self.check(lib().nqe_ctx_trim(self.handle))

    def timing_query(self, name_substr: str = ""):
        ms, cnt = C.c_double(), C.c_int64()
        self.check(lib().nqe_ctx_timing_query(self.handle, name_substr.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def jit_wait(self):
        """blocks until the run-time specialisations being compiled for this context are ready (the next execution takes them)"""
        self.check(lib().nqe_ctx_jit_wait(self.handle))

    def timing_report(self) -> dict:
        """{kernel name: (total ms, launches)} since the last reset, by exact name"""
        need = C.c_int64()
        self.check(lib().nqe_ctx_timing_report(self.handle, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(1, need.value))
        self.check(lib().nqe_ctx_timing_report(self.handle, buf, len(buf), C.byref(need)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt = line.split("\t")
            out[name] = (float(ms), int(cnt))
        return out

    # ---- tables
    def table_from_host(self, columns: Sequence[Column]) -> "Table":
        keep: list = []
        arr = (NqeColumn * max(1, len(columns)))()
        for i, c in enumerate(columns):
            arr[i] = c.as_nqe(keep)
        h = C.c_void_p()
        self.check(lib().nqe_table_create(self.handle, arr, len(columns), C.byref(h)))
        return Table(self, h)

    def table_from_device(self, cols: Sequence[tuple], immutable: bool = False, keepalive=None) -> "Table":
        """cols: [(DType, length, values_device_ptr, validity_device_ptr_or_None)] — zero-copy; the caller keeps the memory alive
        (e.g. torch tensors) while this table is in use.  Operator outputs never alias it — unless `immutable` (NQE_TABLE_IMMUTABLE):
        the caller then promises the memory stays alive and unmodified while ANY table derived from this one lives; pass the
        owning objects as `keepalive` and every table derived from this one holds on to them."""
        arr = (NqeColumn * max(1, len(cols)))()
        for i, (dt, n, vptr, valid) in enumerate(cols):
            arr[i].dtype = int(dt)
            arr[i].location = DEVICE
            arr[i].length = n
            arr[i].null_count = -1 if valid else 0
            arr[i].values = vptr
            arr[i].validity = valid
        h = C.c_void_p()
        self.check(lib().nqe_table_create_flags(self.handle, arr, len(cols), TABLE_IMMUTABLE if immutable else 0, C.byref(h)))
        t = Table(self, h)
        if immutable:
            t._keep = [keepalive]
        return t

    def table_from_arrow(self, record_batch) -> "Table":
        """a pyarrow.RecordBatch through the Arrow C Data Interface (nqe_table_import_arrow): the batch is exported as a struct array
        whose children are the columns; the library copies the buffers to HBM and releases the exported array"""
        arr, sch = ArrowArrayStruct(), ArrowSchemaStruct()
        record_batch._export_to_c(C.addressof(arr), C.addressof(sch))
        h = C.c_void_p()
        try:
            self.check(lib().nqe_table_import_arrow(self.handle, C.addressof(arr), C.addressof(sch), C.byref(h)))
        finally:
            # the schema is only borrowed by the import; the array is released by a successful import (and by us otherwise)
            for st in (arr, sch):
                if st.release:
                    C.CFUNCTYPE(None, C.c_void_p)(st.release)(C.addressof(st))
        return Table(self, h)

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(lib().nqe_device_alloc(self.handle, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr: int):
        self.check(lib().nqe_device_free(self.handle, C.c_void_p(ptr)))

    def synth_fill(self, kind: int, seed: int, first_row: int, n: int, modulus: int, base: int, out_ptr: int):
        self.check(lib().nqe_synth_fill(self.handle, kind, seed, first_row, n, modulus, base, C.c_void_p(out_ptr)))

    # ---- operators (thin wrappers; see include/nqe.h for the reference citations)
    def _nodes(self, nodes):
        nodes = list(nodes or [])
        return (nodes_array(nodes) if nodes else None), len(nodes)

    def _flat(self, exprs):
        flat, offs = [], [0]
        for e in exprs:
            flat.extend(e)
            offs.append(len(flat))
        return nodes_array(flat), (C.c_int32 * len(offs))(*offs), len(exprs)

    def expr_evaluate(self, table: "Table", nodes) -> "Table":
        arr, n = self._nodes(nodes)
        h = C.c_void_p()
        self.check(lib().nqe_expr_evaluate(self.handle, table.handle, arr, n, C.byref(h)))
        return Table(self, h, derived_from=(table,))

    def filter(self, table: "Table", pred_table: "Table", pred_column: int = 0) -> "Table":
        h = C.c_void_p()
        self.check(lib().nqe_filter(self.handle, table.handle, pred_table.handle, pred_column, C.byref(h)))
        return Table(self, h)

    def selection(self, table: "Table", pred_nodes) -> "Table":
        arr, n = self._nodes(pred_nodes)
        h = C.c_void_p()
        self.check(lib().nqe_selection_execute(self.handle, table.handle, arr, n, C.byref(h)))
        return Table(self, h)

    def projection(self, table: "Table", exprs) -> "Table":
        arr, offs, ne = self._flat(exprs)
        h = C.c_void_p()
        self.check(lib().nqe_projection_execute(self.handle, table.handle, arr, offs, ne, C.byref(h)))
        return Table(self, h, derived_from=(table,))

    def selection_projection(self, table: "Table", pred_nodes, exprs) -> "Table":
        parr, pn = self._nodes(pred_nodes)
        arr, offs, ne = self._flat(exprs)
        h = C.c_void_p()
        self.check(lib().nqe_selection_projection_execute(self.handle, table.handle, parr, pn, arr, offs, ne, C.byref(h)))
        return Table(self, h)

    def _aggs(self, aggs):
        a = (NqeAggregate * max(1, len(aggs)))()
        for i, (f, c) in enumerate(aggs):
            a[i].func = int(f)
            a[i].column = int(c)
        return a

    def aggregate(self, table: "Table", aggs, group_nodes=None, pred_nodes=None, with_keys: bool = False):
        parr, pn = self._nodes(pred_nodes)
        garr, gn = self._nodes(group_nodes)
        h, k = C.c_void_p(), C.c_void_p()
        self.check(lib().nqe_aggregate_execute(self.handle, table.handle, parr, pn, garr, gn, self._aggs(aggs), len(aggs),
                                               C.byref(h), C.byref(k) if with_keys else None))
        out = Table(self, h)
        if with_keys:
            return out, (Table(self, k) if k.value else None)
        return out

    def aggregate_partial(self, table: "Table", aggs, group_nodes=None, pred_nodes=None):
        parr, pn = self._nodes(pred_nodes)
        garr, gn = self._nodes(group_nodes)
        h, k = C.c_void_p(), C.c_void_p()
        self.check(lib().nqe_aggregate_partial(self.handle, table.handle, parr, pn, garr, gn, self._aggs(aggs), len(aggs),
                                               C.byref(h), C.byref(k)))
        return Table(self, h), (Table(self, k) if k.value else None)

    def aggregate_merge(self, states: Sequence["Table"], keys: Optional[Sequence["Table"]], aggs):
        n = len(states)
        sarr = (C.c_void_p * n)(*[s.handle for s in states])
        karr = (C.c_void_p * n)(*[k.handle for k in keys]) if keys else None
        h, k = C.c_void_p(), C.c_void_p()
        self.check(lib().nqe_aggregate_merge(self.handle, sarr, karr, n, self._aggs(aggs), len(aggs), C.byref(h), C.byref(k)))
        return Table(self, h), (Table(self, k) if k.value else None)

    def aggregate_merge_packed(self, gathered_ptr: int, num_parts: int, stride_rows: int, grouped: bool, key_dtype: int, aggs):
        """merge straight from an all-gathered pack_words buffer (row counts read on the device); None when some part was sent
        header-only (its row count exceeds the stride) and the exact-size exchange has to be used"""
        h, k = C.c_void_p(), C.c_void_p()
        self.check(lib().nqe_aggregate_merge_packed(self.handle, C.c_void_p(gathered_ptr), num_parts, stride_rows, 1 if grouped else 0,
                                                    int(key_dtype), self._aggs(aggs), len(aggs), C.byref(h), C.byref(k)))
        if not h.value:
            return None
        return Table(self, h), (Table(self, k) if k.value else None)

    def hash_join(self, left: "Table", right: "Table", left_key: int, right_key: int) -> "Table":
        h = C.c_void_p()
        self.check(lib().nqe_hash_join_execute(self.handle, left.handle, right.handle, left_key, right_key, C.byref(h)))
        return Table(self, h, derived_from=(right,))

    def hash_join_build(self, left: "Table", left_key: int) -> "JoinTable":
        h = C.c_void_p()
        self.check(lib().nqe_hash_join_build(self.handle, left.handle, left_key, C.byref(h)))
        return JoinTable(self, h)

    def hash_join_probe(self, jt: "JoinTable", right: "Table", right_key: int) -> "Table":
        h = C.c_void_p()
        self.check(lib().nqe_hash_join_probe(self.handle, jt.handle, right.handle, right_key, C.byref(h)))
        return Table(self, h, derived_from=(right,))

    def take(self, table: "Table", idx_table: "Table", idx_column: int = 0) -> "Table":
        h = C.c_void_p()
        self.check(lib().nqe_take(self.handle, table.handle, idx_table.handle, idx_column, C.byref(h)))
        return Table(self, h)

    def project(self, table: "Table", indices: Sequence[int]) -> "Table":
        arr = (C.c_int32 * max(1, len(indices)))(*indices)
        h = C.c_void_p()
        self.check(lib().nqe_table_project(self.handle, table.handle, arr, len(indices), C.byref(h)))
        return Table(self, h, derived_from=(table,), view=True)

    def slice(self, table: "Table", offset: int, length: int) -> "Table":
        h = C.c_void_p()
        self.check(lib().nqe_table_slice(self.handle, table.handle, offset, length, C.byref(h)))
        return Table(self, h)

    def concat(self, tables: Sequence["Table"]) -> "Table":
        arr = (C.c_void_p * max(1, len(tables)))(*[t.handle for t in tables])
        h = C.c_void_p()
        self.check(lib().nqe_table_concat(self.handle, arr, len(tables), C.byref(h)))
        return Table(self, h, derived_from=tuple(tables))

    # ---- CSV ingest (datasource/csv.rs)
    def csv_infer_schema(self, data: bytes, has_header: bool = True, delimiter: str = ",", max_read_records: int = 3, batch_size: int = 1_000_000):
        """→ (names, dtypes, nullable) from the first `max_read_records` records (host side)"""
        opt = NqeCsvOptions(int(has_header), ord(delimiter), max_read_records, batch_size)
        cap = 256
        nc, need = C.c_int32(), C.c_int64()
        dts, nul = (C.c_int32 * cap)(), (C.c_int32 * cap)()
        names = C.create_string_buffer(1 << 16)
        self.check(lib().nqe_csv_infer_schema(self.handle, data, len(data), C.byref(opt), cap, C.byref(nc), dts, nul, names, len(names), C.byref(need)))
        n = nc.value
        nm = names.raw[: need.value].split(b"\0")[:n]
        return [x.decode() for x in nm], [DType(dts[i]) for i in range(n)], [bool(nul[i]) for i in range(n)]

    def csv_read(self, data, dtypes: Sequence[DType], has_header: bool = True, delimiter: str = ",", batch_size: int = 1_000_000,
                 device_ptr: Optional[int] = None, nbytes: Optional[int] = None) -> "Table":
        """parses a file image (host `bytes`, or device memory via device_ptr/nbytes) into a device table"""
        opt = NqeCsvOptions(int(has_header), ord(delimiter), 3, batch_size)
        da = (C.c_int32 * max(1, len(dtypes)))(*[int(d) for d in dtypes])
        h = C.c_void_p()
        if device_ptr is not None:
            self.check(lib().nqe_csv_read(self.handle, C.c_void_p(device_ptr), 1, nbytes, C.byref(opt), da, len(dtypes), C.byref(h)))
        else:
            buf = C.c_char_p(data)
            self.check(lib().nqe_csv_read(self.handle, C.cast(buf, C.c_void_p), 0, len(data), C.byref(opt), da, len(dtypes), C.byref(h)))
        return Table(self, h)

    # ---- exchange plumbing (multi-GPU)
    def pack_words(self, tables: Sequence["Table"], stride_rows: int, dst_ptr: int) -> None:
        """columns of `tables` → dst[c*stride + r]; dst[ncols*stride] = row count (header)"""
        arr = (C.c_void_p * max(1, len(tables)))(*[t.handle for t in tables])
        self.check(lib().nqe_table_pack_words(self.handle, arr, len(tables), stride_rows, C.c_void_p(dst_ptr)))

    def unpack_words(self, src_ptr: int, counts: Sequence[int], dtypes: Sequence[DType], stride_rows: int) -> "Table":
        """len(counts) packed buffers back to back → one table, parts concatenated per column"""
        ca = (C.c_int64 * max(1, len(counts)))(*[int(c) for c in counts])
        da = (C.c_int32 * max(1, len(dtypes)))(*[int(d) for d in dtypes])
        h = C.c_void_p()
        self.check(lib().nqe_table_unpack_words(self.handle, C.c_void_p(src_ptr), len(counts), len(dtypes), stride_rows, ca, da, C.byref(h)))
        return Table(self, h)


class Comm:
    """nqe_comm: the exchange of the sharded operators over one context (include/nqe.h, "sharded operators").

    Comm.rccl(ctx, unique_id, rank, world): RCCL over xGMI, one device per rank; rank 0 draws the id with Comm.unique_id() and the
    host distributes the 128 bytes.  Comm.custom(ctx, transport, rank, world): the host's own transport — an object with
    all_gather(send_ptr, recv_ptr, nbytes) and all_gather_v(send_ptr, send_bytes, recv_ptr, offsets, sizes) over device pointers.
    The callbacks run on the calling thread while the context's stream may still be producing `send`: they must order themselves
    against it (parallel.HostStagedTransport synchronises the context first)."""

    def __init__(self, ctx: Context, handle, keep=None):
        self.ctx = ctx
        self.handle = handle
        self._keep = keep

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        st = lib().nqe_comm_get_unique_id(buf)
        if st != 0:
            raise ErrorCode(st, lib().nqe_last_global_error().decode())
        return buf.raw

    @staticmethod
    def rccl_version() -> int:
        v = C.c_int32()
        st = lib().nqe_comm_rccl_version(C.byref(v))
        if st != 0:
            raise ErrorCode(st, lib().nqe_last_global_error().decode())
        return v.value

    @classmethod
    def rccl(cls, ctx: Context, unique_id: bytes, rank: int, world: int) -> "Comm":
        assert len(unique_id) == COMM_ID_BYTES
        h = C.c_void_p()
        ctx.check(lib().nqe_comm_create(ctx.handle, C.c_char_p(unique_id), rank, world, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def custom(cls, ctx: Context, transport, rank: int, world: int) -> "Comm":
        def ag(user, send, recv, nbytes, stream):
            try:
                transport.all_gather(send or 0, recv or 0, int(nbytes))
                return 0
            except Exception:  # noqa: BLE001 - the status code is all that crosses the ABI
                import traceback

                traceback.print_exc()
                return 1

        def agv(user, send, send_bytes, recv, offs, sizes, stream):
            try:
                transport.all_gather_v(send or 0, int(send_bytes), recv or 0, [int(offs[r]) for r in range(world)], [int(sizes[r]) for r in range(world)])
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return 1

        tr = NqeTransport()
        tr.user = None
        tr.all_gather = _AG_FN(ag)
        tr.all_gather_v = _AGV_FN(agv)
        tr.group_begin = _GRP_FN(0)
        tr.group_end = _GRP_FN(0)
        tr.destroy = _DESTROY_FN(0)
        h = C.c_void_p()
        ctx.check(lib().nqe_comm_create_custom(ctx.handle, C.byref(tr), rank, world, C.byref(h)))
        return cls(ctx, h, keep=(tr, transport))

    @property
    def rank(self) -> int:
        return int(lib().nqe_comm_rank(self.handle))

    @property
    def world(self) -> int:
        return int(lib().nqe_comm_world(self.handle))

    def close(self):
        if self.handle and self.ctx.handle:
            lib().nqe_comm_destroy(self.handle)
        self.handle = None

    # ---- sharded operators
    def all_gather_table(self, table: "Table") -> "Table":
        h = C.c_void_p()
        self.ctx.check(lib().nqe_table_all_gather(self.handle, table.handle, C.byref(h)))
        return Table(self.ctx, h)

    def sharded_aggregate(self, table: "Table", aggs, group_nodes=None, pred_nodes=None):
        c = self.ctx
        parr, pn = c._nodes(pred_nodes)
        garr, gn = c._nodes(group_nodes)
        h, k = C.c_void_p(), C.c_void_p()
        c.check(lib().nqe_sharded_aggregate_execute(self.handle, table.handle, parr, pn, garr, gn, c._aggs(aggs), len(aggs), C.byref(h), C.byref(k)))
        return Table(c, h), (Table(c, k) if k.value else None)

    def sharded_hash_join_probe(self, jt: "JoinTable", right_local: "Table", right_key: int, gather: bool = False) -> "Table":
        h = C.c_void_p()
        self.ctx.check(lib().nqe_sharded_hash_join_probe(self.handle, jt.handle, right_local.handle, right_key, 1 if gather else 0, C.byref(h)))
        return Table(self.ctx, h, derived_from=(right_local,))

    def sharded_selection_projection(self, table: "Table", pred_nodes, exprs, gather: bool = False) -> "Table":
        c = self.ctx
        parr, pn = c._nodes(pred_nodes)
        arr, offs, ne = c._flat(exprs)
        h = C.c_void_p()
        c.check(lib().nqe_sharded_selection_projection_execute(self.handle, table.handle, parr, pn, arr, offs, ne, 1 if gather else 0, C.byref(h)))
        return Table(c, h)


class Table:
    """nqe_table: one device-resident RecordBatch (owned handle)."""

    def __init__(self, ctx: Context, handle, derived_from=(), view: bool = False):
        self.ctx = ctx
        self.handle = handle
        # what an NQE_TABLE_IMMUTABLE table was told to keep alive travels to every table that may alias its buffers; a zero-copy
        # view (nqe_table_project) of plainly borrowed memory keeps its parent table object (a handle, no memory) instead
        self._keep = [k for t in derived_from for k in getattr(t, "_keep", ())]
        if view:
            self._keep.append(derived_from)

    def __del__(self):
        self.release()

    def release(self):
        if getattr(self, "handle", None) and self.handle.value and _lib is not None and self.ctx.handle:
            _lib.nqe_table_release(self.handle)
        self.handle = None

    @property
    def num_rows(self) -> int:
        return int(lib().nqe_table_num_rows(self.handle))

    @property
    def num_columns(self) -> int:
        return int(lib().nqe_table_num_columns(self.handle))

    def column_info(self, i: int) -> NqeColumn:
        c = NqeColumn()
        self.ctx.check(lib().nqe_table_column(self.handle, i, C.byref(c)))
        return c

    def dtypes(self) -> List[DType]:
        return [DType(self.column_info(i).dtype) for i in range(self.num_columns)]

    def download_column(self, i: int) -> Column:
        info = self.column_info(i)
        dt, n = DType(info.dtype), int(info.length)
        if dt == DType.BOOLEAN:
            vals = np.zeros(bitmap_bytes(n), dtype=np.uint8)
        elif dt == DType.UTF8:
            vals = np.zeros(n + 1, dtype=np.int32)
        else:
            vals = np.zeros(n, dtype={DType.INT64: np.int64, DType.UINT64: np.uint64, DType.FLOAT64: np.float64}[dt])
        valid = np.zeros(bitmap_bytes(n), dtype=np.uint8) if info.validity else None
        data = np.zeros(int(info.data_length), dtype=np.uint8) if dt == DType.UTF8 else None
        self.ctx.check(lib().nqe_table_download_column(
            self.handle, i, vals.ctypes.data if vals.size else None,
            valid.ctypes.data if valid is not None and valid.size else None,
            data.ctypes.data if data is not None and data.size else None))
        return Column(dt, n, vals, valid, data)

    def to_host(self) -> List[Column]:
        return [self.download_column(i) for i in range(self.num_columns)]

    def to_arrow(self, names: Optional[Sequence[str]] = None):
        """the table as a pyarrow.RecordBatch through the Arrow C Data Interface (nqe_table_export_arrow): pyarrow takes over the
        exported structs and calls their release callbacks when the batch is garbage-collected"""
        import pyarrow as pa

        arr, sch = ArrowArrayStruct(), ArrowSchemaStruct()
        n = self.num_columns
        na = None
        if names is not None:
            assert len(names) == n
            na = (C.c_char_p * max(1, n))(*[s.encode() for s in names])
        self.ctx.check(lib().nqe_table_export_arrow(self.handle, na, C.addressof(arr), C.addressof(sch)))
        return pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))


class JoinTable:
    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value and _lib is not None and self.ctx.handle:
            _lib.nqe_join_table_release(self.handle)
        self.handle = None


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        _default_ctx = Context(dev)
    return _default_ctx
