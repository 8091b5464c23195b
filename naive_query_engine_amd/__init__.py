"""naive_query_engine_amd — MI355X-native physical execution layer for naive-query-engine's hot
operators (filter, projection, hash group-by aggregate, inner hash join) and its CSV ingest,
hand-written HIP for gfx950 behind the C ABI in include/nqe.h.

Layout:
  csrc/            HIP kernels + the C-ABI implementation (libnqe_hip.so, built in-tree)
  host/naive_db.hpp C++ mirror of the reference's operator classes over the C ABI
  capi.py          ctypes binding of include/nqe.h (fails loudly if the library is missing)
  arrow_host.py    numpy-backed Arrow-layout host containers (plumbing)
  expression.py    host mirror of ColumnExpr / PhysicalLiteralExpr / PhysicalBinaryExpr
  physical_plan.py host mirror of the PhysicalPlan operators (ScanPlan, SelectionPlan, ...)
  rewrite.py       plan rewrite pass (unfused reference-shaped tree → fused device operators), Catalog / NaiveDB surface
  parallel.py      row-range sharding across GPUs: thin caller of the C ABI's sharded entry points (RCCL)
"""
from .arrow_host import (AggregateFunc, Column, DType, ErrorCode, Field, Operator, RecordBatch, ScalarValue, Status,
                         read_csv)
from .expression import ColumnExpr, PhysicalBinaryExpr, PhysicalExpr, PhysicalLiteralExpr

__all__ = [
    "AggregateFunc", "Column", "DType", "ErrorCode", "Field", "Operator", "RecordBatch", "ScalarValue", "Status",
    "read_csv", "ColumnExpr", "PhysicalBinaryExpr", "PhysicalExpr", "PhysicalLiteralExpr",
]


def __getattr__(name):
    # capi / physical_plan load libnqe_hip.so; keep `import naive_query_engine_amd` usable by tools
    # that only need the host containers, but never fall back to anything else.
    if name in ("capi", "physical_plan", "parallel", "rewrite"):
        import importlib

        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
