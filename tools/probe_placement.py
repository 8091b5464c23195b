#!/usr/bin/env python
"""Where does the headline's process-to-process spread come from?  (profiles/r06_notes.md: 2.40-2.65 ms for the same kernel on one box, stable
inside a process.)  ONE process: the headline's two 8 GB columns allocated again and again — fresh hipMalloc each time, the previous pair
freed or kept, the second column at different offsets inside one larger allocation — and the headline's kernel timed over each placement.
   python tools/probe_placement.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from naive_query_engine_amd import AggregateFunc, DType, Operator, capi  # noqa: E402
from naive_query_engine_amd.expression import binop, col, lit_i64  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**9
ctx = capi.Context(0)
ctx.reserve(8 << 30)
fields = [type("F", (), {"name": "id"})(), type("F", (), {"name": "v"})()]
pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(fields)
key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(fields)
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]


def time_pair(idp, vp, tag):
    t = ctx.table_from_device([(DType.INT64, n, idp, None), (DType.FLOAT64, n, vp, None)])
    import time
    t0 = time.time()
    while time.time() - t0 < 0.4:  # (the clocks settle: the first executions after a fill measured 15 % slower)
        ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)
    ctx.synchronize()
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(30):
        ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)
    ctx.synchronize()
    ctx.timing_enable(False)
    rep = ctx.timing_report()
    ms = rep["agg_grouped_fast"][0] / rep["agg_grouped_fast"][1]
    print(f"{tag:46s} id @ {idp:#x} (mod 2 MB {idp % (2 << 20):#9x})  v @ {vp:#x} (v - id = {(vp - idp) / 2**20:10.2f} MB)  kernel {ms:.4f} ms", flush=True)


def fill(ids, v):
    ids.copy_(torch.arange(ids.numel(), device="cuda", dtype=torch.int64))
    v.uniform_(0.0, 100.0)


keep = []
for rep in range(4):  # fresh allocations, the previous pair freed first
    ids = torch.empty(n, dtype=torch.int64, device="cuda")
    v = torch.empty(n, dtype=torch.float64, device="cuda")
    fill(ids, v)
    time_pair(ids.data_ptr(), v.data_ptr(), f"fresh pair {rep} (previous freed)")
    del ids, v
    torch.cuda.empty_cache()
for rep in range(3):  # fresh allocations while the previous ones stay allocated (other physical pages)
    ids = torch.empty(n, dtype=torch.int64, device="cuda")
    v = torch.empty(n, dtype=torch.float64, device="cuda")
    fill(ids, v)
    time_pair(ids.data_ptr(), v.data_ptr(), f"fresh pair {rep} (previous kept)")
    keep.append((ids, v))
del keep
torch.cuda.empty_cache()
# one allocation, the value column at different distances behind the id column
big = torch.empty(2 * n + (1100 << 20) // 8, dtype=torch.int64, device="cuda")
for off_bytes in (32 << 20, 0, 65536, 4096, 1 << 20, 0, (2 << 20) + 4096, 4096, 3 << 20, (16 << 20) + 65536, 0, 32 << 20, 20480, 45056, 1 << 30):
    ids = big[:n]
    v = big[n + off_bytes // 8: 2 * n + off_bytes // 8].view(torch.float64)
    fill(ids, v)
    time_pair(ids.data_ptr(), v.data_ptr(), f"one allocation, v {off_bytes} bytes behind id's end")
