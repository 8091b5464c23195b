#!/usr/bin/env python
"""Phase timing of the sharded aggregate's exchange path on one rank through RCCL (NQE_FORCE_EXCHANGE=1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NQE_FORCE_EXCHANGE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

from naive_query_engine_amd import AggregateFunc, DType, Operator, capi, parallel
from naive_query_engine_amd.expression import binop, col, lit_i64


class F:
    def __init__(self, n):
        self.name = n


torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**9
idt = torch.empty(n, dtype=torch.int64, device=dev)
vt = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ctx.synth_fill(0, 0, 0, n, 1, 0, idt.data_ptr())
ctx.synth_fill(2, 3, 0, n, 1, 0, vt.data_ptr())
t = ctx.table_from_device([(DType.INT64, n, idt.data_ptr(), None), (DType.FLOAT64, n, vt.data_ptr(), None)])
f = [F("id"), F("v")]
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)


def sync():
    torch.cuda.synchronize()
    ctx.synchronize()


def timed(fn, reps=10):
    for _ in range(3):
        r = fn(); del r
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); del r
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"single-GPU aggregate            : {timed(lambda: ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred)):.3f} ms")
print(f"aggregate_partial               : {timed(lambda: ctx.aggregate_partial(t, aggs, group_nodes=key, pred_nodes=pred)):.3f} ms")
state, keys = ctx.aggregate_partial(t, aggs, group_nodes=key, pred_nodes=pred)
print(f"merge of own partial            : {timed(lambda: ctx.aggregate_merge([state], [keys], aggs)):.3f} ms")
cols = parallel.table_columns_as_tensors(keys, dev) + parallel.table_columns_as_tensors(state, dev)
print(f"all_gather_rows ({len(cols)} cols x {cols[0].numel()}) : {timed(lambda: parallel.all_gather_rows(cols)):.3f} ms")
print(f"sharded_aggregate (whole)       : {timed(lambda: parallel.sharded_aggregate(ctx, t, aggs, group_nodes=key, pred_nodes=pred)):.3f} ms")
dist.barrier()
dist.destroy_process_group()
