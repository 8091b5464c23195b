#!/usr/bin/env python
"""One rank through real RCCL (world size 1): what the exchange of the sharded headline adds to a step — partial aggregate, pack,
ncclAllGather, merge from the gathered buffer — all behind nqe_sharded_aggregate_execute on the context's stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29549"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
import torch
import torch.distributed as dist

from naive_query_engine_amd import AggregateFunc, DType, Operator, capi, parallel
from naive_query_engine_amd.expression import binop, col, lit_i64


class F:
    def __init__(self, n):
        self.name = n


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
ctx = capi.Context(0)
comm = parallel.make_comm(ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**9
ids = torch.empty(n, dtype=torch.int64, device=dev)
v = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ctx.synth_fill(0, 0, 0, n, 1, 0, ids.data_ptr())
ctx.synth_fill(2, 3, 0, n, 1, 0, v.data_ptr())
ctx.synchronize()
t = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
f = [F("id"), F("v")]
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        r = fn(); del r
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); del r
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


local = timed(lambda: ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred))
partial = timed(lambda: ctx.aggregate_partial(t, aggs, group_nodes=key, pred_nodes=pred))
sharded = timed(lambda: comm.sharded_aggregate(t, aggs, group_nodes=key, pred_nodes=pred))
print(f"{n} rows: single-GPU aggregate {local:.3f} ms; partial only {partial:.3f} ms; sharded (partial + pack + all-gather + merge) {sharded:.3f} ms: "
      f"the exchange adds {sharded - local:.3f} ms to a step (RCCL {capi.Comm.rccl_version()})")
comm.close()
dist.destroy_process_group()
