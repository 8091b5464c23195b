#!/usr/bin/env python
"""One rank through RCCL with NQE_FORCE_EXCHANGE=1: the equal-count gather fast path of parallel._gather_table and the
one-collective aggregate exchange must reproduce the local results (run on the GPU box; 2-rank runs need 2 GPUs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NQE_FORCE_EXCHANGE"] = "1"
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29547"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
import numpy as np
import torch
import torch.distributed as dist

from naive_query_engine_amd import AggregateFunc, Column, capi, parallel
from naive_query_engine_amd.expression import col

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = capi.Context(0)
rng = np.random.default_rng(0)
nb, npr = 5000, 200_000
left = ctx.table_from_host([Column.from_numpy(rng.permutation(nb).astype(np.int64)), Column.from_numpy(rng.integers(0, 99, nb).astype(np.int64))])
right = ctx.table_from_host([Column.from_numpy(rng.integers(0, nb, npr).astype(np.int64)), Column.from_numpy(rng.random(npr))])
local = ctx.hash_join(left, right, 0, 0).to_host()
gathered = parallel.sharded_hash_join(ctx, left, right, 0, 0, gather=True).to_host()
assert all((a.to_numpy() == b.to_numpy()).all() for a, b in zip(local, gathered)), "gathered join differs"


class F:
    def __init__(self, n):
        self.name = n


aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 1)]
a1 = ctx.aggregate(right, aggs, group_nodes=col(0).flatten([F("k"), F("v")])).to_host()
a2, _ = parallel.sharded_aggregate(ctx, right, aggs, group_nodes=col(0).flatten([F("k"), F("v")]))
assert all(np.allclose(a.to_numpy().astype(float), b.to_numpy().astype(float), rtol=1e-12) for a, b in zip(a1, a2.to_host())), "sharded aggregate differs"
print("nccl single-rank exchange checks passed")
dist.barrier()
dist.destroy_process_group()
