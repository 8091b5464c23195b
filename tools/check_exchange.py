#!/usr/bin/env python
"""Sharded operators through the C ABI's exchange over real RCCL (backend "nccl"), one device per rank: every rank runs the
sharded operator on its row range and checks the result against the single-GPU operator over the whole input.

  python tools/check_exchange.py                                    # one rank (a 1-GPU box): RCCL with world size 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/check_exchange.py

Covers: the one-collective aggregate exchange (few groups), the exact-size path (> NQE_EXCHANGE_ROWS groups on a rank), the
un-grouped aggregate, disjoint key sets per rank, the join with the build side replicated and the probe side range-split
(gathered, probe order kept, shared key columns), filter+projection gathered with EQUAL shard counts (no count-dependent
shortcut may skip a synchronisation) and with ragged ones, a Utf8-key aggregate (strings exchanged and merged) and the gather of
nullable / Boolean / Utf8 columns."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29547"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import numpy as np
import torch
import torch.distributed as dist

from naive_query_engine_amd import AggregateFunc, Column, Operator, capi, parallel
from naive_query_engine_amd.expression import binop, col, lit_i64

rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
if torch.cuda.device_count() < world:
    sys.exit(f"check_exchange: {world} ranks need {world} devices, this box has {torch.cuda.device_count()}")
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
ctx = capi.Context(local_rank)
comm = parallel.make_comm(ctx)
assert comm.rank == rank and comm.world == world


class F:
    def __init__(self, n):
        self.name = n


def host_cols(t):
    return [c.to_numpy() for c in t.to_host()]


def shard(cols, n):
    lo, hi = parallel.shard_range(n, rank, world)
    return ctx.table_from_host([Column.from_numpy(c.to_numpy()[lo:hi], None if c.valid_mask().all() else c.valid_mask()[lo:hi]) for c in cols])


rng = np.random.default_rng(0)  # the same data on every rank
N = 400_003
ids = np.arange(N, dtype=np.int64)
v = rng.random(N) * 100.0
vmask = rng.random(N) > 0.03
cols = [Column.from_numpy(ids), Column.from_numpy(v, vmask)]
f = [F("id"), F("v")]
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
whole = ctx.table_from_host(cols)
mine = shard(cols, N)
pred = binop(col(0), Operator.Lt, lit_i64(N // 2)).flatten(f)
for name, key in (("128 groups", binop(col(0), Operator.Modulos, lit_i64(128)).flatten(f)),
                  ("6000 groups (exact-size path)", binop(col(0), Operator.Modulos, lit_i64(6000)).flatten(f)),
                  ("disjoint key sets", binop(col(0), Operator.Divide, lit_i64(N // 8192 + 1)).flatten(f)),
                  ("un-grouped", None)):
    p = None if name == "disjoint key sets" else pred
    exp, expk = ctx.aggregate(whole, aggs, group_nodes=key, pred_nodes=p, with_keys=True)
    got, gotk = comm.sharded_aggregate(mine, aggs, group_nodes=key, pred_nodes=p)
    e, g = host_cols(exp), host_cols(got)
    assert len(e) == len(g) and all(a.shape == b.shape for a, b in zip(e, g)), name
    assert (e[0] == g[0]).all(), f"{name}: counts differ"
    assert all(np.allclose(a.astype(float), b.astype(float), rtol=1e-9, atol=0, equal_nan=True) for a, b in zip(e, g)), f"{name}: aggregates differ"
    if key is not None:
        assert (host_cols(expk)[0] == host_cols(gotk)[0]).all(), f"{name}: keys differ"

# ---- join: build replicated, probe range-split, gathered
nb, npr = 5000, 200_001
left_cols = [Column.from_numpy(rng.permutation(nb).astype(np.int64)), Column.from_numpy(rng.integers(0, 99, nb).astype(np.int64))]
right_cols = [Column.from_numpy(rng.integers(-3, nb + 3, npr).astype(np.int64)), Column.from_numpy(rng.random(npr))]
left = ctx.table_from_host(left_cols)
exp = host_cols(ctx.hash_join(left, ctx.table_from_host(right_cols), 0, 0))
jt = ctx.hash_join_build(left, 0)
got = host_cols(comm.sharded_hash_join_probe(jt, shard(right_cols, npr), 0, gather=True))
assert len(exp) == len(got) == 4 and all(a.shape == b.shape and (a.view(np.int64) == b.view(np.int64)).all() for a, b in zip(exp, got)), "gathered join differs"
local = comm.sharded_hash_join_probe(jt, shard(right_cols, npr), 0, gather=False)
cnt = torch.tensor([local.num_rows], dtype=torch.int64, device=torch.device("cuda", local_rank))
dist.all_reduce(cnt)
assert int(cnt.item()) == len(exp[0]), "local join outputs do not add up"
# duplicate build keys (general two-pass probe)
dup_left = ctx.table_from_host([Column.from_numpy((np.arange(nb) // 2).astype(np.int64)), left_cols[1]])
exp = host_cols(ctx.hash_join(dup_left, ctx.table_from_host(right_cols), 0, 0))
got = host_cols(comm.sharded_hash_join_probe(ctx.hash_join_build(dup_left, 0), shard(right_cols, npr), 0, gather=True))
assert all(a.shape == b.shape and (a.view(np.int64) == b.view(np.int64)).all() for a, b in zip(exp, got)), "gathered join (duplicate keys) differs"

# ---- filter + projection, gathered: equal shard output counts (every row passes, N2 divisible by world) and ragged ones
N2 = 65536 * world
c2 = [Column.from_numpy(np.arange(N2, dtype=np.int64)), Column.from_numpy(rng.integers(18, 78, N2).astype(np.int64))]
w2 = ctx.table_from_host(c2)
lo2, hi2 = parallel.shard_range(N2, rank, world)
m2 = ctx.table_from_host([Column.from_numpy(c.to_numpy()[lo2:hi2]) for c in c2])
f2 = [F("id"), F("age")]
proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f2), col(0).flatten(f2)]
for name, pr in (("equal counts", binop(col(0), Operator.GtEq, lit_i64(0)).flatten(f2)), ("ragged", binop(col(0), Operator.Lt, lit_i64(N2 // 3)).flatten(f2))):
    for _ in range(3):  # repeated: a missing stream synchronisation shows up as a nondeterministic mismatch
        exp = host_cols(ctx.selection_projection(w2, pr, proj))
        got = host_cols(comm.sharded_selection_projection(m2, pr, proj, gather=True))
        assert all(a.shape == b.shape and (a == b).all() for a, b in zip(exp, got)), f"gathered selection+projection differs ({name})"

# ---- Utf8 group keys travel as strings and are merged by string; nullable / Boolean / Utf8 columns are gathered
from naive_query_engine_amd import DType

NS = 30_011
names = ["alice", "bob", "", "véé", "grandmaster", "x" * 40]
sk = [names[int(i)] + (str(int(j)) if j % 4 == 0 else "") for i, j in zip(rng.integers(0, len(names), NS), rng.integers(0, 200, NS))]
sv = rng.random(NS)
svm = rng.random(NS) > 0.1
sb = rng.random(NS) < 0.5
ucols = [Column.from_list(sk, DType.UTF8), Column.from_numpy(sv, svm), Column.from_numpy(sb)]
fu = [F("s"), F("x"), F("b")]
uaggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 1), (AggregateFunc.Count, 2)]
uwhole = ctx.table_from_host(ucols)
lo3, hi3 = parallel.shard_range(NS, rank, world)
umine = ctx.table_from_host([Column.from_list(c.to_list()[lo3:hi3], c.dtype) for c in ucols])
exp, expk = ctx.aggregate(uwhole, uaggs, group_nodes=col(0).flatten(fu), with_keys=True)
got, gotk = comm.sharded_aggregate(umine, uaggs, group_nodes=col(0).flatten(fu))
def by_key(keys, res):
    ks = keys.to_host()[0].to_list()
    cs = [c.to_numpy() for c in res.to_host()]
    return {k: tuple(float(c[i]) for c in cs) for i, k in enumerate(ks)}
em, gm = by_key(expk, exp), by_key(gotk, got)
assert em.keys() == gm.keys(), "Utf8-key aggregate: key sets differ"
for k in em:
    assert em[k][0] == gm[k][0] and em[k][3] == gm[k][3] and np.allclose(em[k], gm[k], rtol=1e-9, atol=0), f"Utf8-key aggregate differs at {k!r}"
allg = comm.all_gather_table(umine).to_host()
for c_exp, c_got in zip(ucols, allg):
    assert c_exp.to_list() == c_got.to_list(), "gathered nullable / Boolean / Utf8 columns differ"

dist.barrier()
print(f"exchange checks passed on rank {rank} of {world} (RCCL {capi.Comm.rccl_version()})", flush=True)
comm.close()
dist.destroy_process_group()
