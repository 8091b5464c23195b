"""GPU, 2 ranks sharing cuda:0 (gloo rendezvous; RCCL needs one device per rank, the driver's 8-GPU
run covers that): the sharded aggregate = per-rank nqe_aggregate_partial → ordered all-gather →
nqe_aggregate_merge on every rank must equal the single-GPU result and the oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 200_000


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_cols():
    from naive_query_engine_amd import Column

    rng = np.random.default_rng(11)
    ids = np.arange(N, dtype=np.int64)
    v = rng.random(N) * 100.0
    mask = rng.random(N) > 0.05
    return [Column.from_numpy(ids), Column.from_numpy(v, mask)]


def plan(mod=128):
    from naive_query_engine_amd import AggregateFunc, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from tests.helpers import fields

    f = fields("id", "v")
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
    if mod == "disjoint":
        # 8192 groups, the two ranks hold 4096 each and none in common: every partial fits the exchange buffer, their union
        # does not fit the merge's small first-attempt table (TABLE_FULL → retry with the worst-case table)
        return aggs, binop(col(0), Operator.Divide, lit_i64(N // 8192 + 1)).flatten(f), binop(col(0), Operator.GtEq, lit_i64(0)).flatten(f)
    key = binop(col(0), Operator.Modulos, lit_i64(mod)).flatten(f) if mod else None
    return aggs, key, binop(col(0), Operator.Lt, lit_i64(N // 2)).flatten(f)


def worker(rank, world, port, q, mod):
    import torch
    import torch.distributed as dist

    from naive_query_engine_amd import Column, capi
    from naive_query_engine_amd.parallel import shard_range, sharded_aggregate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        cols = make_cols()
        lo, hi = shard_range(N, rank, world)
        sub = [Column.from_numpy(c.to_numpy()[lo:hi], c.valid_mask()[lo:hi]) for c in cols]
        aggs, key, pred = plan(mod)
        out, keys = sharded_aggregate(ctx, ctx.table_from_host(sub), aggs, group_nodes=key, pred_nodes=pred)
        res = np.stack([c.to_numpy().astype(np.float64) for c in out.to_host()], axis=1)
        q.put((rank, keys.to_host()[0].to_numpy().tolist() if keys is not None else None, res.tolist()))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mod", [128, 6000, None, "disjoint"])  # one-collective exchange / more groups than the exchange buffer / un-grouped / union overflows the small merge table
def test_sharded_aggregate_two_ranks_one_gpu(mod):
    import torch.multiprocessing as mp

    from naive_query_engine_amd import capi
    from oracle import oracle as orc

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = free_port()
    procs = [mpc.Process(target=worker, args=(r, 2, port, q, mod)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cols = make_cols()
    aggs, key, pred = plan(mod)
    exp = orc.aggregate([cols], aggs, group_nodes=key, pred_nodes=pred)[0]
    exp_m = np.stack([c.to_numpy().astype(np.float64) for c in exp], axis=1)
    exp_m = exp_m[np.lexsort(exp_m.T[::-1])]
    ctx = capi.Context(0)
    single = np.stack([c.to_numpy().astype(np.float64) for c in ctx.aggregate(ctx.table_from_host(cols), aggs, group_nodes=key, pred_nodes=pred).to_host()], axis=1)
    for rank, keys, res in results:
        nkeys = (N - 1) // (N // 8192 + 1) + 1 if mod == "disjoint" else mod
        assert keys == (list(range(nkeys)) if mod else None)
        got = np.array(res)
        assert np.allclose(got, single, rtol=1e-9, atol=0)          # rows are key-sorted on both sides
        g = got[np.lexsort(got.T[::-1])]
        assert (g[:, 0] == exp_m[:, 0]).all() and np.allclose(g, exp_m, rtol=1e-9, atol=0)
    ctx.close()


def join_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from naive_query_engine_amd import Column, capi
    from naive_query_engine_amd.parallel import shard_range, sharded_hash_join

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        left, right = join_data()
        lo, hi = shard_range(right[0].length, rank, world)
        sub = [Column.from_numpy(c.to_numpy()[lo:hi]) for c in right]
        out = sharded_hash_join(ctx, ctx.table_from_host(left), ctx.table_from_host(sub), 0, 0, gather=True)
        q.put((rank, [c.to_numpy().view(np.int64).tolist() for c in out.to_host()]))
        ctx.close()
    finally:
        dist.destroy_process_group()


def join_data():
    from naive_query_engine_amd import Column

    rng = np.random.default_rng(3)
    nb, npr = 3000, 10001
    left = [Column.from_numpy(rng.permutation(nb).astype(np.int64)), Column.from_numpy(rng.integers(0, 99, nb).astype(np.int64))]
    right = [Column.from_numpy(rng.integers(-3, nb + 3, npr).astype(np.int64)), Column.from_numpy(rng.random(npr))]
    return left, right


@pytest.mark.timeout(300)
def test_sharded_hash_join_two_ranks_keeps_probe_order():
    import torch.multiprocessing as mp

    from oracle import oracle as orc

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = free_port()
    procs = [mpc.Process(target=join_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    left, right = join_data()
    exp = [c.to_numpy().view(np.int64).tolist() for c in orc.hash_join([left], [right], 0, 0)[0]]
    for rank, got in results:
        assert got == exp  # build replicated, probe range-split, rank order == probe order


@pytest.mark.timeout(300)
def test_single_rank_rccl_exchange_paths():
    """real RCCL (backend "nccl") needs one device per rank, so on a 1-GPU box the collectives are exercised with ONE rank
    and NQE_FORCE_EXCHANGE=1: the packed one-collective aggregate exchange and the equal-count zero-copy table gather must
    reproduce the local results (tools/check_gather_nccl.py runs in its own process: torch initialises the device first)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(free_port()))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_gather_nccl.py")], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "nccl single-rank exchange checks passed" in out.stdout
