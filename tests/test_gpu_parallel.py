"""GPU: the sharded operators of the C ABI (nqe_sharded_*, csrc/exchange.hip).

* 2 ranks sharing cuda:0: RCCL needs one device per rank, so the two ranks drive the SAME C++ sharding code through the host-staged
  transport (parallel.HostStagedTransport over gloo, plugged in with nqe_comm_create_custom) — sharded aggregate / join /
  filter+projection must equal the single-GPU result and the oracle;
* real RCCL (backend "nccl"): tools/check_exchange.py with one rank on a 1-GPU box, and with one rank per device under
  torch.distributed.run whenever the box has two or more devices."""
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
    from naive_query_engine_amd.parallel import make_staged_comm, shard_range, sharded_aggregate

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
        comm = make_staged_comm(ctx)
        out, keys = sharded_aggregate(comm, ctx.table_from_host(sub), aggs, group_nodes=key, pred_nodes=pred)
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
    from naive_query_engine_amd.parallel import make_staged_comm, shard_range, sharded_hash_join, sharded_selection_projection

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        left, right = join_data()
        lo, hi = shard_range(right[0].length, rank, world)
        sub = [Column.from_numpy(c.to_numpy()[lo:hi]) for c in right]
        comm = make_staged_comm(ctx)
        out = sharded_hash_join(comm, ctx.table_from_host(left), ctx.table_from_host(sub), 0, 0, gather=True)
        # filter + projection over the same shard, gathered (ragged counts)
        from naive_query_engine_amd import Operator
        from naive_query_engine_amd.expression import binop, col, lit_i64
        from tests.helpers import fields

        f = fields("k", "v")
        sp = sharded_selection_projection(comm, ctx.table_from_host(sub), binop(col(0), Operator.Lt, lit_i64(700)).flatten(f),
                                          [binop(col(0), Operator.Plus, lit_i64(100)).flatten(f)], gather=True)
        q.put((rank, [c.to_numpy().view(np.int64).tolist() for c in out.to_host()], sp.to_host()[0].to_numpy().tolist()))
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
    k = right[0].to_numpy()
    exp_sp = (k[k < 700] + 100).tolist()
    for rank, got, sp in results:
        assert got == exp  # build replicated, probe range-split, rank order == probe order
        assert sp == exp_sp


def wide_worker(rank, world, port, q):
    """nullable + Boolean + Utf8 columns through the exchange: README query 2's shape (Utf8 payloads on both join sides) sharded, a
    gathered selection over every column type, and a group-by over a Utf8 key"""
    import torch
    import torch.distributed as dist

    from naive_query_engine_amd import AggregateFunc, Column, DType, Operator, capi
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from naive_query_engine_amd.parallel import make_staged_comm, shard_range, sharded_aggregate, sharded_hash_join, sharded_selection_projection
    from tests.helpers import fields

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        left, right = wide_data()
        lo, hi = shard_range(right[0].length, rank, world)
        sub = [Column.from_list(c.to_list()[lo:hi], c.dtype) for c in right]  # (from_list drops a validity bitmap without nulls)
        comm = make_staged_comm(ctx)
        t = ctx.table_from_host(sub)
        out = sharded_hash_join(comm, ctx.table_from_host(left), t, 0, 0, gather=True)
        f = fields("k", "v", "name", "flag")
        sp = sharded_selection_projection(comm, t, binop(col(0), Operator.Lt, lit_i64(40)).flatten(f),
                                          [col(2).flatten(f), col(1).flatten(f), col(3).flatten(f), binop(col(0), Operator.Plus, lit_i64(1)).flatten(f)], gather=True)
        aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 1), (AggregateFunc.Count, 3)]
        res, keys = sharded_aggregate(comm, t, aggs, group_nodes=col(2).flatten(f))
        pack = lambda cols: [(c.dtype.value, c.to_list()) for c in cols]
        q.put((rank, pack(out.to_host()), pack(sp.to_host()), pack(keys.to_host()), pack(res.to_host())))
        ctx.close()
    finally:
        dist.destroy_process_group()


def wide_data():
    from naive_query_engine_amd import Column, DType
    from tests.helpers import random_utf8

    rng = np.random.default_rng(5)
    nb, npr = 64, 5003
    left = [Column.from_numpy(rng.permutation(nb).astype(np.int64)), random_utf8(rng, nb, null_frac=0.2),
            Column.from_numpy(rng.random(nb), rng.random(nb) > 0.3)]
    # nulls only in the first third of the probe rows: the second rank's shard has no validity bitmap at all
    third = npr // 3
    vmask = np.concatenate([rng.random(third) > 0.2, np.ones(npr - third, dtype=bool)])
    right = [Column.from_numpy(rng.integers(-2, nb + 2, npr).astype(np.int64)), Column.from_numpy(rng.random(npr) * 10, vmask),
             random_utf8(rng, npr, null_frac=0.0), Column.from_numpy(rng.random(npr) < 0.5, np.concatenate([rng.random(third) > 0.1, np.ones(npr - third, dtype=bool)]))]
    return left, right


@pytest.mark.timeout(300)
def test_sharded_operators_move_nullable_boolean_and_utf8_columns():
    import torch.multiprocessing as mp

    from naive_query_engine_amd import AggregateFunc, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from oracle import oracle as orc
    from tests.helpers import fields

    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = free_port()
    procs = [mpc.Process(target=wide_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    left, right = wide_data()
    pack = lambda cols: [(c.dtype.value, c.to_list()) for c in cols]
    exp_join = pack(orc.hash_join([left], [right], 0, 0)[0])
    f = fields("k", "v", "name", "flag")
    sel = orc.selection([right], binop(col(0), Operator.Lt, lit_i64(40)).flatten(f), raw=True)
    exp_sp = pack(orc.projection(sel, [col(2).flatten(f), col(1).flatten(f), col(3).flatten(f), binop(col(0), Operator.Plus, lit_i64(1)).flatten(f)])[0])
    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Max, 1), (AggregateFunc.Count, 3)]
    # the oracle has no key output: expected rows by key string, computed on the host
    names, v, flag = right[2].to_list(), right[1].to_list(), right[3].to_list()
    exp = {}
    for s, x, b in zip(names, v, flag):
        e = exp.setdefault(s, [0, 0.0, -np.finfo(np.float64).max, 0])
        if x is not None:
            e[0] += 1
            e[1] += x
            e[2] = max(e[2], x)
        if b is not None:
            e[3] += 1
    ref_rows = orc.aggregate([right], aggs, group_nodes=col(2).flatten(f))[0]
    assert ref_rows[0].length == len(exp)  # the oracle agrees on the number of groups
    for rank, got_join, got_sp, got_keys, got_res in results:
        assert got_join == exp_join      # Utf8 + nullable payloads of both sides, probe order across the ranks
        assert got_sp == exp_sp          # Utf8, nullable Float64 (validity on rank 0 only), nullable Boolean, computed Int64
        keys = got_keys[0][1]
        assert sorted(keys) == sorted(exp) and len(set(keys)) == len(keys)
        for i, s in enumerate(keys):
            cnt, sm, mx, cb = (got_res[c][1][i] for c in range(4))
            assert cnt == exp[s][0] and cb == exp[s][3] and mx == exp[s][2] and abs(sm - exp[s][1]) <= 1e-9 * abs(exp[s][1])


FABRIC_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_exchange_fabric")


def build_fabric_exe():
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "cpp", "test_exchange_fabric.cpp")
    libdir = os.path.join(root, "naive_query_engine_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", FABRIC_EXE, f"-L{libdir}", "-lnqe_hip",
                           f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    return FABRIC_EXE


@pytest.mark.timeout(900)
def test_exchange_over_point_to_point_fabric_2_3_8_ranks():
    """tests/cpp/test_exchange_fabric.cpp: ranks as threads over a send/recv test fabric with RCCL's matching rules — the library's
    own p2p_all_gather_v (what runs over ncclSend/ncclRecv) with world = 2, 3, 8; every column type; collective failure"""
    import subprocess

    exe = build_fabric_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=850)
    print(out.stdout[-6000:], out.stderr[-3000:])
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-3000:]
    assert "FAIL" not in out.stdout and "checks passed" in out.stdout


def _run_check_exchange(nranks):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "check_exchange.py")
    env = dict(os.environ, MASTER_PORT=str(free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if nranks == 1:
        cmd = [sys.executable, tool]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", env["MASTER_PORT"], tool]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("exchange checks passed") == nranks, out.stdout[-2000:]


@pytest.mark.timeout(600)
def test_single_rank_rccl_exchange_paths():
    """real RCCL needs one device per rank, so a 1-GPU box exercises nqe_comm_create / ncclAllGather / the sharded entry points
    with ONE rank (tools/check_exchange.py runs in its own process: torch initialises the device first)"""
    _run_check_exchange(1)


@pytest.mark.timeout(600)
def test_multi_device_rccl_exchange():
    """one rank per device over RCCL (the peer-to-peer all_gather_v included) whenever the box has at least two devices"""
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two or more GPUs (the driver's multi-GPU run)")
    _run_check_exchange(min(n, 8))


@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` launches N ranks itself and must fail loudly — never fall back to one GPU — when the box has
    fewer devices"""
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=280, env=env)
    assert out.returncode != 0
    assert "n_gpus" not in out.stdout


def test_bench_multi_rank_block_dry_run_on_one_rank():
    """the part of bench.py that only runs with more than one rank (headline with and without its exchange, C5 probe / gather /
    end to end), driven on ONE rank through real RCCL: same code, same JSON keys as an N-GPU run of the driver"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(NQE_FORCE_EXCHANGE="1", NQE_BENCH_MULTI_CONFIGS="1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    d = json.loads(line)  # the JSON line is the last line of stdout
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["rccl_version"] > 0 and d["result_check"]["ok"]
    assert {"headline_local_only", "c5_probe_only", "c5_probe_and_gather", "c5"} <= set(d["configs"])
    assert d["configs"]["c5_probe_and_gather"]["gather_check"]["ok"]
    c5 = d["configs"]["c5"]
    assert c5["end_to_end_ms"] >= c5["probe_only_ms"] > 0 and "exchange_ms_per_step" in d


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_through_the_host_transport():
    """`bench.py --gpus 2` on a ONE-GPU box: NQE_BENCH_TRANSPORT=host lets the two ranks share the device and exchange through host
    memory (gloo), so the multi-rank blocks of the bench — the sharded headline with its analytic result check, the headline without
    its exchange, C5 with gather = 0 as its headline and the gathered form beside it — execute with world > 1, at reduced rows.
    A functional run, not a scaling measurement (the line says so)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(NQE_BENCH_TRANSPORT="host", NQE_BENCH_C5_ROWS="20000000")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "20000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=850, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and "HOST-STAGED" in d["exchange"] and d["scaling"] == "weak"
    assert d["config"]["total_rows"] == 40000000 and d["result_check"]["ok"]          # the sharded headline on every rank: analytic counts
    cfg = d["configs"]
    assert {"headline_local_only", "c5_probe_only", "c5_probe_and_gather", "c5"} <= set(cfg)
    assert "exchange_ms_per_step" in d and cfg["headline_local_only"]["ms"] > 0
    c5 = cfg["c5"]
    assert c5["gather"] == 0 and c5["fact_rows_per_gpu"] == 10000000 and c5["probe_only_ms"] > 0 and c5["ms"] == c5["probe_only_ms"]
    assert c5["gather_ms"] is not None and c5["xgmi_GBps_per_gpu_inbound"] is not None and c5["end_to_end_ms"] >= c5["probe_only_ms"]
    assert cfg["c5_probe_and_gather"]["gather_check"]["ok"] and cfg["c5_probe_and_gather"]["gather_check"]["rows"] == 20000000
