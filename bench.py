#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X, plus every BASELINE config in the same JSON line.

Headline (BASELINE.json `metric`: "rows/s + achieved HBM GB/s, 10^9-row filter→hash-agg"):

    select count(v), sum(v), avg(v), min(v), max(v) from t where id < N/2 group by id % 1024

over t(id Int64 = row number, v Float64 in [0,100)) with N = 10^9 rows PER GPU (weak scaling: rank r holds rows [r*N, (r+1)*N)
of a world*N-row table and the predicate is `id < world*N/2`), synthetic, generated on the device (SURVEY §8d generators),
HBM-resident before the timed region.  A step = one full pass of the fused filter→hash-aggregate over the rank's table, plus
(N>1) the all-gather + merge of the per-rank partial tables (nqe_sharded_aggregate_execute: RCCL on the context's stream).
Algorithmic bytes = 16 B/row (id + v each read once; no skip credit for filtered-out rows — the kernel loads unconditionally).

Launch: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one rank per device) when WORLD_SIZE is not
set, and exits non-zero when the box has fewer than N devices or WORLD_SIZE disagrees with --gpus: there is no silent 1-GPU
fallback.  Under the driver's own `python -m torch.distributed.run … bench.py --gpus N` it reads RANK/LOCAL_RANK/WORLD_SIZE.

Rank 0 prints ONE JSON line: the contract's keys for the headline, `roofline` (dominant kernel timed with HIP events on the
launch stream), `cpu_baseline` (the oracle — a C++ restatement of the reference's single-threaded algorithm — on a bounded
sample, N=1 only), `parity_checked` (the GPU result on that sample compared with the oracle's: counts exact, f64 within 1e-9)
and `configs`: the other BASELINE configs (C2, C3, C4 with build time, the random-key variants, a general hash join on sparse
keys, a many-groups aggregate; N>1: the headline without its exchange and C5 = the join range-split over the ranks with its
output all-gathered), each with its own timing, roofline block and oracle parity on a sample.  `--workload X` runs one config
as the main line; `--no-configs` skips the block (used under rocprofv3).
"""
import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable)
# kernels that do not belong to an operator's step (data generation, diagnostics)
NOT_STEP_KERNELS = ("synth_fill",)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "c2", "c3", "c4", "c4_sparse", "agg_groups"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the BASELINE size of the workload)")
    ap.add_argument("--random-keys", action="store_true", help="c3/headline: group by a random id column instead of the row number")
    ap.add_argument("--groups", type=int, default=65536, help="agg_groups: distinct keys")
    ap.add_argument("--dim-rows", type=int, default=10**6, help="c4: build-side rows")
    ap.add_argument("--pass-frac", type=float, default=0.5, help="headline: fraction of rows passing `id < K` (diagnostics; the metric uses 0.5)")
    ap.add_argument("--gather", action="store_true", help="c4 with --gpus N: also all-gather every rank's output batch in rank order (BASELINE config C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="only the main workload's line (no `configs` block)")
    ap.add_argument("--cpu-sample-rows", type=int, default=150_000_000, help="rows of the CPU baseline sample (about 10 s of single-thread work for the headline)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: one rank per device under torch.distributed.run; never returns"""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this box has {have}; refusing to run on fewer\n")
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


class F:
    def __init__(self, name):
        self.name = name


class Bench:
    def __init__(self, args, world, rank, local_rank):
        import torch
        import torch.distributed as dist

        from naive_query_engine_amd import capi, parallel

        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        self.torch, self.dist, self.capi = torch, dist, capi
        self.distributed = world > 1 or bool(os.environ.get("NQE_FORCE_EXCHANGE"))  # one rank through RCCL too (diagnostics)
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.comm = None
        if self.distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=self.dev)
            if dist.get_world_size() != world:
                sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {world}")
        self.ctx = capi.Context(local_rank)
        if self.distributed:
            self.comm = parallel.make_comm(self.ctx)  # the data path's own RCCL communicator, on the context's stream
        self.keep = []
        self.min_warm_s = 0.0

    def synth(self, kind, seed, rows, first_row=0, mod=1, base=0, dtype=None):
        """a synthetic column in a torch tensor, filled by the library's generator on the CONTEXT's stream.  Set-up code, fully
        synchronous on both sides: torch's caching allocator hands out blocks that earlier torch-stream work (the temporaries of a
        randperm, say) may still be using — safe for torch's own stream only — and torch ops that read the column later know
        nothing of the context's stream.  (Found the hard way: a 10^7-row randperm came back with duplicates.)"""
        self.torch.cuda.synchronize()
        t = self.torch.empty(rows, dtype=dtype or self.torch.int64, device=self.dev)
        self.ctx.synth_fill(kind, seed, first_row, rows, mod, base, t.data_ptr())
        self.ctx.synchronize()
        return t

    def barrier(self):
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.ctx.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if not self.distributed:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.  Kernel times come
        from HIP events the library records around every launch on its stream while timing is enabled.  The side configs
        (self.min_warm_s > 0) additionally warm up for a minimum wall time: they start right behind seconds of CPU-only work (the
        oracle), from a GPU that has clocked down."""
        t_w = time.perf_counter()
        for _ in range(warmup):
            r = step()
            del r
        if self.min_warm_s > 0:
            # the extra steps are a COUNT every rank agrees on (max over ranks): a step may hold a collective, and ranks that
            # looped on their own clocks would issue different numbers of them
            if warmup == 0:
                r = step()
                del r
            self.ctx.synchronize()
            elapsed = time.perf_counter() - t_w
            per = max(elapsed / max(warmup, 1), 1e-4)
            extra = min(2000, int(math.ceil(max(0.0, self.min_warm_s - elapsed) / per)))
            extra = int(self.max_over_ranks(float(extra)))
            for _ in range(extra):
                r = step()
                del r
        self.barrier()
        self.ctx.timing_enable(True)
        self.ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
            del r
        self.barrier()
        dt = time.perf_counter() - t0
        self.ctx.timing_enable(False)
        kernels = {kn: {"ms_per_step": ms / steps, "launches_per_step": cnt / steps} for kn, (ms, cnt) in self.ctx.timing_report().items()
                   if kn not in NOT_STEP_KERNELS}
        return self.max_over_ranks(dt) / steps * 1e3, kernels


def pick(kernels, prefixes):
    """the kernels of a step that stream the data (by name prefix), as opposed to its scans / table set-up / tails"""
    return sorted(k for k in kernels if any(k.startswith(p) for p in prefixes))


def kernel_ms(kernels, prefixes):
    return sum(kernels[k]["ms_per_step"] for k in pick(kernels, prefixes))


def roofline(algo_bytes, kernels, prefixes, extra=None):
    """achieved = SURVEY §8d algorithmic bytes of one step ÷ the summed HIP-event time of the step's data kernels"""
    kms = kernel_ms(kernels, prefixes)
    achieved = algo_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "kernel": "+".join(pick(kernels, prefixes)), "kernel_ms_per_step": kms, "algorithmic_bytes_per_step": algo_bytes,
           "kernels": kernels}
    if extra:
        out.update(extra)
    return out


def attach_traffic(roof, config_name):
    """HBM bytes per launch from this round's PMC passes (tools/profile_round.sh → profiles/r02/pmc_traffic_<config>.json;
    rocprofv3 cannot run inside the timed process).  Only when the file was made from the csrc/ tree that is running."""
    path = os.path.join(ROOT, "profiles", "r02", f"pmc_traffic_{config_name}.json")
    if not os.path.exists(path):
        return
    try:
        with open(path) as f:
            rec = json.load(f)
        from tools.csrc_rev import csrc_rev

        if rec.get("csrc_rev") != csrc_rev():
            roof["traffic_source"] = f"{os.path.relpath(path, ROOT)} is stale (made from csrc rev {rec.get('csrc_rev')}, running {csrc_rev()}): not reported"
            return
        roof["traffic"] = rec["hbm_bytes_per_step_corrected"]
        roof["traffic_ratio"] = rec["hbm_bytes_per_step_corrected"] / roof["algorithmic_bytes_per_step"] if roof["algorithmic_bytes_per_step"] else None
        roof["traffic_source"] = f"{os.path.relpath(path, ROOT)} ({rec.get('method', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE')})"
    except Exception as e:  # noqa: BLE001 - a missing/odd profile file must not fail the bench
        roof["traffic_source"] = f"unreadable {path}: {e}"


AGGS5 = None  # filled in main (needs the package)


# ------------------------------------------------------------------------------------------------ workloads
def wl_aggregate(B, rows, with_filter, random_keys, steps, warmup, groups=None, exchange=True):
    """headline / C3 / many-groups aggregate.  groups=None: key `id % 1024`; else: key = a random Int64 column in [0, groups)."""
    from naive_query_engine_amd import DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64

    n, total, first = rows, rows * B.world, B.rank * rows
    torch = B.torch
    if groups is None:
        ids = B.synth(1, 1, n, first, total, 0) if random_keys else B.synth(0, 0, n, first)
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten([F("id"), F("v")])
        kdesc = "id % 1024"
    else:
        ids = B.synth(1, 7, n, first, groups, 0)
        key = col(0).flatten([F("id"), F("v")])
        kdesc = f"k (random in [0, {groups}))"
    v = B.synth(2, 3, n, first, dtype=torch.float64)
    table = B.ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
    pred = binop(col(0), Operator.Lt, lit_i64(int(total * B.args.pass_frac))).flatten([F("id"), F("v")]) if with_filter else None

    def step():
        if B.comm is not None and exchange:
            return B.comm.sharded_aggregate(table, AGGS5, group_nodes=key, pred_nodes=pred)
        return B.ctx.aggregate(table, AGGS5, group_nodes=key, pred_nodes=pred)

    ms, kernels = B.timed(step, steps, warmup)
    desc = (f"select count(v),sum(v),avg(v),min(v),max(v) from t{' where id < N/2' if with_filter else ''} group by {kdesc}; "
            f"t(id Int64 {'random' if (random_keys or groups) else 'row number'}, v Float64), {n} rows per GPU")
    names = ["agg_grouped", "agg_partition", "agg_segments", "agg_subpartition", "agg_sample"]
    res = {"metric": "filter_hash_aggregate_rows_per_s" if with_filter else "hash_aggregate_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s",
           "ms_per_step": ms, "workload": desc, "rows_per_gpu": n, "roofline": roofline(16.0 * n, kernels, names)}
    return res, dict(table=table, ids=ids, v=v, key=key, n=n, total=total, with_filter=with_filter, random_keys=random_keys, groups=groups)


def parity_aggregate(B, st, sample_rows):
    """the same query on the first `sample_rows` rows: GPU (the same tensors, a prefix table) vs the oracle; also the cpu_baseline"""
    import numpy as np

    from naive_query_engine_amd import Column, DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    fields = [F("id"), F("v")]
    if st["groups"] is not None:
        ids = orc.synth_fill(1, 7, 0, m, st["groups"], 0).view(np.int64)
    elif st["random_keys"]:
        ids = orc.synth_fill(1, 1, 0, m, st["total"], 0).view(np.int64)
    else:
        ids = orc.synth_fill(0, 0, 0, m).view(np.int64)
    x = orc.synth_fill(2, 3, 0, m).view(np.float64)
    h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(x)]])
    plimit = st["total"] // 2 if st["random_keys"] else m // 2
    pred = binop(col(0), Operator.Lt, lit_i64(plimit)).flatten(fields) if st["with_filter"] else None
    t0 = time.perf_counter()
    ref = orc.aggregate(h, AGGS5, group_nodes=st["key"], pred_nodes=pred)[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["ids"].data_ptr(), None), (DType.FLOAT64, m, st["v"].data_ptr(), None)])
    got = B.ctx.aggregate(prefix, AGGS5, group_nodes=st["key"], pred_nodes=pred).to_host()
    g = np.stack([c.to_numpy().astype(np.float64) for c in got], axis=1)
    e = np.stack([c.to_numpy().astype(np.float64) for c in ref], axis=1)
    ok = g.shape == e.shape
    if ok:
        g, e = g[np.lexsort(g.T[::-1])], e[np.lexsort(e.T[::-1])]
        ok = bool((g[:, 0] == e[:, 0]).all() and np.allclose(g, e, rtol=1e-9, atol=0))
    cpu = {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"same query on the first {m} rows (single thread; the reference is single-threaded; host has {os.cpu_count()} cores)", "seconds": dt}
    return {"rows": m, "ok": ok, "groups": int(e.shape[0]), "tolerance": "counts exact, f64 rtol 1e-9"}, cpu


def wl_c2(B, rows, steps, warmup):
    from naive_query_engine_amd import DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64

    n, total, first = rows, rows * B.world, B.rank * rows
    ids = B.synth(0, 0, n, first)
    age = B.synth(1, 2, n, first, 60, 18)
    table = B.ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None)])
    fields = [F("id"), F("age")]
    pred = binop(col(0), Operator.Lt, lit_i64(total // 2)).flatten(fields)
    proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(fields)]
    ms, kernels = B.timed(lambda: B.ctx.selection_projection(table, pred, proj), steps, warmup)
    names = ["select_fused", "keep_from_simple", "compact_expr"]
    res = {"metric": "filter_project_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms,
           "workload": f"select age + 100 from t where id < N/2; t(id Int64, age Int64), {n} rows per GPU",
           "rows_per_gpu": n, "roofline": roofline((16.0 + 0.5 * 8.0) * n, kernels, names)}
    # the compaction does not read the source words of 4096-row tiles in which nothing was kept: with ids = row numbers the kept
    # rows are the first half, so only that half of `age` moves.  frac stays SURVEY 8d's 2.0 GB (age counted as read in full);
    # frac_physical is what this data actually moves (the PMC `traffic` of the same config shows it)
    phys = (8.0 + 0.5 * 8.0 + 0.5 * 8.0) * n
    kms = res["roofline"]["kernel_ms_per_step"]
    res["roofline"].update({"physical_bytes_per_step": phys, "frac_physical": (phys / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0,
                            "note": "frac = SURVEY 8d's 20 B/row (age counted as read in full); the compaction skips tiles without kept rows, "
                                    "so with sorted ids only the kept half of `age` is read: frac_physical = 16 B/row"})
    return res, dict(ids=ids, age=age, n=n, proj=proj, fields=fields)


def parity_c2(B, st, sample_rows):
    import numpy as np

    from naive_query_engine_amd import Column, DType, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    ids = orc.synth_fill(0, 0, 0, m).view(np.int64)
    x = orc.synth_fill(1, 2, 0, m, 60, 18).view(np.int64)
    h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(x)]])
    pred = binop(col(0), Operator.Lt, lit_i64(m // 2)).flatten(st["fields"])
    t0 = time.perf_counter()
    sel = orc.selection(h, pred, raw=True)
    ref = orc.projection(sel, st["proj"])[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["ids"].data_ptr(), None), (DType.INT64, m, st["age"].data_ptr(), None)])
    got = B.ctx.selection_projection(prefix, pred, st["proj"]).to_host()
    ok = len(got) == len(ref) == 1 and got[0].length == ref[0].length and bool((got[0].to_numpy() == ref[0].to_numpy()).all())
    cpu = {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port", "sample": f"same query on the first {m} rows (single thread)", "seconds": dt}
    return {"rows": m, "ok": ok, "tolerance": "bit-exact"}, cpu


def make_join_data(B, rows, nb, sparse, first, wide=False):
    """dim(id unique, attr) = LEFT/build, fact(key, val) = RIGHT/probe, every probe row matches once.
    dense: id = a permutation of 0..nb-1 (the direct-address PK-FK path); sparse: unique ids spread over a 2^20 x nb domain (the
    general hashed path)."""
    torch = B.torch
    g = torch.Generator(device=B.dev).manual_seed(7)
    perm = torch.randperm(nb, device=B.dev, generator=g).to(torch.int64)
    # the dim attribute: 20 bits of range (stored bit-packed in the key-ordered payload table of the dense path), or — `wide` — 62 bits
    attr = B.synth(1, 4, nb, 0, (1 << 62) if wide else (1 << 20), 0)
    fidx = B.synth(1, 5, rows, first, nb, 0)  # which dim row a fact row references
    if sparse:
        # unique by construction: key j = j * 2^20 + (u(j, 11) mod 2^20): span 2^20 x nb, far beyond the direct-address limit
        dom = (torch.arange(nb, device=B.dev, dtype=torch.int64) << 20) + B.synth(1, 11, nb, 0, 1 << 20, 0)
        dkey = dom[perm].contiguous()          # build rows in shuffled order
        fkey = dom[fidx].contiguous()
    else:
        dkey = perm
        fkey = fidx
    val = B.synth(2, 3, rows, first, dtype=torch.float64)
    torch.cuda.synchronize()
    return dkey, attr, fkey, val


def wl_c4(B, rows, nb, sparse, steps, warmup, gather=False, wide=False):
    from naive_query_engine_amd import DType

    torch = B.torch

    n, first = rows, B.rank * rows
    dkey, attr, fkey, val = make_join_data(B, n, nb, sparse, first, wide)
    dim = B.ctx.table_from_device([(DType.INT64, nb, dkey.data_ptr(), None), (DType.INT64, nb, attr.data_ptr(), None)])
    fact = B.ctx.table_from_device([(DType.INT64, n, fkey.data_ptr(), None), (DType.FLOAT64, n, val.data_ptr(), None)])
    # build (HashJoin::build, hash_join.rs:124-166): timed on its own — replicated on every rank, once per query
    build_ms, bk = B.timed(lambda: B.ctx.hash_join_build(dim, 0), max(3, steps // 2), 1)
    jt = B.ctx.hash_join_build(dim, 0)

    def probe():
        if B.comm is not None and gather:  # C5: ordered variable-length all-gather of the per-rank outputs
            return B.comm.sharded_hash_join_probe(jt, fact, 0, gather=True)
        return B.ctx.hash_join_probe(jt, fact, 0)

    ms, kernels = B.timed(probe, steps, warmup)
    names = ["join_probe", "join_fused_write", "compact_gather", "compact_column"]
    total = n * B.world
    algo = 48.0 * n + 16.0 * nb     # 16 B/probe row read + 32 B/output row written + the build side once
    phys = 40.0 * n + 16.0 * nb     # the output's two key columns are one shared buffer: 24 B/row are written
    kms = kernel_ms(kernels, names)
    extra = {"probe_kernels_ms": kms, "build_ms": build_ms, "probe_ms": ms, "execute_ms": ms + build_ms,
             "frac_physical": (phys / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms > 0 else 0.0, "physical_bytes_per_step": phys,
             "frac_end_to_end": algo / ((ms + build_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "frac = SURVEY 8d's 48 B/probe row over the probe kernels; frac_physical = the 40 B/row that move (shared key column); "
                     "frac_end_to_end = 8d bytes over build + probe wall time (HashJoin::execute)"}
    gather_check = None
    if B.comm is not None and gather:
        # no oracle at this size: size-independent properties of the GATHERED output on every rank — every probe row matches once,
        # so the output has n x world rows in rank order: this rank's slice of the fact-key column is its own fact keys, the whole
        # column sums to the sum of every rank's keys, and the dim key column equals the fact key column row by row
        from naive_query_engine_amd.parallel import table_columns_as_tensors

        out_t = probe()
        B.ctx.synchronize()
        ok = out_t.num_rows == n * B.world and out_t.num_columns == 4
        if ok:
            cols_t = table_columns_as_tensors(out_t, B.dev)
            mine = cols_t[2][B.rank * n:(B.rank + 1) * n]
            ok = bool(torch.equal(mine, fkey)) and bool(torch.equal(cols_t[0], cols_t[2]))
            sums = torch.stack([fkey.sum(), val.view(torch.int64).sum()])
            if B.distributed:
                B.dist.all_reduce(sums)  # int64 sums wrap identically everywhere
            ok = ok and int(cols_t[2].sum()) == int(sums[0]) and int(cols_t[3].sum()) == int(sums[1])
            del cols_t, mine
        okt = torch.tensor([1 if ok else 0], device=B.dev)
        if B.distributed:
            B.dist.all_reduce(okt, op=B.dist.ReduceOp.MIN)
        gather_check = {"ok": bool(int(okt.item())), "rows": int(out_t.num_rows),
                        "what": "gathered join output on every rank: n x world rows, own slice == own fact keys, dim key == fact key, column sums == all-rank sums"}
        del out_t
        if not gather_check["ok"]:
            sys.stderr.write("bench.py: the gathered join output failed its check\n")
            sys.exit(3)
    res = {"metric": "hash_join_probe_rows_per_s", "value": total / (ms * 1e-3), "unit": "rows/s", "ms_per_step": ms,
           "workload": f"dim(id,attr{' of 62-bit range' if wide else ' of 20-bit range'}) {nb} rows ({'sparse 2^40-domain' if sparse else 'dense'} unique keys, LEFT/build) join fact(key,val) {n} rows per GPU "
                       f"(RIGHT/probe), 1 match per probe row; 4 output columns{'; outputs all-gathered in rank order' if gather else ''}",
           "rows_per_gpu": n, "build_rows": nb, "roofline": roofline(algo, kernels, names, extra)}
    if gather_check:
        res["gather_check"] = gather_check
    return res, dict(dim=dim, jt=jt, dkey=dkey, attr=attr, fkey=fkey, val=val, n=n, nb=nb, fact=fact)


def parity_c4(B, st, sample_rows):
    import numpy as np

    from naive_query_engine_amd import Column, DType
    from oracle import oracle as orc

    m = min(st["n"], sample_rows)
    left = [Column.from_numpy(st["dkey"].cpu().numpy()), Column.from_numpy(st["attr"].cpu().numpy())]
    right = [Column.from_numpy(st["fkey"][:m].cpu().numpy()), Column.from_numpy(st["val"][:m].cpu().numpy())]
    t0 = time.perf_counter()
    ref = orc.hash_join([left], [right], 0, 0)[0]
    dt = time.perf_counter() - t0
    prefix = B.ctx.table_from_device([(DType.INT64, m, st["fkey"].data_ptr(), None), (DType.FLOAT64, m, st["val"].data_ptr(), None)])
    got = B.ctx.hash_join_probe(st["jt"], prefix, 0).to_host()
    ok = len(got) == len(ref) and all(g.length == r.length and bool((g.to_numpy().view(np.int64) == r.to_numpy().view(np.int64)).all()) for g, r in zip(got, ref))
    cpu = {"value": m / dt, "unit": "probe rows/s", "cores": 1, "kind": "port",
           "sample": f"HashJoin build ({st['nb']} rows) + probe of the first {m} fact rows (single thread)", "seconds": dt}
    return {"rows": m, "ok": ok, "output_rows": int(ref[0].length), "tolerance": "bit-exact, row order included"}, cpu


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a different configuration than asked\n")
        sys.exit(2)
    import torch

    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py: rank {rank} needs device {local_rank}, this box has {torch.cuda.device_count()}\n")
        sys.exit(2)

    from naive_query_engine_amd import AggregateFunc

    global AGGS5
    AGGS5 = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
    B = Bench(args, world, rank, local_rank)
    default_rows = {"headline": 10**9, "c3": 10**9, "c2": 10**8, "c4": 10**8, "c4_sparse": 10**8, "agg_groups": 10**8}[args.workload]
    n = args.rows or default_rows
    want_cpu = world == 1 and not args.no_cpu_baseline
    csteps, cwarm = max(3, min(args.steps, 10)), 2  # the side configs: a few steps each

    # ---- the main line
    if args.workload in ("headline", "c3"):
        res, st = wl_aggregate(B, n, args.workload == "headline", args.random_keys, args.steps, args.warmup)
        par = parity_aggregate(B, st, args.cpu_sample_rows) if want_cpu else None
        name = args.workload + ("_random_keys" if args.random_keys else "")
    elif args.workload == "agg_groups":
        res, st = wl_aggregate(B, n, False, False, args.steps, args.warmup, groups=args.groups)
        par = parity_aggregate(B, st, min(args.cpu_sample_rows, 20_000_000)) if want_cpu else None
        name = f"agg_{args.groups}_groups"
    elif args.workload == "c2":
        res, st = wl_c2(B, n, args.steps, args.warmup)
        par = parity_c2(B, st, args.cpu_sample_rows) if want_cpu else None
        name = "c2"
    else:
        sparse = args.workload == "c4_sparse"
        res, st = wl_c4(B, n, args.dim_rows, sparse, args.steps, args.warmup, gather=args.gather)
        par = parity_c4(B, st, 5_000_000) if want_cpu else None
        name = args.workload
    attach_traffic(res["roofline"], name)
    n_main = res["rows_per_gpu"]
    out = {
        "metric": res["metric"], "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": res["workload"], "rows_per_gpu": n_main, "total_rows": n_main * world, "parallelism": f"row-range x{world}"},
        "roofline": res["roofline"],
    }
    if B.distributed:
        out["rccl_ranks"] = B.dist.get_world_size()
        out["rccl_version"] = B.capi.Comm.rccl_version()
        out["exchange"] = "nqe_sharded_* (C ABI) on RCCL, collectives on the context's stream"
    if par:
        out["parity_checked"], out["cpu_baseline"] = par
    if B.comm is not None and args.workload == "headline" and not args.random_keys and args.pass_frac == 0.5:
        # no oracle at this size: a size-independent check of the SHARDED result on every rank — ids are row numbers, so group g of
        # `id % 1024` holds exactly the ids g, g + 1024, ... below total/2, and every value lies in [0, 100)
        import numpy as np

        from naive_query_engine_amd import Operator
        from naive_query_engine_amd.expression import binop, col, lit_i64

        pred = binop(col(0), Operator.Lt, lit_i64(st["total"] // 2)).flatten([F("id"), F("v")])
        res, keys = B.comm.sharded_aggregate(st["table"], AGGS5, group_nodes=st["key"], pred_nodes=pred)
        cols = [c.to_numpy() for c in res.to_host()]
        kk = keys.to_host()[0].to_numpy()
        half = st["total"] // 2
        exp_cnt = np.array([(half - g + 1023) // 1024 if g < half else 0 for g in range(1024)], dtype=np.uint64)
        ok = bool(len(kk) == 1024 and (kk == np.arange(1024)).all() and (cols[0] == exp_cnt).all() and (cols[3] >= 0).all() and (cols[4] < 100).all()
                  and np.allclose(cols[2], cols[1] / cols[0].astype(np.float64), rtol=1e-12))
        okt = B.torch.tensor([1 if ok else 0], dtype=B.torch.int64, device=B.dev)
        B.dist.all_reduce(okt, op=B.dist.ReduceOp.MIN)
        out["result_check"] = {"ok": bool(int(okt.item())), "what": "sharded headline on every rank: 1024 keys, analytic counts, avg = sum / count, min/max in [0, 100)"}
    del st

    # ---- every other config, in the same line
    if args.workload == "headline" and not args.no_configs and not args.random_keys:
        cfg = {}

        B.min_warm_s = 0.15

        def add(cname, fn, parity=None):
            # (no allocator trimming between configs: memory handed back to the driver and mapped again came back slower — C3 over
            # re-allocated columns ran 8-10 % below the same kernel over the process's first allocations)
            r, s = fn()
            attach_traffic(r["roofline"], cname)
            if parity and want_cpu:
                r["parity_checked"], r["cpu_baseline"] = parity(s)
            del s
            cfg[cname] = r

        # (NQE_BENCH_MULTI_CONFIGS with NQE_FORCE_EXCHANGE: the multi-rank block on one rank through RCCL — a dry run of that code)
        if world == 1 and not (B.distributed and os.environ.get("NQE_BENCH_MULTI_CONFIGS")):
            add("c3", lambda: wl_aggregate(B, n, False, False, csteps, cwarm), lambda s: parity_aggregate(B, s, 20_000_000))
            add("headline_random_keys", lambda: wl_aggregate(B, n, True, True, csteps, cwarm), lambda s: parity_aggregate(B, s, 20_000_000))
            add("c3_random_keys", lambda: wl_aggregate(B, n, False, True, csteps, cwarm))
            add("c2", lambda: wl_c2(B, 10**8, csteps, cwarm), lambda s: parity_c2(B, s, 20_000_000))
            add("c4", lambda: wl_c4(B, 10**8, 10**6, False, csteps, cwarm), lambda s: parity_c4(B, s, 5_000_000))
            add("c4_wide_payload", lambda: wl_c4(B, 10**8, 10**6, False, csteps, cwarm, wide=True))  # attr spans 2^62: an 8 MB payload table
            add("c4_dim_1e7", lambda: wl_c4(B, 10**8, 10**7, False, csteps, cwarm))
            add("c4_sparse_keys", lambda: wl_c4(B, 10**8, 10**6, True, csteps, cwarm), lambda s: parity_c4(B, s, 5_000_000))
            for G in (4096, 65536, 1 << 20):
                add(f"agg_{G}_groups", lambda G=G: wl_aggregate(B, 10**8, False, False, csteps, cwarm, groups=G),
                    (lambda s: parity_aggregate(B, s, 10_000_000)) if G == 65536 else None)
        else:
            # the headline without its exchange (every rank aggregates its shard only): step time with and without
            add("headline_local_only", lambda: wl_aggregate(B, n, True, False, csteps, cwarm, exchange=False))
            out["exchange_ms_per_step"] = out["ms_per_step"] - cfg["headline_local_only"]["ms_per_step"]
            # C5: the C4 join strong-scaled — build replicated, 10^8 fact rows range-split over the ranks
            shard = 10**8 // world
            add("c5_probe_only", lambda: wl_c4(B, shard, 10**6, False, csteps, cwarm, gather=False))
            add("c5_probe_and_gather", lambda: wl_c4(B, shard, 10**6, False, csteps, cwarm, gather=True))
            p, g = cfg["c5_probe_only"], cfg["c5_probe_and_gather"]
            gather_ms = g["ms_per_step"] - p["ms_per_step"]
            out_bytes = 24.0 * shard * world  # three distinct 8-byte output columns per row (the shared key column travels once)
            inbound = out_bytes * (world - 1) / world
            cfg["c5"] = {"workload": f"C4 with the probe side range-split over {world} GPUs ({shard} fact rows each), build replicated; output all-gathered on every rank",
                         "probe_only_ms": p["ms_per_step"], "gather_ms": gather_ms, "end_to_end_ms": g["ms_per_step"],
                         "gathered_bytes_per_rank_inbound": inbound, "xgmi_GBps_per_gpu_inbound": inbound / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else None,
                         "probe_rows_per_s_all_gpus": 10**8 / (p["ms_per_step"] * 1e-3)}
        out["configs"] = cfg

    # RCCL writes a version banner through C stdio, which (redirected) is flushed at process exit — after Python's own output.
    # Everything buffered so far goes out on every rank first, so that rank 0's JSON line is the LAST line of the job's stdout.
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if B.distributed:
        B.dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if par and not par[0]["ok"]:
        sys.stderr.write("bench.py: PARITY FAILURE against the oracle on the sample\n")
        sys.exit(3)
    if "result_check" in out and not out["result_check"]["ok"]:
        sys.stderr.write("bench.py: the sharded result failed its analytic check\n")
        sys.exit(3)
    if "configs" in out and any(not c.get("parity_checked", {"ok": True})["ok"] for c in out["configs"].values()):
        sys.stderr.write("bench.py: PARITY FAILURE against the oracle in a side config\n")
        sys.exit(3)
    if B.distributed:
        B.dist.barrier()
        if B.comm is not None:
            B.comm.close()
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
