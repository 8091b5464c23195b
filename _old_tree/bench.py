#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

Workload (BASELINE.json `metric`: "rows/s + achieved HBM GB/s, 10^9-row filter→hash-agg"):

    select count(v), sum(v), avg(v), min(v), max(v) from t where id < N/2 group by id % 1024

over t(id Int64 = row number, v Float64 in [0,100)) with N = 10^9 rows PER GPU (weak scaling: rank r
holds rows [r*N, (r+1)*N) of a world*N-row table and the predicate is `id < world*N/2`), synthetic,
generated on the device (SURVEY §8d generators), HBM-resident before the timed region.
A step = one full pass of the fused filter→hash-aggregate over the rank's table, plus (N>1) the
all-gather + merge of the per-rank partial tables.  Algorithmic bytes = 16 B/row (id + v each read
once; no skip credit for filtered-out rows — the kernel loads unconditionally).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel
`agg_grouped`, timed with HIP events on the launch stream) and `cpu_baseline` (the oracle — a C++
restatement of the reference's single-threaded algorithm — on a bounded sample, rank 0, N=1 only).

Other BASELINE configs: --workload c2 (filter+project, 10^8 rows), c3 (hash-agg without filter),
c4 (hash join 10^8 ⋈ 10^6).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "c2", "c3", "c4"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: the BASELINE size of the workload)")
    ap.add_argument("--random-keys", action="store_true", help="c3/headline: group by a random id column instead of the row number")
    ap.add_argument("--pass-frac", type=float, default=0.5, help="headline: fraction of rows passing `id < K` (diagnostics; the metric uses 0.5)")
    ap.add_argument("--gather", action="store_true", help="c4 with --gpus N: also all-gather every rank's output batch in rank order (BASELINE config C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=150_000_000, help="rows of the CPU baseline sample (about 10 s of single-thread work for the headline)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import numpy as np

    from naive_query_engine_amd import AggregateFunc, DType, Operator, capi
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from naive_query_engine_amd.parallel import sharded_aggregate, sharded_hash_join

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = bool(os.environ.get("NQE_FORCE_EXCHANGE"))  # one rank, but through RCCL and the exchange path (diagnostics)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    if world > 1 or force_dist:
        # one stream for the kernels and for torch's collectives' dependencies: the exchange needs no host synchronisation
        # between pack, all-gather and merge
        ts = torch.cuda.Stream(dev)
        torch.cuda.set_stream(ts)
        ctx = capi.Context(local_rank, stream=ts.cuda_stream)
    else:
        ctx = capi.Context(local_rank)

    default_rows = {"headline": 10**9, "c3": 10**9, "c2": 10**8, "c4": 10**8}[args.workload]
    n = args.rows or default_rows
    total = n * world
    first = rank * n

    class F:
        def __init__(self, name):
            self.name = name

    def synth(kind, seed, rows, first_row=0, mod=1, base=0, dtype=torch.int64):
        t = torch.empty(rows, dtype=dtype, device=dev)
        ctx.synth_fill(kind, seed, first_row, rows, mod, base, t.data_ptr())
        return t

    aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
    keep = []  # torch tensors backing the tables
    if args.workload in ("headline", "c3"):
        ids = synth(1, 1, n, first, total, 0) if args.random_keys else synth(0, 0, n, first)
        v = synth(2, 3, n, first, dtype=torch.float64)
        keep += [ids, v]
        table = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
        fields = [F("id"), F("v")]
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(fields)
        pred = binop(col(0), Operator.Lt, lit_i64(int(total * args.pass_frac))).flatten(fields) if args.workload == "headline" else None
        algo_bytes_per_row = 16.0
        kernel_name = "agg_grouped"

        def step():
            if world > 1 or force_dist:
                return sharded_aggregate(ctx, table, aggs, group_nodes=key, pred_nodes=pred)
            return ctx.aggregate(table, aggs, group_nodes=key, pred_nodes=pred)

        metric = "filter_hash_aggregate_rows_per_s" if args.workload == "headline" else "hash_aggregate_rows_per_s"
        desc = (f"select count(v),sum(v),avg(v),min(v),max(v) from t{' where id < N/2' if pred else ''} group by id % 1024; "
                f"t(id Int64 {'random' if args.random_keys else 'row number'}, v Float64), {n} rows per GPU")
    elif args.workload == "c2":
        ids = synth(0, 0, n, first)
        age = synth(1, 2, n, first, 60, 18)
        keep += [ids, age]
        table = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None)])
        fields = [F("id"), F("age")]
        pred = binop(col(0), Operator.Lt, lit_i64(total // 2)).flatten(fields)
        proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(fields)]
        algo_bytes_per_row = 16.0 + 0.5 * 8.0  # selectivity 0.5 on one GPU
        kernel_name = "keep_from_simple+compact_expr"

        def step():
            return ctx.selection_projection(table, pred, proj)

        metric = "filter_project_rows_per_s"
        desc = f"select age + 100 from t where id < N/2; t(id Int64, age Int64), {n} rows per GPU"
    else:  # c4
        nb = 10**6
        perm = torch.randperm(nb, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(torch.int64)
        attr = synth(1, 4, nb, 0, 1 << 20, 0)
        fkey = synth(1, 5, n, first, nb, 0)
        val = synth(2, 3, n, first, dtype=torch.float64)
        keep += [perm, attr, fkey, val]
        dim = ctx.table_from_device([(DType.INT64, nb, perm.data_ptr(), None), (DType.INT64, nb, attr.data_ptr(), None)])
        fact = ctx.table_from_device([(DType.INT64, n, fkey.data_ptr(), None), (DType.FLOAT64, n, val.data_ptr(), None)])
        jt = ctx.hash_join_build(dim, 0)  # build replicated on every rank, outside the timed probe
        algo_bytes_per_row = 16.0 + 32.0
        kernel_name = "join_probe+join_fused+compact_gather+compact_column"

        def step():
            if (world > 1 or force_dist) and args.gather:  # C5: ordered variable-length all-gather of the per-rank outputs over RCCL
                return sharded_hash_join(ctx, dim, fact, 0, 0, gather=True, join_table=jt)
            return ctx.hash_join_probe(jt, fact, 0)

        metric = "hash_join_probe_rows_per_s"
        desc = (f"dim(id,attr) 10^6 rows (LEFT/build) join fact(key,val) {n} rows per GPU (RIGHT/probe), 1 match per probe row; 4 output columns "
                "(SURVEY 8d: 16 B read + 32 B written per probe row; the output's two key columns are one shared buffer, so 24 B are physically written)")

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        r = step()
        del r
    barrier()
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
        del r
    barrier()
    dt = time.perf_counter() - t0
    ctx.timing_enable(False)
    kern_ms, launches = 0.0, 0
    for kn in kernel_name.split("+"):
        a_ms, a_n = ctx.timing_query(kn)
        kern_ms += a_ms
        launches += a_n
    breakdown = {}
    for kn in ("agg_grouped", "agg_table_init", "agg_collect", "agg_finalize", "agg_rank_finalize", "bitonic_small", "keep_from_simple", "compact_expr",
               "compact_column", "compact_gather", "join_probe_unique", "join_probe_presence", "join_fused_write", "join_probe_count", "join_probe_write", "scan_chunk", "scan_add", "scan_single"):
        b_ms, b_n = ctx.timing_query(kn)
        if b_n:
            breakdown[kn] = {"ms_per_step": b_ms / args.steps, "launches_per_step": b_n / args.steps}
    if world > 1 or force_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    rows_per_s = total * args.steps / dt

    if rank == 0:
        kavg_ms = kern_ms / args.steps  # per step (sum over the kernel's launches in one step)
        achieved = algo_bytes_per_row * n / (kavg_ms * 1e-3) / 1e9 if kavg_ms > 0 else 0.0
        out = {
            "metric": metric, "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": desc, "rows_per_gpu": n, "total_rows": total, "parallelism": f"row-range x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": kernel_name, "kernel_ms_per_step": kavg_ms,
                         "algorithmic_bytes_per_step": algo_bytes_per_row * n, "kernels": breakdown},
        }
        # HBM bytes per launch from the committed PMC passes of this workload (rocprofv3 cannot be run from inside
        # the timed process; see profiles/): only reported for the exact configuration that was profiled.
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic_headline.json")
        if args.workload == "headline" and n == 10**9 and not args.random_keys and args.pass_frac == 0.5 and os.path.exists(pmc):
            with open(pmc) as f:
                out["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch_corrected"]
            out["roofline"]["traffic_source"] = "profiles/r01/pmc_traffic_headline.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 fetch correction)"
        if world == 1 and not args.no_cpu_baseline and args.workload in ("headline", "c3", "c2"):
            out["cpu_baseline"] = cpu_baseline(args, n)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, n):
    """The oracle (C++ restatement of the reference's CPU algorithm, single-threaded like the
    reference) on a bounded sample of the same workload.  Reported, never the target."""
    import numpy as np

    from naive_query_engine_amd import AggregateFunc, Column, Operator
    from naive_query_engine_amd.expression import binop, col, lit_i64
    from oracle import oracle as orc

    class F:
        def __init__(self, name):
            self.name = name

    m = min(n, args.cpu_sample_rows)
    ids = orc.synth_fill(0, 0, 0, m).view(np.int64)
    fields = [F("id"), F("x")]
    if args.workload == "c2":
        x = orc.synth_fill(1, 2, 0, m, 60, 18).view(np.int64)
        h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(x)]])
        pred = binop(col(0), Operator.Lt, lit_i64(m // 2)).flatten(fields)
        proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(fields)]
        t0 = time.perf_counter()
        sel = orc.selection(h, pred, raw=True)
        r = orc.projection(sel, proj, raw=True)
        dt = time.perf_counter() - t0
    else:
        x = orc.synth_fill(2, 3, 0, m).view(np.float64)
        h = orc.upload([[Column.from_numpy(ids), Column.from_numpy(x)]])
        aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
        key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(fields)
        pred = binop(col(0), Operator.Lt, lit_i64(m // 2)).flatten(fields) if args.workload == "headline" else None
        t0 = time.perf_counter()
        r = orc.aggregate(h, aggs, group_nodes=key, pred_nodes=pred, raw=True)
        dt = time.perf_counter() - t0
    del r
    return {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"same query on the first {m} rows (single thread; the reference is single-threaded; host has {os.cpu_count()} cores)",
            "seconds": dt}


if __name__ == "__main__":
    main()
