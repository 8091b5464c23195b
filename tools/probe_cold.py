#!/usr/bin/env python
"""First-execution cost of the BASELINE query shapes, each in a FRESH process (the reference's run_sql is one-shot, db.rs:24-37):
ctx creation, the first execution, the first execution over a second table of the same shape (modules and pool warm, plan hints
cold) and the steady state.  `python tools/probe_cold.py` runs every shape under the default settings and with NQE_LAZY_MODULES=1
/ NQE_NO_PLAN_HINTS=1; `--one SHAPE` is the per-process worker."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ["headline", "c3", "agg_4096_groups", "agg_6000_groups", "agg_65536_groups", "agg_1048576_groups", "c2", "c2_random_ids", "c4", "c4_dim_1e7", "c4_dim_1e8", "c4_sparse_keys", "c4_dup_keys"]


def worker(shape):
    import torch

    from naive_query_engine_amd import AggregateFunc as A
    from naive_query_engine_amd import DType, Operator, capi
    from naive_query_engine_amd.expression import binop, col, lit_i64

    torch.cuda.init()
    torch.zeros(1, device="cuda")
    t0 = time.perf_counter()
    ctx = capi.Context(0)
    ctx.synchronize()
    ctx_ms = (time.perf_counter() - t0) * 1e3
    dev = torch.device("cuda", 0)

    class F:
        def __init__(s, n):
            s.name = n

    def synth(kind, seed, n, mod=1, base=0, f64=False):
        torch.cuda.synchronize()
        t = torch.empty(n, dtype=torch.float64 if f64 else torch.int64, device=dev)
        ctx.synth_fill(kind, seed, 0, n, mod, base, t.data_ptr())
        ctx.synchronize()
        return t

    five = lambda c: [(A.Count, c), (A.Sum, c), (A.Avg, c), (A.Min, c), (A.Max, c)]
    keep = []

    def make(seed_off):
        """→ a step() over a NEW set of input buffers"""
        if shape in ("headline", "c3"):
            n = 10**9
            ids, v = synth(0, 0, n), synth(2, 3 + seed_off, n, f64=True)
            keep.extend([ids, v])
            t = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
            f = [F("id"), F("v")]
            key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
            pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f) if shape == "headline" else None
            return lambda: ctx.aggregate(t, five(1), group_nodes=key, pred_nodes=pred)
        if shape.startswith("agg_"):
            n, g = 10**8, int(shape.split("_")[1])
            k, v = synth(1, 7 + seed_off, n, g), synth(2, 3, n, f64=True)
            keep.extend([k, v])
            t = ctx.table_from_device([(DType.INT64, n, k.data_ptr(), None), (DType.FLOAT64, n, v.data_ptr(), None)])
            key = col(0).flatten([F("k"), F("v")])
            return lambda: ctx.aggregate(t, five(1), group_nodes=key)
        if shape in ("c2", "c2_random_ids"):
            n = 10**8
            ids = synth(1, 1 + seed_off, n, n) if shape == "c2_random_ids" else synth(0, 0, n)
            age = synth(1, 2, n, 60, 18)
            keep.extend([ids, age])
            t = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None)])
            f = [F("id"), F("age")]
            pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
            proj = [binop(col(1), Operator.Plus, lit_i64(100)).flatten(f)]
            return lambda: ctx.selection_projection(t, pred, proj)
        n = 10**8
        nb = {"c4": 10**6, "c4_dim_1e7": 10**7, "c4_dim_1e8": 10**8}.get(shape, 10**6)
        g = torch.Generator(device=dev).manual_seed(7 + seed_off)
        perm = torch.randperm(nb, device=dev, generator=g).to(torch.int64)
        attr = synth(1, 4, nb, 1 << 20)
        fidx = synth(1, 5 + seed_off, n, nb)
        if shape == "c4_sparse_keys":
            dom = (torch.arange(nb, device=dev, dtype=torch.int64) << 20) + synth(1, 11, nb, 1 << 20)
            dkey, fkey = dom[perm].contiguous(), dom[fidx].contiguous()
        elif shape == "c4_dup_keys":
            dkey, fkey = (perm % (nb // 4)).contiguous(), fidx
        else:
            dkey, fkey = perm, fidx
        val = synth(2, 3, n, f64=True)
        torch.cuda.synchronize()
        keep.extend([dkey, attr, fkey, val])
        dim = ctx.table_from_device([(DType.INT64, nb, dkey.data_ptr(), None), (DType.INT64, nb, attr.data_ptr(), None)])
        fact = ctx.table_from_device([(DType.INT64, n, fkey.data_ptr(), None), (DType.FLOAT64, n, val.data_ptr(), None)])
        return lambda: ctx.hash_join(dim, fact, 0, 0)  # HashJoin::execute: build + probe

    def once(step):
        ctx.synchronize()
        t0 = time.perf_counter()
        r = step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        del r
        return dt

    s1 = make(0)
    first = once(s1)
    s2 = make(1)
    second_table = once(s2)
    for _ in range(3):
        once(s2)
    steady = min(once(s2) for _ in range(7))
    live, pooled = ctx.memory_stats()
    print(json.dumps({"shape": shape, "ctx_ms": round(ctx_ms, 2), "first_ms": round(first, 3), "second_table_ms": round(second_table, 3), "steady_ms": round(steady, 3),
                      "first_over_steady": round(first / steady, 2), "second_over_steady": round(second_table / steady, 2), "pooled_MB": round(pooled / 1e6, 1)}))


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        worker(sys.argv[2])
        return
    shapes = [s for s in (os.environ.get("NQE_COLD_SHAPES", "").split(",")) if s] or SHAPES
    # default = code objects loaded at context creation + a 24 GB block reserved at context creation (NQE_RESERVE_MB)
    variants = [("default (NQE_RESERVE_MB=24576)", {"NQE_RESERVE_MB": "24576"})] + [(v, {v: "1"}) for v in os.environ.get("NQE_COLD_VARIANTS", "NQE_NO_RESERVE,NQE_LAZY_MODULES").split(",") if v]
    for label, extra in variants:
        print(f"# {label}")
        for s in shapes:
            env = dict(os.environ, **extra)
            env.pop("NQE_NO_RESERVE", None)  # (a label, not a switch: this variant simply has no NQE_RESERVE_MB)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", s], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else f"{s}: FAILED rc={r.returncode} {r.stderr[-300:]}")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
