#!/usr/bin/env python
"""Diagnostics on the GPU box: join BUILD times by size, key order and form (partitioned dense build vs the forms it replaces),
with the per-kernel breakdown.  usage: probe_build.py [sizes…]   (NQE_JOIN_PART_BUILD_MIN / NQE_JOIN_PART_ONE_LEVEL select the form)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from naive_query_engine_amd import DType, capi

sizes = [int(float(a)) for a in sys.argv[1:]] or [10_000_000, 1 << 25, 100_000_000]
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
KEYS = ("join_build_minmax", "join_build_part_count", "scan_", "join_build_part_scatter", "join_build_part_fine_offsets", "join_build_part_split", "join_build_part_fill", "join_build_part_place", "join_build_dense",
        "join_build_finish")


def build_ms(dim, reps=3):
    jt = ctx.hash_join_build(dim, 0); del jt
    ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); jt = ctx.hash_join_build(dim, 0); ctx.synchronize(); best = min(best, time.perf_counter() - t0); del jt
    ctx.timing_enable(True); ctx.timing_reset()
    jt = ctx.hash_join_build(dim, 0); ctx.synchronize(); del jt
    ctx.timing_enable(False)
    br = {k: round(ctx.timing_query(k)[0], 3) for k in KEYS}
    return best * 1e3, {k: v for k, v in br.items() if v}


for nb in sizes:
    for order in ("random", "ascending"):
        bk = torch.randperm(nb, device=dev, generator=g) if order == "random" else torch.arange(nb, device=dev, dtype=torch.int64)
        ba = torch.arange(nb, device=dev, dtype=torch.int64) * 3
        bf = torch.rand(nb, device=dev, dtype=torch.float64)
        torch.cuda.synchronize()
        for name, cols in (("key + int payload", [bk, ba]), ("key + f64 payload", [bk, bf]), ("key only", [bk])):
            dim = ctx.table_from_device([(DType.FLOAT64 if c.dtype == torch.float64 else DType.INT64, nb, c.data_ptr(), None) for c in cols])
            forms = (("two-level", "1000", None), ("one-level", "1000", "1")) if order == "random" else (("partitioned", "1000", None),)
            if os.environ.get("NQE_PROBE_BUILD_PREVIOUS"):
                forms += (("previous", str(1 << 40), None),)
            for form, env, one in forms:
                os.environ["NQE_JOIN_PART_BUILD_MIN"] = env
                os.environ.pop("NQE_JOIN_PART_ONE_LEVEL", None)
                if one:
                    os.environ["NQE_JOIN_PART_ONE_LEVEL"] = one
                ms, br = build_ms(dim)
                print(f"build {nb:>11} rows, {order:9} keys, {name:18} [{form:11}] {ms:8.3f} ms  {br}", flush=True)
            del dim
        del bk, ba, bf
