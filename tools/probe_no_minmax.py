#!/usr/bin/env python
"""10^8 rows over G random groups: `count, sum, avg` (the MM = false instance: 12-byte slots, one workgroup table up to 13632 keys) against the
five-aggregate list (28-byte slots: one table up to 5840 keys, two key subsets up to 8192, then the partitioned path's range tier), with
the per-kernel breakdown.  usage: python tools/probe_no_minmax.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from naive_query_engine_amd import AggregateFunc as A, DType, capi
from naive_query_engine_amd.expression import col
n = 10**8
ctx = capi.Context(0)
F = [type("F", (), {"name": x})() for x in ("k", "v")]
for G in (4096, 6000, 8000, 12000, 13632, 14000):
    k, v = ctx.device_alloc(n * 8), ctx.device_alloc(n * 8)
    ctx.synth_fill(1, 7, 0, n, G, 0, k); ctx.synth_fill(2, 3, 0, n, 1, 0, v)
    t = ctx.table_from_device([(DType.INT64, n, k, None), (DType.FLOAT64, n, v, None)])
    for name, aggs in (("count,sum,avg", [(A.Count, 1), (A.Sum, 1), (A.Avg, 1)]), ("five", [(A.Count, 1), (A.Sum, 1), (A.Avg, 1), (A.Min, 1), (A.Max, 1)])):
        for _ in range(4):
            r = ctx.aggregate(t, aggs, group_nodes=col(0).flatten(F)); del r
        ctx.synchronize(); ctx.timing_enable(True); ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(20):
            r = ctx.aggregate(t, aggs, group_nodes=col(0).flatten(F)); del r
        ctx.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        ctx.timing_enable(False)
        ks = {kk: round(ms / 20, 4) for kk, (ms, c) in ctx.timing_report().items()}
        print(f"G={G} [{name}]: {wall:.4f} ms per step, kernels {sum(ks.values()):.4f} {ks}", flush=True)
    del t; ctx.device_free(k); ctx.device_free(v)
