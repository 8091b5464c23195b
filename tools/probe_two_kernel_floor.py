#!/usr/bin/env python
"""What the TWO-kernel form of `select f(v), id from t where p(id)` costs with this build's best static kernels (task 2): the
same bytes as that form moves — the predicate's column read by the keep pass, then BOTH columns read by the compaction and their kept
halves written — measured as `select id, v from t where id < N/2` over RANDOM ids (keep_from_range_tile + scan + compact_staged x 2; no
expression arithmetic at all), against the one-pass run-time specialised kernel on the expression-tree query itself.
usage: python tools/probe_two_kernel_floor.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naive_query_engine_amd import DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_f64, lit_i64

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
ctx = capi.Context(0)
ids, v = ctx.device_alloc(n * 8), ctx.device_alloc(n * 8)
ctx.synth_fill(1, 1, 0, n, n, 0, ids)
ctx.synth_fill(2, 3, 0, n, 1, 0, v)
t = ctx.table_from_device([(DType.INT64, n, ids, None), (DType.FLOAT64, n, v, None)])
F = [type("F", (), {"name": x})() for x in ("id", "v")]
plain = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(F)
tree_pred = binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5)).flatten(F)
tree_proj = [binop(binop(col(1), Operator.Multiply, col(1)), Operator.Plus, binop(col(1), Operator.Divide, lit_f64(4.0))).flatten(F), col(0).flatten(F)]


def timed(fn, reps=20):
    for _ in range(3):
        r = fn(); del r
    ctx.jit_wait()
    for _ in range(2):
        r = fn(); del r
    ctx.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); del r
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.timing_enable(False)
    return wall, {k: round(ms / reps, 4) for k, (ms, c) in ctx.timing_report().items()}


for name, fn in (("two kernels, plain columns: select id, v where id < N/2 (random ids)", lambda: ctx.selection(t, plain)),
                 ("one pass, expression trees: select v*v + v/4, id where (id+1) % 10 < 5", lambda: ctx.selection_projection(t, tree_pred, tree_proj))):
    wall, ks = timed(fn)
    print(f"{name}: {wall:.4f} ms per step, kernels {sum(ks.values()):.4f} ms {ks}", flush=True)
