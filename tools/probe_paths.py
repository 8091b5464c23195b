#!/usr/bin/env python
"""Diagnostics on the GPU box: PCIe-inclusive ingest, nullable aggregate path, high-cardinality group-by."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from naive_query_engine_amd import AggregateFunc, Column, DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_i64


class F:
    def __init__(self, n):
        self.name = n


SECTION = sys.argv[1] if len(sys.argv) > 1 else "all"
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
f = [F("id"), F("v")]
aggs5 = lambda c: [(AggregateFunc.Count, c), (AggregateFunc.Sum, c), (AggregateFunc.Avg, c), (AggregateFunc.Min, c), (AggregateFunc.Max, c)]
aggs = aggs5(1)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        r = fn(); del r
    ctx.synchronize()
    ctx.jit_wait()  # expression trees: the steady state is the run-time specialised kernel (NQE_NO_JIT=1 for the interpreter)
    r = fn(); del r
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn(); del r
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)   # (used by several sections below)
# ---- 1. PCIe-inclusive: host numpy columns → nqe_table_create → aggregate (only when asked for: SECTION all / pcie)
if SECTION in ('all', 'pcie'):
    n = 100_000_000
    ids = np.arange(n, dtype=np.int64)
    v = np.random.default_rng(0).random(n) * 100
    gb = 2 * 8 * n / 1e9
    ctx.synchronize()
    t0 = time.perf_counter()
    t = ctx.table_from_host([Column.from_numpy(ids), Column.from_numpy(v)])
    ctx.synchronize()
    up = time.perf_counter() - t0
    key = binop(col(0), Operator.Modulos, lit_i64(1024)).flatten(f)
    pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
    q = timeit(lambda: ctx.aggregate(t, aggs, group_nodes=key, pred_nodes=pred))
    print(f"upload {n} rows x 2 cols ({gb:.2f} GB, pageable numpy): {up*1e3:.1f} ms = {gb/up:.1f} GB/s; query {q*1e3:.3f} ms; "
          f"PCIe-inclusive {n/(up+q):.3e} rows/s vs resident {n/q:.3e} rows/s")
    del t, ids, v

# ---- 2. nullable columns (1% nulls) → general kernel
n = 200_000_000 if SECTION in ('all', 'paths', 'keys', 'exprs', 'trees') else 100_000_000
pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)   # `id < n / 2` over THIS table (rounds 1-3 left it at `id < 500`: their key / path sweeps had next to no row pass)
idt = torch.empty(n, dtype=torch.int64, device=dev); ctx.synchronize()
ctx.synth_fill(0, 0, 0, n, 1, 0, idt.data_ptr())
vt = torch.empty(n, dtype=torch.float64, device=dev)
ctx.synth_fill(2, 3, 0, n, 1, 0, vt.data_ptr())
valid = (torch.rand(n, device=dev) > 0.01)
# pack LSB-first validity bitmap on the device
pad = (-n) % 64
bits = torch.cat([valid, torch.zeros(pad, dtype=torch.bool, device=dev)]).view(-1, 8).to(torch.uint8)
w = (bits * torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)).sum(dim=1).to(torch.uint8).contiguous()
torch.cuda.synchronize()
plain = ctx.table_from_device([(DType.INT64, n, idt.data_ptr(), None), (DType.FLOAT64, n, vt.data_ptr(), None)])
nullable = ctx.table_from_device([(DType.INT64, n, idt.data_ptr(), None), (DType.FLOAT64, n, vt.data_ptr(), w.data_ptr())])
pred = binop(col(0), Operator.Lt, lit_i64(n // 2)).flatten(f)
for name, tab in ((("plain", plain), ("1% null values", nullable)) if SECTION in ("all", "paths") else ()):
    q = timeit(lambda: ctx.aggregate(tab, aggs, group_nodes=key, pred_nodes=pred))
    print(f"aggregate {n} rows [{name}]: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")
if SECTION in ("all", "keys"):
    for name, kexpr in (("id % 1024", binop(col(0), Operator.Modulos, lit_i64(1024))), ("id % 3", binop(col(0), Operator.Modulos, lit_i64(3))),
                        ("id % 16", binop(col(0), Operator.Modulos, lit_i64(16))), ("id % 64", binop(col(0), Operator.Modulos, lit_i64(64))),
                        ("id % 256", binop(col(0), Operator.Modulos, lit_i64(256))), ("id % 512", binop(col(0), Operator.Modulos, lit_i64(512))),
                        ("id % 2048", binop(col(0), Operator.Modulos, lit_i64(2048))),
                        ("id % 1000", binop(col(0), Operator.Modulos, lit_i64(1000))), ("id % 2000", binop(col(0), Operator.Modulos, lit_i64(2000))),
                        ("(id + 1) % 1000", binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(1000)))):
        q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=kexpr.flatten(f), pred_nodes=pred), reps=20)
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(5):
            r = ctx.aggregate(plain, aggs, group_nodes=kexpr.flatten(f), pred_nodes=pred); del r
        ctx.timing_enable(False)
        br = {k: round(ctx.timing_query(k)[0] / 5, 4) for k in ("agg_grouped", "agg_table_init", "agg_rank_finalize", "agg_collect", "bitonic", "agg_finalize")}
        print(f"aggregate {n} rows key {name}: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s  kernels(ms) {br}")
if SECTION in ("all", "keys"):
    from naive_query_engine_amd.expression import lit_i64 as _li
    for name, pe in (("id % 10 < 5", binop(binop(col(0), Operator.Modulos, _li(10)), Operator.Lt, _li(5))),
                     ("id * 3 >= 300000000", binop(binop(col(0), Operator.Multiply, _li(3)), Operator.GtEq, _li(300000000)))):
        q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=pe.flatten(f)), reps=20)
        print(f"aggregate {n} rows, chain predicate {name}, key id % 1024: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")
if SECTION in ("all", "paths"):
    gen = binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(1000)).flatten(f)   # non-pow2 modulus: general key
    q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=gen, pred_nodes=pred))
    print(f"aggregate {n} rows [key (id+1) % 1000, out-of-line divide]: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")
    q = timeit(lambda: ctx.aggregate(plain, aggs, pred_nodes=pred))
    print(f"un-grouped aggregate {n} rows: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")

# ---- 2b. expression trees: single-pass stack machine vs algorithmic bytes
if SECTION in ("all", "exprs"):
    from naive_query_engine_amd.expression import lit_f64
    cases = [("v*v + v/4 (2 reads? no: 1 col, 1 write)", binop(binop(col(1), Operator.Multiply, col(1)), Operator.Plus, binop(col(1), Operator.Divide, lit_f64(4.0))), 16),
             ("(id + 1) * (id - 1)", binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Multiply, binop(col(0), Operator.Minus, lit_i64(1))), 16),
             ("(id % 1000) * 3 + id / 7", binop(binop(binop(col(0), Operator.Modulos, lit_i64(1000)), Operator.Multiply, lit_i64(3)), Operator.Plus, binop(col(0), Operator.Divide, lit_i64(7))), 16),
             ("v > 50 and id % 3 == 0 (bool out)", binop(binop(col(1), Operator.Gt, lit_f64(50.0)), Operator.And, binop(binop(col(0), Operator.Modulos, lit_i64(3)), Operator.Eq, lit_i64(0))), 16.125),
             ("id + 1 (single node)", binop(col(0), Operator.Plus, lit_i64(1)), 16),
             ("id+1+1 (2 ops)", binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Plus, lit_i64(1)), 16),
             ("id+1+1+1+1 (4 ops)", binop(binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Plus, lit_i64(1)), Operator.Plus, lit_i64(1)), Operator.Plus, lit_i64(1)), 16),
             ("8 ops chain", __import__("functools").reduce(lambda a, _: binop(a, Operator.Plus, lit_i64(1)), range(8), col(0)), 16),
             ("id + v? no: id + id (2 pushes col)", binop(binop(col(0), Operator.Plus, col(0)), Operator.Plus, col(0)), 16)]
    for name, e, bpr in cases:
        q = timeit(lambda: ctx.expr_evaluate(plain, e.flatten(f)))
        print(f"expr {name}: {n} rows {q*1e3:.3f} ms = {bpr*n/q/1e9:.0f} GB/s algorithmic")
    tp = binop(binop(binop(col(0), Operator.Plus, lit_i64(1)), Operator.Modulos, lit_i64(10)), Operator.Lt, lit_i64(5))
    q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=tp.flatten(f)))
    print(f"aggregate with tree predicate ((id+1)%10 < 5): {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")
    for name, tp2 in (("v * 2.0 > 100.0 (Float64 chain)", binop(binop(col(1), Operator.Multiply, lit_f64(2.0)), Operator.Gt, lit_f64(100.0))),
                      ("(v + 5.0) / 3.0 <= 20.0 (Float64 chain)", binop(binop(binop(col(1), Operator.Plus, lit_f64(5.0)), Operator.Divide, lit_f64(3.0)), Operator.LtEq, lit_f64(20.0))),
                      ("id < N/2 and v > 10 (two tests, in-kernel)", binop(binop(col(0), Operator.Lt, lit_i64(n // 2)), Operator.And, binop(col(1), Operator.Gt, lit_f64(10.0)))),
                      ("v < 20 or id % 3 == 0 (general tree: materialised Boolean column)", binop(binop(col(1), Operator.Lt, lit_f64(20.0)), Operator.Or, binop(binop(col(0), Operator.Modulos, lit_i64(3)), Operator.Eq, lit_i64(0))))):
        q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=tp2.flatten(f)))
        print(f"aggregate with predicate {name}: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s")
    q = timeit(lambda: ctx.selection_projection(plain, tp.flatten(f), [cases[0][1].flatten(f), col(0).flatten(f)]))
    print(f"selection(tree pred)+projection(tree, id): {q*1e3:.3f} ms = {(16*n + 8*n)/q/1e9:.0f} GB/s algorithmic (50% pass)")
    ctx.timing_enable(True); ctx.timing_reset()
    r = ctx.selection_projection(plain, tp.flatten(f), [cases[0][1].flatten(f), col(0).flatten(f)]); del r
    ctx.timing_enable(False)
    print("   kernels(ms):", {k: round(ctx.timing_query(k)[0], 3) for k in ("expr_tree", "expr_tree_compact", "keep_from_pred", "scan_", "compact_column", "compact_expr", "pack_bytes")})
    ctx.timing_enable(True); ctx.timing_reset()
    r = ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=tp.flatten(f)); del r
    ctx.timing_enable(False)
    print("   agg kernels(ms):", {k: round(ctx.timing_query(k)[0], 3) for k in ("expr_tree", "agg_grouped_fast", "agg_grouped", "agg_table_init", "agg_collect", "agg_finalize")})

# ---- 2c. CSV ingest: 1M rows (the reference keeps only the first 1M-row batch, quirk Q1)
if SECTION in ("all", "csv"):
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    m = 1_000_000
    ids = np.arange(m)
    price = rng.random(m) * 1000
    qty = rng.integers(-10**9, 10**9, m)
    names = np.array(["alice", "bob", "carol", "dave", "eve", "mallory, \"the\" quoted"])[rng.integers(0, 6, m)]
    lines = ["id,price,qty,name"] + [f'{i},{p!r},{q},"{nm.replace(chr(34), chr(34) * 2)}"' if "," in nm else f"{i},{p!r},{q},{nm}" for i, p, q, nm in zip(ids, price, qty, names)]
    data = ("\n".join(lines) + "\n").encode()
    t0 = time.perf_counter(); nm_, dts, _ = ctx.csv_infer_schema(data); t_inf = time.perf_counter() - t0
    q_host = timeit(lambda: ctx.csv_read(data, dts), reps=5, warm=2)
    padded = data + b"\0" * ((-len(data)) % 8)
    holder = ctx.table_from_host([Column.from_numpy(np.frombuffer(padded, dtype=np.uint64).copy())])
    dptr = int(holder.column_info(0).values)
    q_dev = timeit(lambda: ctx.csv_read(None, dts, device_ptr=dptr, nbytes=len(data)), reps=5, warm=2)
    ctx.timing_enable(True); ctx.timing_reset()
    r = ctx.csv_read(None, dts, device_ptr=dptr, nbytes=len(data)); del r
    ctx.timing_enable(False)
    br = {k: round(ctx.timing_query(k)[0], 3) for k in ("csv_vec", "csv_block_scan", "csv_mark_count", "csv_mark_write", "csv_fields", "csv_fields_copy", "scan_", "pack_bytes")}
    t0 = time.perf_counter(); o = orc.csv_read(data); t_cpu = time.perf_counter() - t0
    print(f"csv {m} rows, {len(data)/1e6:.1f} MB: infer {t_inf*1e3:.2f} ms (host); read from host bytes {q_host*1e3:.2f} ms ({len(data)/q_host/1e9:.2f} GB/s), "
          f"HBM-resident image {q_dev*1e3:.2f} ms ({len(data)/q_dev/1e9:.2f} GB/s = {m/q_dev:.3e} rows/s); oracle (1 thread) {t_cpu*1e3:.0f} ms ({len(data)/t_cpu/1e9:.3f} GB/s)")
    print("   kernels(ms):", br)

# ---- 2c. Utf8 columns in filter / group key / join key: 10^7 rows, strings of 3-22 bytes
if SECTION in ("all", "strings"):
    from oracle import oracle as orc
    from naive_query_engine_amd.expression import lit_utf8
    rng = np.random.default_rng(1)
    m = 10_000_000

    def utf8_column(words, idx):
        lens = np.array([len(w) for w in words], dtype=np.int32)
        width = int(lens.max())
        mat = np.zeros((len(words), width), dtype=np.uint8)
        for i, w in enumerate(words):
            mat[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        ln = lens[idx]
        offs = np.zeros(len(idx) + 1, dtype=np.int32)
        np.cumsum(ln, out=offs[1:])
        data = mat[idx][np.arange(width)[None, :] < ln[:, None]]
        return Column(DType.UTF8, len(idx), offs, None, data)

    few = [b"alice", b"bob", b"carol", b"dave", b"eve", b"mallory, the quoted one"]
    many = [f"customer-{i:07d}".encode() for i in range(100_000)]
    name = utf8_column(few, rng.integers(0, len(few), m))
    cust = utf8_column(many, rng.integers(0, len(many), m))
    v = Column.from_numpy(rng.random(m) * 100)
    fs = [type("F", (), {"name": x})() for x in ("name", "cust", "v")]
    t = ctx.table_from_host([name, cust, v])
    nbytes = name.data.size + cust.data.size + 8 * m + 8 * m
    pred = binop(col(0), Operator.Eq, lit_utf8("carol")).flatten(fs)
    q = timeit(lambda: ctx.selection(t, pred), reps=5, warm=2)
    print(f"utf8 filter name = 'carol' over {m} rows x (Utf8, Utf8, Float64), all columns compacted: {q*1e3:.2f} ms = {m/q:.3e} rows/s")
    for label, kc in (("6 distinct strings", 0), ("100000 distinct strings", 1)):
        q = timeit(lambda: ctx.aggregate(t, aggs5(2), group_nodes=col(kc).flatten(fs)), reps=5, warm=2)
        print(f"utf8 group key ({label}): count/sum/avg/min/max(v) over {m} rows: {q*1e3:.2f} ms = {m/q:.3e} rows/s")
    dim = ctx.table_from_host([utf8_column(many, rng.permutation(len(many))), Column.from_numpy(rng.integers(0, 1 << 20, len(many)).astype(np.int64))])
    q = timeit(lambda: ctx.hash_join(dim, t, 0, 1), reps=3, warm=1)
    print(f"utf8 join key: dim(100000 strings, attr) join fact({m} rows) on cust: {q*1e3:.2f} ms = {m/q:.3e} probe rows/s (output {m} rows x 5 columns incl. 3 Utf8)")
    sample = 1_000_000
    hs = orc.upload([[Column(DType.UTF8, sample, name.values[:sample + 1].copy(), None, name.data[:name.values[sample]].copy()),
                      Column(DType.UTF8, sample, cust.values[:sample + 1].copy(), None, cust.data[:cust.values[sample]].copy()), Column.from_numpy(v.to_numpy()[:sample])]])
    t0 = time.perf_counter(); orc.selection(hs, pred, raw=True); t_f = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.aggregate(hs, aggs5(2), group_nodes=col(1).flatten(fs)); t_a = time.perf_counter() - t0
    print(f"   oracle (1 thread, first {sample} rows): filter {sample/t_f:.3e} rows/s, utf8-key aggregate {sample/t_a:.3e} rows/s")

# ---- 3. high-cardinality group-by
for gi, groups in enumerate((1 << 10, 2000, 3000, 1 << 12, 6000, 1 << 14, 1 << 17, 500_000, 1 << 20, 1 << 24) if SECTION in ('all', 'groups') else ()):
    kt = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.synth_fill(1, 7, 0, n, groups, 0, kt.data_ptr())
    # (the caching allocator hands every case the same buffer, and what the context remembers about a query is keyed by buffer and row
    # count: 64 rows fewer per case, or a case starts from the previous one's plan and key range)
    ng = n - 64 * gi
    tab = ctx.table_from_device([(DType.INT64, ng, kt.data_ptr(), None), (DType.FLOAT64, ng, vt.data_ptr(), None)])
    q = timeit(lambda: ctx.aggregate(tab, aggs, group_nodes=col(0).flatten(f)), reps=3, warm=1)
    ctx.timing_enable(True); ctx.timing_reset()
    r = ctx.aggregate(tab, aggs, group_nodes=col(0).flatten(f)); del r
    ctx.timing_enable(False)
    br = {k: round(ctx.timing_query(k)[0], 3) for k in ("agg_grouped_fast", "agg_partition_count", "agg_partition_scatter", "agg_segments", "scan_", "agg_table_init", "agg_collect", "radix", "agg_finalize")}
    print(f"group by random key, {groups} groups, {n} rows: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s  kernels(ms) {br}")
    del tab, kt

# ---- 4. join with duplicate build keys (general two-pass path)
if SECTION in ("all", "dupjoin"):
    nb, npr = 1_000_000, 50_000_000
    for dup in (1, 2, 4):
        bk = torch.arange(nb, device=dev, dtype=torch.int64) // dup          # every key `dup` times
        ba = torch.arange(nb, device=dev, dtype=torch.int64) * 3
        pk = torch.empty(npr, dtype=torch.int64, device=dev); torch.cuda.synchronize()
        ctx.synth_fill(1, 5, 0, npr, nb // dup, 0, pk.data_ptr())
        pv = torch.empty(npr, dtype=torch.float64, device=dev); ctx.synth_fill(2, 3, 0, npr, 1, 0, pv.data_ptr())
        dim = ctx.table_from_device([(DType.INT64, nb, bk.data_ptr(), None), (DType.INT64, nb, ba.data_ptr(), None)])
        fact = ctx.table_from_device([(DType.INT64, npr, pk.data_ptr(), None), (DType.FLOAT64, npr, pv.data_ptr(), None)])
        jt = ctx.hash_join_build(dim, 0)
        q = timeit(lambda: ctx.hash_join_probe(jt, fact, 0), reps=3, warm=1)
        ctx.timing_enable(True); ctx.timing_reset()
        r = ctx.hash_join_probe(jt, fact, 0); rows = r.num_rows; del r
        ctx.timing_enable(False)
        br = {k: round(ctx.timing_query(k)[0], 3) for k in ("join_probe_count", "join_probe_write", "join_probe_presence", "join_fused_write", "join_probe_unique", "compact", "scan_")}
        print(f"join dup={dup}: {npr} probe rows -> {rows} rows: {q*1e3:.3f} ms  ({(npr*16+rows*32)/q/1e9:.0f} GB/s algorithmic)  kernels(ms) {br}")

# ---- 5. join shapes beyond C4 (dense unique 1e6-row build): sparse keys, larger builds, partial match
if SECTION in ("all", "joinshapes"):
    npr = 100_000_000
    pv = torch.empty(npr, dtype=torch.float64, device=dev); ctx.synth_fill(2, 3, 0, npr, 1, 0, pv.data_ptr())
    def run(name, bk, pk):
        nb = bk.numel()
        ba = torch.arange(nb, device=dev, dtype=torch.int64) * 3
        torch.cuda.synchronize()
        dim = ctx.table_from_device([(DType.INT64, nb, bk.data_ptr(), None), (DType.INT64, nb, ba.data_ptr(), None)])
        fact = ctx.table_from_device([(DType.INT64, npr, pk.data_ptr(), None), (DType.FLOAT64, npr, pv.data_ptr(), None)])
        t0 = time.perf_counter(); jt = ctx.hash_join_build(dim, 0); ctx.synchronize(); tb = time.perf_counter() - t0
        q = timeit(lambda: ctx.hash_join_probe(jt, fact, 0), reps=3, warm=1)
        ctx.timing_enable(True); ctx.timing_reset()
        r = ctx.hash_join_probe(jt, fact, 0); rows = r.num_rows; del r
        ctx.timing_enable(False)
        br = {k: round(ctx.timing_query(k)[0], 3) for k in ("join_probe_count", "join_probe_write", "join_probe_presence", "join_fused_write", "join_probe_unique", "compact", "scan_")}
        br = {k: v for k, v in br.items() if v}
        print(f"join [{name}] build {nb} rows {tb*1e3:.2f} ms; probe {npr} rows -> {rows}: {q*1e3:.3f} ms = {npr/q:.3e} probe rows/s ({(npr*16+rows*32)/q/1e9:.0f} GB/s algorithmic)  {br}")
    g = torch.Generator(device=dev); g.manual_seed(1)
    for nb in (1_000_000, 10_000_000, 100_000_000):
        bk = torch.randperm(nb, device=dev, generator=g)
        pk = torch.randint(0, nb, (npr,), device=dev, generator=g)
        run(f"dense unique, every probe row matches, build {nb:.0e}", bk, pk)
        del bk, pk
    nb = 1_000_000
    sparse = torch.unique(torch.randint(0, 1 << 40, (nb + nb // 8,), device=dev, generator=g))[:nb]
    sparse = sparse[torch.randperm(sparse.numel(), device=dev, generator=g)]
    pk = sparse[torch.randint(0, sparse.numel(), (npr,), device=dev, generator=g)]
    run("sparse unique keys (2^40 domain), all match", sparse, pk)
    pk2 = torch.where(torch.rand(npr, device=dev, generator=g) < 0.1, pk, pk + 1)
    run("sparse unique keys, ~10% match", sparse, pk2)
    bk = torch.randperm(nb, device=dev, generator=g)
    pk3 = torch.randint(0, nb * 10, (npr,), device=dev, generator=g)
    run("dense unique, ~10% match", bk, pk3)

# ---- 6. predicate trees inside the aggregate's streaming kernel (PRED = 5) against chains / range lists / the materialised form
if SECTION in ("all", "trees"):
    from naive_query_engine_amd.expression import lit_f64
    B = binop
    O = Operator
    I, V = col(0), col(1)
    trees = [("id < N/2 (range test)", B(I, O.Lt, lit_i64(n // 2))),
             ("id % 10 < 5 (chain)", B(B(I, O.Modulos, lit_i64(10)), O.Lt, lit_i64(5))),
             ("id < N/2 and v > 10 (range list)", B(B(I, O.Lt, lit_i64(n // 2)), O.And, B(V, O.Gt, lit_f64(10.0)))),
             ("v < 20 or id % 3 == 0 (tree)", B(B(V, O.Lt, lit_f64(20.0)), O.Or, B(B(I, O.Modulos, lit_i64(3)), O.Eq, lit_i64(0)))),
             ("v < 20 or id % 4 == 0 (tree, pow2)", B(B(V, O.Lt, lit_f64(20.0)), O.Or, B(B(I, O.Modulos, lit_i64(4)), O.Eq, lit_i64(0)))),
             ("v < 20 or id + 5 < 1000 (tree, add)", B(B(V, O.Lt, lit_f64(20.0)), O.Or, B(B(I, O.Plus, lit_i64(5)), O.Lt, lit_i64(1000)))),
             ("v * 2.0 < id-free: v * 2.0 < 40.0 or v > 90.0 (tree)", B(B(B(V, O.Multiply, lit_f64(2.0)), O.Lt, lit_f64(40.0)), O.Or, B(V, O.Gt, lit_f64(90.0)))),
             ("6 compares and/or", B(B(B(V, O.Lt, lit_f64(20.0)), O.Or, B(B(I, O.Modulos, lit_i64(3)), O.Eq, lit_i64(0))), O.And,
                                     B(B(B(I, O.Plus, lit_i64(7)), O.Gt, lit_i64(100)), O.Or, B(B(V, O.Multiply, V), O.Lt, lit_f64(2500.0)))))]
    for name, tr in trees:
        q = timeit(lambda: ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=tr.flatten(f)), reps=10)
        ctx.timing_enable(True); ctx.timing_reset()
        r = ctx.aggregate(plain, aggs, group_nodes=key, pred_nodes=tr.flatten(f)); del r
        ctx.timing_enable(False)
        br = {k: round(v[0], 3) for k, v in ctx.timing_report().items() if v[0] > 0.02}
        print(f"aggregate {n} rows under {name}: {q*1e3:.3f} ms = {16*n/q/1e9:.0f} GB/s  {br}")
