import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.zeros(1, device="cuda")
from naive_query_engine_amd import AggregateFunc, Column, DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_utf8
ctx = capi.Context(0)
rng = np.random.default_rng(1)
m = 10_000_000
def utf8_column(words, idx):
    lens = np.array([len(w) for w in words], dtype=np.int32); width = int(lens.max())
    mat = np.zeros((len(words), width), dtype=np.uint8)
    for i, w in enumerate(words): mat[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
    ln = lens[idx]; offs = np.zeros(len(idx) + 1, dtype=np.int32); np.cumsum(ln, out=offs[1:])
    data = mat[idx][np.arange(width)[None, :] < ln[:, None]]
    return Column(DType.UTF8, len(idx), offs, None, data)
few = [b"alice", b"bob", b"carol", b"dave", b"eve", b"mallory, the quoted one"]
many = [f"customer-{i:07d}".encode() for i in range(100_000)]
name = utf8_column(few, rng.integers(0, len(few), m)); cust = utf8_column(many, rng.integers(0, len(many), m)); v = Column.from_numpy(rng.random(m) * 100)
fs = [type("F", (), {"name": x})() for x in ("name", "cust", "v")]
t = ctx.table_from_host([name, cust, v])
dim = ctx.table_from_host([utf8_column(many, rng.permutation(len(many))), Column.from_numpy(rng.integers(0, 1 << 20, len(many)).astype(np.int64))])
pred = binop(col(0), Operator.Eq, lit_utf8("carol")).flatten(fs)
def prof(label, fn):
    for _ in range(2): r = fn(); del r
    ctx.synchronize(); ctx.timing_enable(True); ctx.timing_reset()
    t0 = time.perf_counter(); r = fn(); ctx.synchronize(); dt = time.perf_counter() - t0
    ctx.timing_enable(False)
    rep = sorted(ctx.timing_report().items(), key=lambda kv: -kv[1][0])
    print(label, f"{dt*1e3:.2f} ms wall;", ", ".join(f"{k} {ms:.3f}ms x{int(c)}" for k, (ms, c) in rep[:10]))
prof("join", lambda: ctx.hash_join(dim, t, 0, 1))
prof("filter", lambda: ctx.selection(t, pred))
aggs=[(AggregateFunc.Count,2),(AggregateFunc.Sum,2)]
prof("agg100k", lambda: ctx.aggregate(t, aggs, group_nodes=col(1).flatten(fs)))
