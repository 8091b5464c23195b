import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from naive_query_engine_amd import AggregateFunc as A, DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_i64
class F:
    def __init__(s, n): s.name = n
ctx = capi.Context(0); dev = torch.device("cuda", 0)
n = 10**9
ids = torch.empty(n, dtype=torch.int64, device=dev); age = torch.empty(n, dtype=torch.int64, device=dev); sc = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ctx.synth_fill(0, 0, 0, n, 1, 0, ids.data_ptr()); ctx.synth_fill(1, 2, 0, n, 60, 18, age.data_ptr()); ctx.synth_fill(2, 3, 0, n, 1, 0, sc.data_ptr()); ctx.synchronize()
t = ctx.table_from_device([(DType.INT64, n, ids.data_ptr(), None), (DType.INT64, n, age.data_ptr(), None), (DType.FLOAT64, n, sc.data_ptr(), None)])
f = [F("id"), F("age"), F("score")]
def run(name, aggs, key):
    k = key.flatten(f)
    for _ in range(3): r = ctx.aggregate(t, aggs, group_nodes=k); del r
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = ctx.aggregate(t, aggs, group_nodes=k); del r
    ctx.synchronize(); print(name, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
k3 = binop(col(0), Operator.Modulos, lit_i64(3)); k1024 = binop(col(0), Operator.Modulos, lit_i64(1024))
for kn, k in (("id%3", k3), ("id%1024", k1024)):
    run(kn + " count(id)", [(A.Count, 0)], k)
    run(kn + " sum(age),avg(score)", [(A.Sum, 1), (A.Avg, 2)], k)
    run(kn + " sum(age)", [(A.Sum, 1)], k)
    run(kn + " all three", [(A.Count, 0), (A.Sum, 1), (A.Avg, 2)], k)
    run(kn + " 5 aggs of score", [(A.Count, 2), (A.Sum, 2), (A.Avg, 2), (A.Min, 2), (A.Max, 2)], k)
# what a changing key costs the single-load instance: mask key vs magic-multiply key, few vs many groups
for m in (3, 1000, 1024, 2048, 2047):
    run(f"id%{m} count(id)", [(A.Count, 0)], binop(col(0), Operator.Modulos, lit_i64(m)))
    run(f"id%{m} sum(id)", [(A.Sum, 0)], binop(col(0), Operator.Modulos, lit_i64(m)))
