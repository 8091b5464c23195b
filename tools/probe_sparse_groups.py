"""10^8 rows over G random groups whose keys are SPARSE (key = SPREAD (default 7) x a random number below G: the value range is seven times a workgroup
table's reach, so no tier can address a table by key - min): the hashed two-subset streaming form against the partitioned tier
(NQE_AGG_SUBSETS_MAX=0), wall time per execution included.  usage: python tools/probe_sparse_groups.py [G ...]"""
import os
import sys
import time

import torch  # noqa: F401 (device initialisation order: torch first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from naive_query_engine_amd import AggregateFunc as A
from naive_query_engine_amd import DType, Operator, capi
from naive_query_engine_amd.expression import binop, col, lit_i64


class F:
    def __init__(self, n):
        self.name = n


def main():
    n = 10**8
    ctx = capi.Context(0)
    f = [F("k"), F("v")]
    aggs = [(A.Count, 1), (A.Sum, 1), (A.Avg, 1), (A.Min, 1), (A.Max, 1)]
    for G in [int(x) for x in sys.argv[1:]] or [3000, 5000, 7000]:
        k = ctx.device_alloc(n * 8)
        v = ctx.device_alloc(n * 8)
        ctx.synth_fill(1, 7, 0, n, G, 0, k)
        ctx.synth_fill(2, 3, 0, n, 1, 0, v)
        t = ctx.table_from_device([(DType.INT64, n, k, None), (DType.FLOAT64, n, v, None)])
        # the sparse key column, materialised once: k * 7
        proj = ctx.projection(t, [binop(col(0), Operator.Multiply, lit_i64(int(os.environ.get("SPREAD", "7")))).flatten(f), col(1).flatten(f)])
        kn = col(0).flatten(f)
        for _ in range(3):
            ctx.aggregate(proj, aggs, group_nodes=kn)
        ctx.timing_enable(True)
        ctx.timing_reset()
        reps = 10
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.aggregate(proj, aggs, group_nodes=kn)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ctx.timing_enable(False)
        rep = {kk: round(ms / reps, 4) for kk, (ms, cnt) in ctx.timing_report().items()}
        print(f"G={G} sparse keys [SUBSETS_MAX={os.environ.get('NQE_AGG_SUBSETS_MAX', '1')}]: wall {wall:.3f} ms  {rep}", flush=True)
        del t, proj
        ctx.device_free(k)
        ctx.device_free(v)


if __name__ == "__main__":
    main()
