"""10^8 rows over G random groups through the streaming tier: the static kernel (plain key column addressed by key - min: its workgroup
tables are folded into the group table with device-scope atomics) against the run-time specialised kernel (`k % G`, NQE_AGG_JIT_ALL=2:
workgroup tables written out whole and folded by a small kernel).  usage: python tools/probe_direct.py [G ...]"""
import os
import sys

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
    for G in [int(x) for x in sys.argv[1:]] or [1024, 4096]:
        k = ctx.device_alloc(n * 8)
        v = ctx.device_alloc(n * 8)
        ctx.synth_fill(1, 7, 0, n, G, 0, k)
        ctx.synth_fill(2, 3, 0, n, 1, 0, v)
        t = ctx.table_from_device([(DType.INT64, n, k, None), (DType.FLOAT64, n, v, None)])
        for name, key in (("plain key column", col(0)), (f"k % {G}", binop(col(0), Operator.Modulos, lit_i64(G)))):
            kn = key.flatten(f)
            for _ in range(3):
                ctx.aggregate(t, aggs, group_nodes=kn)
            ctx.jit_wait()
            ctx.aggregate(t, aggs, group_nodes=kn)
            ctx.timing_enable(True)
            ctx.timing_reset()
            reps = 10
            for _ in range(reps):
                ctx.aggregate(t, aggs, group_nodes=kn)
            ctx.synchronize()
            ctx.timing_enable(False)
            rep = {kk: round(ms / reps, 4) for kk, (ms, cnt) in ctx.timing_report().items()}
            print(f"G={G} {name} [JIT_ALL={os.environ.get('NQE_AGG_JIT_ALL', '1')}]: {rep}", flush=True)
        del t
        ctx.device_free(k)
        ctx.device_free(v)


if __name__ == "__main__":
    main()
