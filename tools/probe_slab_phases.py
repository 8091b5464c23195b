#!/usr/bin/env python
"""Where a tile of agg_slab_scatter_kernel spends its time (diagnostic build only).

    cd naive_query_engine_amd/csrc && hipcc <Makefile flags> -DNQE_SLAB_PROFILE -c aggregate_partition.hip -o _var_prof/aggregate_partition.o
    hipcc -shared … -o ../libnqe_hip_prof.so <the other objects> _var_prof/aggregate_partition.o
    NQE_LIB_PATH=naive_query_engine_amd/libnqe_hip_prof.so python tools/probe_slab_phases.py [rows] [groups]

Thread 0 of every scatter workgroup accumulates shader-clock deltas between the barriers of a tile; the sums divided by the number of
workgroups and tiles give clocks per tile per phase."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naive_query_engine_amd import AggregateFunc, capi  # noqa: E402
from naive_query_engine_amd.expression import col  # noqa: E402
from tests.helpers import fields  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
import torch  # noqa: E402

from naive_query_engine_amd import DType  # noqa: E402

ctx = capi.Context(0)
dev = torch.device("cuda", 0)
kt = torch.empty(rows, dtype=torch.int64, device=dev)
vt = torch.empty(rows, dtype=torch.float64, device=dev)
ctx.synth_fill(1, 7, 0, rows, groups, 0, kt.data_ptr())   # random keys in [0, groups)
ctx.synth_fill(2, 3, 0, rows, 1, 0, vt.data_ptr())        # Float64 values in [0, 100)
t = ctx.table_from_device([(DType.INT64, rows, kt.data_ptr(), None), (DType.FLOAT64, rows, vt.data_ptr(), None)])
aggs = [(AggregateFunc.Count, 1), (AggregateFunc.Sum, 1), (AggregateFunc.Avg, 1), (AggregateFunc.Min, 1), (AggregateFunc.Max, 1)]
kn = col(0).flatten(fields("id", "v"))
lib = capi.lib()
lib.nqe_debug_slab_profile.argtypes = [C.POINTER(C.c_ulonglong)]
lib.nqe_debug_slab_profile.restype = None
out = (C.c_ulonglong * 8)()
for rep in range(3):
    ctx.aggregate(t, aggs, group_nodes=kn)
    ctx.synchronize()
    lib.nqe_debug_slab_profile(out)
    wgs = max(1, out[7])
    tiles = rows / 8192 / wgs
    names = ["wait loads + key + rank atomics", "scan", "tuples to LDS + issue next loads", "copy-out", "cursors"]
    tot = sum(out[i] for i in range(5))
    print(f"rep {rep}: {wgs} workgroups, {tiles:.1f} tiles each, {tot / wgs / tiles:.0f} clocks per tile")
    for i, nm in enumerate(names):
        print(f"    {nm:36s} {out[i] / wgs / tiles:9.0f} clocks per tile  {100.0 * out[i] / max(tot, 1):5.1f} %")
