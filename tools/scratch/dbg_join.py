import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from naive_query_engine_amd import DType, capi
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
def synth(kind, seed, rows, first=0, mod=1, base=0, dtype=torch.int64):
    t = torch.empty(rows, dtype=dtype, device=dev); ctx.synth_fill(kind, seed, first, rows, mod, base, t.data_ptr()); return t
for nb in (10**6, 10**7):
    for attr_kind in ("synth20", "arange3"):
        g = torch.Generator(device=dev).manual_seed(7)
        perm = torch.randperm(nb, device=dev, generator=g).to(torch.int64)
        print(nb, attr_kind, "unique:", int(torch.unique(perm).numel()), "min", int(perm.min()), "max", int(perm.max()))
        attr = synth(1, 4, nb, 0, 1 << 20, 0) if attr_kind == "synth20" else torch.arange(nb, device=dev) * 3
        n = 10**7
        fkey = synth(1, 5, n, 0, nb, 0); val = synth(2, 3, n, dtype=torch.float64)
        ctx.synchronize(); torch.cuda.synchronize()
        dim = ctx.table_from_device([(DType.INT64, nb, perm.data_ptr(), None), (DType.INT64, nb, attr.data_ptr(), None)])
        fact = ctx.table_from_device([(DType.INT64, n, fkey.data_ptr(), None), (DType.FLOAT64, n, val.data_ptr(), None)])
        for rep in range(3):
            jt = ctx.hash_join_build(dim, 0)
            ctx.timing_enable(True); ctx.timing_reset()
            r = ctx.hash_join_probe(jt, fact, 0); rows = r.num_rows; del r
            ctx.timing_enable(False)
            print("   rep", rep, "rows", rows, {k: round(v[0], 3) for k, v in ctx.timing_report().items()})
            del jt
