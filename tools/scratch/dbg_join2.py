import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench
args = bench.parse()
from naive_query_engine_amd import AggregateFunc
B = bench.Bench(args, 1, 0, 0)
import torch
for nb in (10**6, 10**7, 10**6, 10**7):
    B.ctx.trim(); torch.cuda.empty_cache()
    r, s = bench.wl_c4(B, 10**8, nb, False, 3, 1)
    print(nb, r["roofline"]["kernel"], r["ms_per_step"], flush=True)
    del s
