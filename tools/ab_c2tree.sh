#!/bin/bash
# A/B of the one-pass selection + projection kernel (expr_jit.hpp: nqe_jit_selproj): bench.py's c2_tree workload under its switches
out() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$1', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms_per_step'], 'frac %.3f'%r['frac'], {k:round(v['ms_per_step'],4) for k,v in json.load(open('gpurun_out/ab_details.json'))['main']['kernels'].items()})
"; }
for env in "NQE_SP_STAGED=0 NQE_SP_R=16" "NQE_SP_STAGED=0 NQE_SP_R=12" "NQE_SP_STAGED=0 NQE_SP_R=16 NQE_SP_BLOCK=512" "NQE_SP_STAGED=0 NQE_SP_BLOCK=1024 NQE_SP_R=8" "NQE_SP_STAGED=0 NQE_SP_BLOCK=1024 NQE_SP_R=12" "NQE_SP_STAGED=0 NQE_SP_BLOCK=256 NQE_SP_R=16"; do
  env $env NQE_JIT_SYNC=1 python bench.py --workload c2_tree --no-cpu-baseline --no-configs --steps 20 --warmup 5 --details gpurun_out/ab_details.json 2>/dev/null | out "c2_tree [$env]"
done
