#!/bin/bash
# A/B of the one-pass selection + projection kernel (expr_jit.hpp: nqe_jit_selproj): bench.py's c2_tree workload under its switches
out() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$1', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms_per_step'], 'frac %.3f'%r['frac'], {k:round(v['ms_per_step'],4) for k,v in json.load(open('gpurun_out/ab_details.json'))['main']['kernels'].items()})
"; }
# (one configuration per line in $ENV_FILE, default: the round-5 switches)
ENV_FILE=${ENV_FILE:-tools/ab_c2tree.envs}
while read -r env; do
  [ -z "$env" ] && continue
  env $env NQE_JIT_SYNC=1 python bench.py --workload c2_tree --no-cpu-baseline --no-configs --steps 20 --warmup 5 --details gpurun_out/ab_details.json 2>/dev/null | out "c2_tree [$env]"
done < "$ENV_FILE"
