#!/bin/bash
# A/B of the C2 kernels on one box: bench.py's c2 / c2_random workloads under the switches of selection.hip
out() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$1', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms_per_step'], 'frac %.3f'%r['frac'], {k:round(v['ms_per_step'],4) for k,v in json.load(open('gpurun_out/ab_details.json'))['main']['kernels'].items()})
"; }
for wl in c2_random c2; do
  for env in "NQE_COMPACT_STAGED=0 NQE_KEEP_TILE=0" "NQE_COMPACT_STAGED=1 NQE_KEEP_TILE=0" "NQE_COMPACT_STAGED=0 NQE_KEEP_TILE=1" "NQE_COMPACT_STAGED=1 NQE_KEEP_TILE=1" "NQE_COMPACT_STAGED_WGS=3"; do
    env $env python bench.py --workload $wl --no-cpu-baseline --no-configs --steps 20 --warmup 5 --details gpurun_out/ab_details.json 2>/dev/null | out "$wl [$env]"
  done
done
