#!/bin/bash
# the many-group aggregate (bench.py --workload agg_groups) under the per-call switches of the aggregate operator
out() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$1', 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms_per_step'], 'frac %.3f'%r['frac'], {k:round(v['ms_per_step'],4) for k,v in json.load(open('gpurun_out/ab_details.json'))['main']['kernels'].items()})
"; }
for G in ${GROUPS_LIST:-65536 1048576}; do
  for env in ${ENV_LIST:-NQE_DEFAULT=1 NQE_NO_RANGE_PARTITION=1 NQE_NO_PLAN_HINTS=1}; do
    env $env python bench.py --workload agg_groups --groups $G --no-cpu-baseline --no-configs --steps 10 --warmup 3 --details gpurun_out/ab_details.json 2>/dev/null | out "G=$G [$env]"
  done
done
