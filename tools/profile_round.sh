#!/bin/bash
# Runs on the GPU box (via gpurun): every bench workload + rocprofv3 kernel stats + PMC traffic passes for the
# headline.  Output under gpurun_out/$1/ ; tools/summarize_profiles.py turns it into profiles/$1/.
#   usage: tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py 2>&1 | tail -1 > $OUT/bench_headline.json
# the profiled run of the headline right behind the plain one (the kernel slows by a few per cent as the box warms up)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_headline -o headline -- python $R/bench.py --workload headline --no-cpu-baseline --steps 20 --warmup 3 > $OUT/prof_headline.log 2>&1)
python bench.py --random-keys --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_headline_random_keys.json
python bench.py --workload c3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_c3.json
python bench.py --workload c3 --random-keys --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_c3_random_keys.json
python bench.py --workload c2 2>&1 | tail -1 > $OUT/bench_c2.json
python bench.py --workload c4 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_c4.json
cd /tmp
for w in c2 c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 3 > $OUT/prof_$w.log 2>&1
done
# PMC passes: counters in their own runs, kernel-trace only (never combined with other trace domains)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o headline -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o headline -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $R
ls $OUT
