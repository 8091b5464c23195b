#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, rocprofv3 kernel stats per config, and the PMC traffic passes
# (counters in their own runs, --kernel-trace only, one counter per pass) for every config plus the calibration kernels.
# Output under gpurun_out/$1/ ; tools/summarize_profiles.py turns it into profiles/$1/ (tracked).
#   usage: tools/profile_round.sh r06        (NQE_PROFILE_LIGHT=1: the default line + the per-config rocprofv3 passes only)
set -u
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/csrc_rev.py > $OUT/csrc_rev.txt
LIGHT=${NQE_PROFILE_LIGHT:-}
if [ -z "$LIGHT" ]; then
# the same line with nothing remembered between executions (every execution plans from scratch; the key sample still runs)
NQE_NO_PLAN_HINTS=1 python bench.py --details $OUT/bench_details_no_plan_hints.json 2>&1 | tail -1 > $OUT/bench_no_plan_hints.json
# first-execution cost of every query shape, each in a fresh process
NQE_COLD_VARIANTS=NQE_NO_RESERVE,NQE_LAZY_MODULES python tools/probe_cold.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_cold.txt
fi
declare -A WL=( [headline]="--workload headline" [headline_random_keys]="--workload headline --random-keys" [c2]="--workload c2" [c3]="--workload c3" [c3_random_keys]="--workload c3 --random-keys" [c4]="--workload c4" \
                [c4_sparse_keys]="--workload c4_sparse" [agg_65536_groups]="--workload agg_groups --groups 65536" [agg_4096_groups]="--workload agg_groups --groups 4096" [agg_5000_groups]="--workload agg_groups --groups 5000" [agg_6000_groups]="--workload agg_groups --groups 6000" [agg_11000_groups]="--workload agg_groups --groups 11000" [agg_12000_groups_count_sum_avg]="--workload agg_groups --groups 12000 --no-minmax" [c2_expression_trees]="--workload c2_tree" \
                [headline_single_column]="--workload headline_single" [headline_int64_values]="--workload headline_int64" \
                [agg_tree_predicate]="--workload tree_pred" [agg_three_value_columns]="--workload agg3" [c2_random_ids]="--workload c2_random" \
                [c4_dup_keys]="--workload c4_dup" [c4_partial_match]="--workload c4_partial" [c4_dim_1e8]="--workload c4 --dim-rows 100000000" \
                [c4_dim_1e7]="--workload c4 --dim-rows 10000000" [c4_shared_probe_columns]="--workload c4 --immutable" [c4_wide_payload]="--workload c4_wide" \
                [agg_readme_shape]="--workload agg_readme" [headline_nullable]="--workload headline_nullable" [agg_1048576_groups]="--workload agg_groups --groups 1048576" )
CONFIGS=${NQE_PROFILE_CONFIGS:-"headline headline_random_keys c3 c3_random_keys headline_single_column headline_int64_values headline_nullable agg_tree_predicate agg_three_value_columns agg_readme_shape c2 c2_random_ids c2_expression_trees c4 c4_shared_probe_columns c4_wide_payload c4_sparse_keys c4_dup_keys c4_partial_match c4_dim_1e7 c4_dim_1e8 agg_4096_groups agg_5000_groups agg_6000_groups agg_11000_groups agg_12000_groups_count_sum_avg agg_65536_groups agg_1048576_groups"}
cd /tmp
for name in $CONFIGS; do
  args="${WL[$name]} --no-configs --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python $R/bench.py $args --steps 20 --warmup 3 > $OUT/prof_$name.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$name -o $name -- python $R/bench.py $args --steps 3 --warmup 1 > $OUT/pmc_fetch_$name.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$name -o $name -- python $R/bench.py $args --steps 3 --warmup 1 > $OUT/pmc_write_$name.log 2>&1
done
# calibration: known byte counts in the product kernels' access pattern (8-byte non-temporal loads / stores), and random reads
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_calib -o calib -- $R/tools/stream_bench calib > $OUT/pmc_fetch_calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_calib -o calib -- $R/tools/stream_bench calib > $OUT/pmc_write_calib.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_gather -o gather -- $R/tools/micro_bench gather1 > $OUT/pmc_fetch_gather.log 2>&1
cd $R
# the default line LAST among the measurements it quotes: summarised here, on the box, the PMC passes above are what its `roofline.traffic` is
# quoted from (bench.py only quotes a file made from the csrc revision that is running)
python tools/summarize_profiles.py $TAG > /dev/null 2>&1
python bench.py --details $OUT/bench_details.json 2>&1 | tail -1 > $OUT/bench_default.json
if [ -n "$LIGHT" ]; then ls $OUT; exit 0; fi
python tools/probe_paths.py groups 2>&1 | grep -v amdgpu.ids > $OUT/probe_groups.txt
python tools/probe_paths.py joinshapes 2>&1 | grep -v amdgpu.ids > $OUT/probe_joinshapes.txt
python tools/probe_build.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_build.txt
python tools/probe_paths.py keys 2>&1 | grep -v amdgpu.ids > $OUT/probe_keys.txt
python tools/probe_paths.py exprs 2>&1 | grep -v amdgpu.ids > $OUT/probe_exprs.txt
python tools/probe_paths.py trees 2>&1 | grep -v amdgpu.ids > $OUT/probe_trees.txt
python tools/probe_paths.py csv 2>&1 | grep -v amdgpu.ids > $OUT/probe_csv.txt
python tools/probe_paths.py strings 2>&1 | grep -v amdgpu.ids > $OUT/probe_strings.txt
# A/B of the round's switches on this box (two rounds each: the boxes differ by more than some of the effects)
{
  for round in 1 2; do
    for sw in "" NQE_NO_RANGE_TAIL=1 NQE_NO_RANGE_PARTITION=1; do
      for g in 65536 1048576; do
        ms=$(env $sw python bench.py --workload agg_groups --groups $g --no-configs --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
        echo "round $round  agg_groups --groups $g  ${sw:-default}: $ms ms per step"
      done
    done
  done
} > $OUT/probe_switches.txt 2>&1
./tools/micro_bench all > $OUT/micro_bench.txt 2>&1
ls $OUT
