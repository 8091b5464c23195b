set -u
R=$PWD; OUT=$R/gpurun_out/pmc_study; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|VALUBusy|SALUBusy|VALUUtilization|MemUnitBusy|MemUnitStalled|LDSBankConflict|FetchSize|WriteSize|L2CacheHit|Wavefronts|VALUInsts|SALUInsts|LDSInsts|VFetchInsts)" | sort -u | head -80 > $OUT/avail.txt
for wl in ${NQE_PMC_WORKLOADS:-"headline_single:--workload_headline_single" "headline:--workload_headline" "agg65536:--workload_agg_groups_--groups_65536"}; do
  name=${wl%%:*}; args=$(echo ${wl#*:} | tr '_' ' ' | sed 's/headline single/headline_single/; s/agg groups/agg_groups/')
  for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${name}_$tag -o x -- python $R/bench.py $args --no-configs --no-cpu-baseline --steps 2 --warmup 1 > $OUT/${name}_$tag.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('OUT','/root/repo/gpurun_out/pmc_study')
res=collections.defaultdict(dict)
for f in glob.glob(out+'/*/x_counter_collection.csv'):
    name=os.path.basename(os.path.dirname(f)).split('_SQ')[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if any(s in k for s in ('agg_grouped_fast','agg_slab_scatter','agg_slab_segments','agg_range_segments')):
            acc[([x for x in ('agg_grouped_fast','agg_slab_scatter','agg_slab_segments','agg_range_segments') if x in k][0], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in acc.items():
        res[(name,k)][c]=sum(v)/len(v)
for k,v in sorted(res.items()): print(k, {a: '%.4g'%b for a,b in sorted(v.items())})
PY
