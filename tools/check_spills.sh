#!/bin/bash
# Reports every kernel of the fast aggregate instantiations that spills VGPRs (scratch traffic in a streaming kernel is a 2x cliff)
# or uses scratch memory without spilling (a by-value kernel argument indexed at run time lands there: 3.7x slower, found once).
cd "$(dirname "$0")/../naive_query_engine_amd/csrc"
for pv in 0:0 0:1 1:0 1:1 2:0 2:1 3:0 3:1 4:0; do p=${pv%:*}; v=${pv#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc $EXTRA -DNQE_FAST_PRED=$p -DNQE_FAST_VNULL=$v -c aggregate_fast_inst.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - | awk '$NF != 0 || $(NF-3) != 0' | sed 's/Function Name: _ZN3nqe3agg12_GLOBAL__N_1//' > /tmp/spills_${p}_${v}.txt ) &
done; wait
cat /tmp/spills_*.txt | cut -c1-200
echo "kernels with VGPR spills or scratch: $(cat /tmp/spills_*.txt | wc -l)"
