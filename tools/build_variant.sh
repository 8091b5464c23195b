#!/bin/bash
# A/B builds: compiles the library with extra preprocessor flags into naive_query_engine_amd/libnqe_hip_<name>.so (objects under
# csrc/_var_<name>/, git-ignored).  Run a tool against it with NQE_LIB_PATH=naive_query_engine_amd/libnqe_hip_<name>.so.
#   usage: tools/build_variant.sh <name> "<extra hipcc flags>"
set -e
NAME=$1; shift
FLAGS="$*"
cd "$(dirname "$0")/../naive_query_engine_amd/csrc"
D=_var_$NAME
mkdir -p $D
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXX="-O3 -std=c++17 -fPIC --offload-arch=gfx950 --offload-compress -munsafe-fp-atomics -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS"
pids=()
for f in context exchange arrow_c expr selection sort aggregate aggregate_tail aggregate_fast aggregate_partition aggregate_tiny hash_join strings csv; do
  ( $HIPCC $CXX -c $f.hip -o $D/$f.o ) & pids+=($!)
done
for p in 0 1 2 3; do for v in 0 1; do
  ( $HIPCC $CXX -DNQE_FAST_PRED=$p -DNQE_FAST_VNULL=$v -c aggregate_fast_inst.hip -o $D/aggregate_fast_p${p}_v${v}.o ) & pids+=($!)
done; done
for p in 4 5 6; do ( $HIPCC $CXX -DNQE_FAST_PRED=$p -DNQE_FAST_VNULL=0 -c aggregate_fast_inst.hip -o $D/aggregate_fast_p${p}_v0.o ) & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o ../libnqe_hip_$NAME.so $D/*.o -ldl
ls -la ../libnqe_hip_$NAME.so
