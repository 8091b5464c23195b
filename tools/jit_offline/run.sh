#!/bin/bash
# Generates the run-time kernels' sources (gen_sources.hip), compiles each with hipRTC as the library does, prints per kernel:
#   <name> <compile seconds> vgprs=<n> vgpr_spills=<n> scratch=<bytes>
# No GPU needed (hipRTC compiles for gfx950 offline).  Needs ../../naive_query_engine_amd/libnqe_hip.so (make -C csrc).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT=${1:-/tmp/nqe_jit_offline}
mkdir -p "$OUT"
LIBDIR="$HERE/../../naive_query_engine_amd"
/opt/rocm/bin/hipcc -O1 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-function "$HERE/gen_sources.hip" -o "$OUT/gen_sources" \
    -L"$LIBDIR" -lnqe_hip -Wl,-rpath,"$LIBDIR" -ldl
g++ -O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$HERE/rtc_compile.cpp" -o "$OUT/rtc_compile" -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
"$OUT/gen_sources" > "$OUT/all_sources.txt"
cd "$OUT"
awk '/^\/\/==== /{ if (f) close(f); f = $2 ".hip"; opts[$2] = $3; print $2, $3 > "names.txt"; next } { print > f }' all_sources.txt
while read -r name opt; do
    t0=$(date +%s.%N)
    NQE_RTC_OUT="$OUT/$name.co" ./rtc_compile "$name.hip" $opt > "$name.log" 2>&1 || { echo "$name FAILED"; cat "$name.log"; exit 1; }
    t1=$(date +%s.%N)
    notes=$(/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT/$name.co" 2>/dev/null || true)
    v=$(echo "$notes" | grep -m1 "\.vgpr_count:" | awk '{print $2}')
    sp=$(echo "$notes" | grep -m1 "\.vgpr_spill_count:" | awk '{print $2}')
    sc=$(echo "$notes" | grep -m1 "\.private_segment_fixed_size:" | awk '{print $2}')
    printf "%s %ss vgprs=%s vgpr_spills=%s scratch=%s\n" "$name" "$(python3 -c "print(round($t1 - $t0, 2))")" "$v" "$sp" "$sc"
done < names.txt
