#!/bin/bash
# Diagnostic build with per-phase cycle stamps (-DQD_STAMP): quadrs_amd/libquadrs_hip_stamp.so, used through
# QD_LIB_PATH.  It sits next to the main library so that the plan-time compiler finds csrc/.
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-fast-math -I include -DQD_STAMP"
mkdir -p build/obj
/opt/rocm/bin/hipcc $F -c -o build/obj/stamp_main.o quadrs_amd/csrc/quadrs_hip.hip &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c -o build/obj/stamp_longfir.o quadrs_amd/csrc/qd_longfir.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o quadrs_amd/libquadrs_hip_stamp.so build/obj/stamp_main.o build/obj/stamp_longfir.o -lhiprtc -ldl
echo quadrs_amd/libquadrs_hip_stamp.so
