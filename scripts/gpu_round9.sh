#!/bin/bash
mkdir -p gpurun_out
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_wgtime.so
for tune in "" "1:256:1:8:4:1:769:0"; do
  for lg in 29 31; do
    echo "== tune=$tune log2=$lg"
    QD_TUNE=$tune timeout -k 10 200 python bench.py --workload cfg3p --samples-log2 $lg --steps 4 --warmup 1 --no-cpu-baseline --no-others 2>&1 >/dev/null | grep wgtime | tail -2
  done
done 2>&1 | tee gpurun_out/r2_wgtime.log
