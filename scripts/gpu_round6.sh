#!/bin/bash
mkdir -p gpurun_out
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_stamp.so
for wg in 1 2 4; do
  QD_WG_PER_CU=$wg timeout -k 10 120 python bench.py --workload cfg3p --samples-log2 29 --steps 6 --warmup 2 --no-cpu-baseline --no-others 2> gpurun_out/st_wg$wg.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wg=$wg', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'])"
  grep -A6 "stamps" gpurun_out/st_wg$wg.err | tail -7
done 2>&1 | tee gpurun_out/r2_wgstamps_cfg3p.log
