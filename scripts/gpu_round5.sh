#!/bin/bash
mkdir -p gpurun_out
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_stamp.so
for skip in 0 1 2 15; do
  QD_DEBUG_SKIP=$skip timeout -k 10 120 python bench.py --workload cfg3p --steps 6 --warmup 2 --no-cpu-baseline --no-others 2> gpurun_out/st_$skip.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$skip', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'])"
  grep -A2 "stamps" gpurun_out/st_$skip.err | tail -3
done 2>&1 | tee gpurun_out/r2_clock_cfg3p.log
# workgroups-per-CU scaling of the standard and the planar+baked variant
unset QD_LIB_PATH
timeout -k 10 300 python scripts/variant_sweep.py cfg3p --log2 29 --rounds 3 1:256:1:8:4:1:1:1 1:256:1:8:4:1:1:2 1:256:1:8:4:1:1:3 1:256:1:8:4:1:1:4 1:256:1:8:4:1:769:1 1:256:1:8:4:1:769:2 1:256:1:8:4:1:769:3 1:256:1:8:4:1:769:4 2>&1 | tail -9 | tee gpurun_out/r2_wgscale_cfg3p.log
