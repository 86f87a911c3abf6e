#!/bin/bash
mkdir -p gpurun_out
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
for skip in 0 1 2 3 4 8 12 14 15 32 47 63; do
  QD_DEBUG_SKIP=$skip timeout -k 10 120 python bench.py --workload cfg3p --samples-log2 29 --steps 10 --warmup 2 --no-cpu-baseline --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$skip', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], 'hbm_frac=%.3f'%d['roofline']['hbm']['frac'])"
done 2>&1 | tee gpurun_out/r2_ablate_cfg3p.log
