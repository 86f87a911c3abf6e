#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
export QD_TUNE=1:1024:2:4:4:2:32769:0
for abl in 0 1 2 3; do
  QD_JIT_FLAGS="-DQD_FIR_ABL=$abl" step timeout -k 10 200 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-others 2> gpurun_out/abl.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fir_abl=$abl', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], d['config'].get('kernel_kind'))"
done 2>&1 | tee gpurun_out/r2_cfg4_ablate.log
