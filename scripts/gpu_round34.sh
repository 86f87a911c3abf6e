#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
run() { # skip, fir_abl
  QD_DEBUG_SKIP=$1 QD_JIT_FLAGS="-DQD_FIR_ABL=$2" step timeout -k 10 200 python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-others 2> gpurun_out/abl.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$1 fir_abl=$2', 'kernel_ms=%.3f'%d['roofline']['kernel_ms'], d['config'].get('kernel_kind'))"
}
{ run 0 0; run 1 0; run 0 4; run 1 4; run 12 0; run 13 4; run 45 4; } 2>&1 | tee gpurun_out/r2_cfg3_ablate.log
