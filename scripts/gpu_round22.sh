#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_stamp.so
for skip in 0 1 2 15; do
  QD_DEBUG_SKIP=$skip step timeout -k 10 200 python bench.py --workload cfg3p --steps 10 --warmup 2 --no-cpu-baseline --no-others 2> gpurun_out/stamp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$skip', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], d['config'].get('kernel_kind'))"
  grep -A3 "stamps" gpurun_out/stamp.err
done 2>&1 | tee gpurun_out/r2_stamp2_cfg3p.log
