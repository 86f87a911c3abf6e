#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_stamp.so
for tune in 1:1024:2:4:4:2:32769:0 1:1024:2:4:4:2:49154:0; do
  echo "== $tune"
  QD_TUNE=$tune step timeout -k 10 300 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-others 2> gpurun_out/stamp.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms=%.4f'%d['roofline']['kernel_ms'], d['config'].get('kernel_kind'))"
  grep -A17 "stamps" gpurun_out/stamp.err | tail -18
done 2>&1 | tee gpurun_out/r2_stamp_cfg4b.log
