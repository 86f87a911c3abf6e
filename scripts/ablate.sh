#!/bin/bash
# timing-only ablation of the chain kernel phases (results are wrong by construction); needs the development library
# (python quadrs_amd/build.py --dev): the shipped one has no ablation bits
cd "$(dirname "$0")/.."
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
for skip in 0 1 2 3 4 8 7 15; do
  QD_DEBUG_SKIP=$skip python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-others "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$skip', 'kernel_ms=%.4f'%d['roofline']['kernel_ms'], 'GB/s=%.0f'%d['roofline']['hbm']['achieved'])"
done
