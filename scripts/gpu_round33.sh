#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 120 python scripts/variant_sweep.py cfg4 --log2 27 --rounds 2 --reps 3 - 1:1024:2:4:4:2:49154:0 > gpurun_out/r2_cfg4_defer_small.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_defer_small.log | tail -3
grep -q "first variant's: True" gpurun_out/r2_cfg4_defer_small.log || { echo "MISMATCH at small size; stop"; exit 1; }
step timeout -k 10 400 python scripts/variant_sweep.py cfg4 --rounds 3 --reps 5 - 1:1024:2:4:4:2:49154:0 > gpurun_out/r2_cfg4_defer.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_defer.log | tail -3
