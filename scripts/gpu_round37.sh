#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 200 python scripts/variant_sweep.py cfg3 --log2 26 --rounds 2 --reps 3 27:1024:1:8:4:2:8705:0 27:1024:1:8:4:2:25089:0 > gpurun_out/r2_cfg3_defer_small.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg3_defer_small.log | tail -3
grep -q "first variant's: True" gpurun_out/r2_cfg3_defer_small.log || { echo "MISMATCH at small size; stop"; exit 1; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg3 --rounds 3 --reps 5 27:1024:1:8:4:2:8705:0 27:1024:1:8:4:2:25089:0 > gpurun_out/r2_cfg3_defer.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg3_defer.log | tail -3
