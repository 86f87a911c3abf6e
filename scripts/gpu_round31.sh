#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 200 python scripts/variant_sweep.py cfg4 --log2 27 --rounds 2 --reps 3 1:1024:2:4:4:2:1:0 1:1024:2:4:4:2:32769:0 > gpurun_out/r2_cfg4_pk_small.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_pk_small.log | tail -3
grep -q "first variant's: True" gpurun_out/r2_cfg4_pk_small.log || { echo "MISMATCH at small size; stop"; exit 1; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg4 --rounds 3 --reps 5 1:1024:2:4:4:2:1:0 1:1024:2:4:4:2:32769:0 > gpurun_out/r2_cfg4_pk.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_pk.log | tail -3
