#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 120 python scripts/variant_sweep.py cfg3p --log2 26 --rounds 2 --reps 3 1:256:1:8:4:2:3073:0 1:256:1:8:4:2:19458:0 > gpurun_out/r2_defer_small.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_defer_small.log
grep -q "first variant's: True" gpurun_out/r2_defer_small.log || { echo "MISMATCH or failure at small size; stop"; exit 1; }
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p 1:256:1:8:4:2:3073:0 1:256:1:8:4:2:19458:0 > gpurun_out/r2_defer_cfg3p.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_defer_cfg3p.log
