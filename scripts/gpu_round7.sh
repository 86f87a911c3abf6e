#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2_tests.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p --log2 29 - 1:256:1:8:4:1:1:0 1:256:1:8:4:1:769:0 1:256:1:16:4:1:769:0 1:256:1:24:4:1:769:0 1:256:1:16:4:1:769:3 1:256:1:4:4:1:769:0 > gpurun_out/r2_sweep_cfg3p_c.log 2>&1; echo "sweep rc=$?"; tail -8 gpurun_out/r2_sweep_cfg3p_c.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg2 --reps 40 - 2:256:1:8:4:1:1:0 > gpurun_out/r2_sweep_cfg2_c.log 2>&1; echo "sweep rc=$?"; tail -3 gpurun_out/r2_sweep_cfg2_c.log
