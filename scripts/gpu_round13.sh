#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_tests.log
step timeout -k 10 500 python scripts/variant_sweep.py cfg3p --rounds 5 --reps 6 - 1:256:1:8:4:1:1:0 1:256:1:8:4:2:1025:0 1:256:1:8:4:2:1537:0 > gpurun_out/r2_sweep_cfg3p_g.log 2>&1; echo "sweep rc=$?"; tail -4 gpurun_out/r2_sweep_cfg3p_g.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg2 --reps 40 - 2:256:1:8:4:1:1:0 2:256:1:8:4:2:1025:0 2:256:1:8:4:2:1537:0 > gpurun_out/r2_sweep_cfg2_g.log 2>&1; tail -4 gpurun_out/r2_sweep_cfg2_g.log
