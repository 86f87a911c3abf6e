#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p --log2 29 - 1:256:1:8:4:1:1:0 1:256:1:8:4:1:257:0 1:256:1:8:4:1:769:0 1:256:1:8:4:1:769:3 1:256:1:8:4:1:771:0 > gpurun_out/r2_sweep_cfg3p_b.log 2>&1; echo "sweep rc=$?"; tail -22 gpurun_out/r2_sweep_cfg3p_b.log
step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2_tests.log
