#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests/test_gpu_robustness.py -m gpu -q -k "variant_flags" > gpurun_out/r2_tests_flags.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2_tests_flags.log
