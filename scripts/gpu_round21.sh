#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p 1:256:1:8:4:2:3585:0 1:256:1:8:4:2:3073:0 > gpurun_out/r2_asm_cfg3p.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_asm_cfg3p.log
step timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "cfg3p or north or fixed" > gpurun_out/r2_asm_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_asm_tests.log
