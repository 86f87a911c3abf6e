#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2_tests.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p --log2 29 - 1:256:1:8:4:1:1:0 1:256:1:8:4:1:2:0 1:256:1:8:4:1:3:0 1:256:1:8:4:1:3:3 1:256:1:8:4:1:4:3 1:256:1:8:4:1:8:3 > gpurun_out/r2_sweep_cfg3p.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/r2_sweep_cfg3p.log | tail -24
step timeout -k 10 300 python scripts/variant_sweep.py cfg2 --reps 40 - 2:256:1:8:4:1:1:0 2:256:1:8:4:1:2:0 2:256:1:8:4:1:3:3 2:256:1:8:4:1:4:3 1:256:1:8:4:1:4:0 > gpurun_out/r2_sweep_cfg2.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/r2_sweep_cfg2.log | tail -20
step timeout -k 10 400 python bench.py --no-others > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r2_bench.json
