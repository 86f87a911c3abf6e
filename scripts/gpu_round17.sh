#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python scripts/variant_sweep.py cfg3 --rounds 3 --reps 3 - 27:1024:1:8:4:2:1:0 27:1024:1:8:4:2:4097:0 27:512:1:8:2:2:1:0 27:512:1:8:2:2:4097:0 > gpurun_out/r2_sweep_cfg3_a.log 2>&1; echo "sweep rc=$?"; tail -12 gpurun_out/r2_sweep_cfg3_a.log
timeout -k 10 500 python scripts/variant_sweep.py cfg4 --rounds 3 --reps 3 - 1:1024:2:4:4:2:1:0 > gpurun_out/r2_sweep_cfg4_a.log 2>&1; echo "sweep rc=$?"; tail -5 gpurun_out/r2_sweep_cfg4_a.log
