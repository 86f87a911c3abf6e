#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python scripts/variant_sweep.py cfg3p --rounds 5 --reps 6 - 1:256:1:8:4:2:3073:0 1:256:1:8:4:2:3585:0 1:256:1:8:4:1:2049:0 > gpurun_out/r2_sweep_cfg3p_h.log 2>&1; echo "sweep rc=$?"; tail -7 gpurun_out/r2_sweep_cfg3p_h.log
timeout -k 10 300 python scripts/variant_sweep.py cfg2 --reps 40 - 2:256:1:8:4:1:2049:0 2:256:1:8:4:2:3073:0 2:256:1:8:4:2:3585:0 > gpurun_out/r2_sweep_cfg2_h.log 2>&1; tail -7 gpurun_out/r2_sweep_cfg2_h.log
