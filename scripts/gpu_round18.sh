#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python scripts/variant_sweep.py cfg3 --rounds 3 --reps 3 - 27:1024:1:8:4:2:8193:0 27:1024:1:8:4:2:8705:0 > gpurun_out/r2_sweep_cfg3_b.log 2>&1; echo "sweep rc=$?"; tail -8 gpurun_out/r2_sweep_cfg3_b.log
timeout -k 10 300 python scripts/variant_sweep.py cfg3p --rounds 3 --reps 4 - 1:256:1:8:4:2:3585:0 > gpurun_out/r2_sweep_cfg3p_j.log 2>&1; tail -3 gpurun_out/r2_sweep_cfg3p_j.log
