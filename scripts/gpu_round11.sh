#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python scripts/variant_sweep.py cfg3p --rounds 6 --reps 6 - 1:256:1:8:4:1:1:0 1:256:1:8:4:1:769:0 1:256:1:8:4:1:769:3 1:256:1:8:4:1:257:3 1:256:1:8:4:1:1:3 > gpurun_out/r2_sweep_cfg3p_e.log 2>&1; echo "sweep rc=$?"; tail -7 gpurun_out/r2_sweep_cfg3p_e.log
