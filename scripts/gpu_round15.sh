#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python scripts/variant_sweep.py cfg3p --rounds 5 --reps 6 1:256:1:8:4:2:3073:0 1:256:1:8:4:2:3074:0 1:256:1:8:4:2:3585:0 1:256:1:8:4:2:3586:0 1:256:1:8:4:2:3587:0 > gpurun_out/r2_sweep_cfg3p_i.log 2>&1; echo "sweep rc=$?"; tail -12 gpurun_out/r2_sweep_cfg3p_i.log
