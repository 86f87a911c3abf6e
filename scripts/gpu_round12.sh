#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python scripts/variant_sweep.py cfg3p --rounds 5 --reps 6 - 1:256:1:8:4:1:1:0 1:256:1:8:4:2:1025:0 1:256:1:8:4:2:1537:0 1:256:1:8:4:2:1025:3 1:256:1:8:4:2:1027:0 > gpurun_out/r2_sweep_cfg3p_f.log 2>&1; echo "sweep rc=$?"; tail -14 gpurun_out/r2_sweep_cfg3p_f.log
