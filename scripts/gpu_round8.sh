#!/bin/bash
mkdir -p gpurun_out
cd scripts && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_valu ubench_valu.hip && cd .. && timeout -k 10 120 /tmp/ubench_valu 2>&1 | tee gpurun_out/r2_ubench_valu.log
QD_TUNE=1:256:1:8:4:1:1:0 timeout -k 10 400 bash scripts/pmc_one.sh cfg3p --samples-log2 29 2>&1 | tee gpurun_out/r2_pmc_cfg3p_std.log
QD_TUNE=1:256:1:8:4:1:769:0 timeout -k 10 400 bash scripts/pmc_one.sh cfg3p --samples-log2 29 2>&1 | tee gpurun_out/r2_pmc_cfg3p_planar.log
