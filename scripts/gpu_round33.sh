#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg4 --rounds 3 --reps 5 - 1:1024:2:4:4:2:49154:0 > gpurun_out/r2_cfg4_defer.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_defer.log | tail -3
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p - 1:256:1:8:4:2:3073:0 > gpurun_out/r2_pk_cfg3p_b.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_pk_cfg3p_b.log
