#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg4 --rounds 3 --reps 5 1:1024:2:4:4:2:1:0 1:512:4:4:4:2:1:0 1:512:2:4:4:2:1:0 1:512:4:8:4:2:1:0 1:256:4:4:4:2:1:0 1:1024:4:4:4:2:1:0 > gpurun_out/r2_cfg4_shapes.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg4_shapes.log | tail -14
step timeout -k 10 500 python scripts/variant_sweep.py cfg3 --rounds 3 --reps 5 27:1024:1:8:4:2:8705:0 13:512:1:8:4:2:8705:0 13:1024:1:8:4:2:8705:0 6:256:1:8:4:2:8705:0 29:1024:1:8:4:2:8705:0 > gpurun_out/r2_cfg3_shapes.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg3_shapes.log | tail -12
