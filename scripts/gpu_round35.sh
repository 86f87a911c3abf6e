#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg2 --rounds 5 --reps 20 - 2:256:1:8:4:2:3073:0 1:256:1:8:4:2:19458:0 2:256:1:8:4:2:1025:0 1:256:1:8:4:2:3073:0 2:256:1:8:4:2:2049:0 > gpurun_out/r2_cfg2_shapes.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg2_shapes.log | tail -13
