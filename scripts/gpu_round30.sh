#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 500 python scripts/variant_sweep.py cfg3 --rounds 3 --reps 5 27:1024:1:8:4:2:8705:0 12:512:1:8:4:2:8705:0 11:512:1:8:4:2:8705:0 10:512:1:8:4:2:8705:0 8:512:1:8:4:2:8705:0 8:256:1:8:4:2:8705:0 > gpurun_out/r2_cfg3_g.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_cfg3_g.log | tail -13
