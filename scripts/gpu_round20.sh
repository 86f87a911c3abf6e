#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p 1:256:1:8:4:2:3585:0 1:256:1:8:4:2:19969:0 > gpurun_out/r2_rot_cfg3p.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_rot_cfg3p.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg2 2:256:1:8:4:1:1:0 2:256:1:8:4:1:16385:0 > gpurun_out/r2_rot_cfg2.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_rot_cfg2.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg3 27:1024:1:8:4:2:8705:0 27:1024:1:8:4:2:25089:0 > gpurun_out/r2_rot_cfg3.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_rot_cfg3.log
step timeout -k 10 400 python scripts/variant_sweep.py cfg4 1:1024:2:4:4:2:1:0 1:1024:2:4:4:2:16385:0 > gpurun_out/r2_rot_cfg4.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_rot_cfg4.log
