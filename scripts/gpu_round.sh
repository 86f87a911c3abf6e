#!/bin/bash
# One gpurun call: GPU test suite, default bench line, phase stamps, one-GPU rehearsal of the multi-rank launch path.
# A step that was killed (timeout) ends the call; a failing assertion does not.
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_tests.log
step timeout -k 10 400 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r2_bench.json
for wl in cfg3p cfg2; do
  QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_stamp.so step timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-others > gpurun_out/r2_stamp_$wl.json 2> gpurun_out/r2_stamp_$wl.err
  echo "stamps $wl rc=$?"; grep -A12 "stamps" gpurun_out/r2_stamp_$wl.err | tail -14
done
step timeout -k 10 300 python bench.py --gpus 2 --rehearse --samples-log2 27 --no-others > gpurun_out/r2_rehearse2.json 2> gpurun_out/r2_rehearse2.err; echo "rehearse rc=$?"; cut -c1-600 gpurun_out/r2_rehearse2.json
