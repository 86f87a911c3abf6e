#!/bin/bash
mkdir -p gpurun_out
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
for skip in 15 3 2 12; do
  echo "== QD_DEBUG_SKIP=$skip"
  QD_DEBUG_SKIP=$skip timeout -k 10 200 bash scripts/power_probe.sh cfg3p | awk 'NR%3==0 || /bench|^[0-9]/'
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done 2>&1 | tee gpurun_out/r2_power_ablate2.log
