#!/bin/bash
mkdir -p gpurun_out
for spec in "cfg2 40000" "cfg3 400" "cfg4 300"; do
  set -- $spec
  echo "== $1"
  timeout -k 10 250 bash scripts/power_probe.sh $1 $2 | awk 'NR%3==0 || /bench|^[0-9]/'
  rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
done 2>&1 | tee gpurun_out/r2_power_others.log
