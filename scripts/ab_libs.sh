#!/bin/bash
# Development: A/B two builds of the library on the same box (boxes differ by a few %):
#   git stash; python quadrs_amd/build.py; cp quadrs_amd/libquadrs_hip.so quadrs_amd/libquadrs_hip_A.so; git stash pop; python quadrs_amd/build.py
#   gpurun -- scripts/ab_libs.sh cfg2 quadrs_amd/libquadrs_hip_A.so quadrs_amd/libquadrs_hip.so
wl=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    QD_LIB_PATH=$PWD/$lib python bench.py --workload $wl --no-cpu-baseline --no-others 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$lib', round(d['ms_per_step'],4), round(d["roofline"]["hbm"]["frac"],3), flush=True)"
  done
done
