#!/bin/bash
# Development: A/B two (or more) builds of the library on the same box (boxes differ by a few %):
#   git stash; python quadrs_amd/build.py; cp quadrs_amd/libquadrs_hip.so quadrs_amd/libquadrs_hip_A.so; git stash pop; python quadrs_amd/build.py
#   gpurun -- scripts/ab_libs.sh cfg2 quadrs_amd/libquadrs_hip_A.so quadrs_amd/libquadrs_hip.so
wl=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    QD_LIB_PATH=$PWD/$lib python bench.py --workload $wl --no-cpu-baseline --no-others 2>/dev/null > /tmp/ab_line.json
    python - "$wl" "$lib" <<'PY'
import sys, json
d = json.loads(open("/tmp/ab_line.json").read().strip().splitlines()[-1])
pw = d["roofline"].get("power", {})
print(sys.argv[1], sys.argv[2], round(d["ms_per_step"], 4), "hbm", round(d["roofline"]["hbm"]["frac"], 3), "sclk", pw.get("sclk_mhz_mean"), "W", pw.get("watts_mean"), flush=True)
PY
  done
done
