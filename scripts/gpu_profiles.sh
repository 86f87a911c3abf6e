#!/bin/bash
# rocprofv3 evidence for the round: kernel trace + PMC passes per workload (scripts/profile_round.sh), then the summaries
# usage: scripts/gpu_profiles.sh [round tag, default r03] [workloads...]
rnd=${1:-r04}; shift || true
wls=${@:-cfg3p cfg2 cfg3 cfg4}
mkdir -p gpurun_out
rm -rf gpurun_out/profiles_out
for wl in $wls; do
  timeout -k 10 500 bash scripts/profile_round.sh ${rnd}_$wl --workload $wl > gpurun_out/prof_${rnd}_$wl.txt 2>&1; echo "$wl rc=$?"
  if [ $? -ge 124 ]; then exit 1; fi
done
if [ -z "$NO_DEFAULT_BENCH" ]; then timeout -k 10 400 python bench.py > gpurun_out/profiles_out/${rnd}_bench_default.json 2> gpurun_out/${rnd}_bench_default.err; echo "bench rc=$?"; fi
for wl in $wls; do
  [ $wl = cfg3p ] && continue
  timeout -k 10 300 python bench.py --workload $wl --no-others > gpurun_out/profiles_out/${rnd}_bench_$wl.json 2>/dev/null; echo "bench $wl rc=$?"
done
ls -la gpurun_out/profiles_out
