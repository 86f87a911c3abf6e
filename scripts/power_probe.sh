#!/bin/bash
# Development: board power / clocks while the chain kernel runs (rocm-smi sampled once a second against a long bench run).
mkdir -p gpurun_out
wl=${1:-cfg3p}; steps=${2:-4000}; shift; shift
timeout -k 10 150 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --no-others "$@" > gpurun_out/power_bench_$wl.json 2> gpurun_out/power_bench_$wl.err &
pid=$!
t0=$(date +%s)
while kill -0 $pid 2>/dev/null; do
  echo "t=$(( $(date +%s) - t0 )) $(rocm-smi --showpower --showclocks 2>&1 | grep -i "package power\|sclk\|fclk" | sed -e 's/.*: //' | tr '\n' ' ')"
  sleep 1
done
wait $pid; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/power_bench_$wl.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['hbm']['frac'])"
