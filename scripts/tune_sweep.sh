#!/bin/bash
# Development: time one bench workload under several QD_TUNE=G:NT:FIRR:FIRB tilings (plan-time builds).
# usage: scripts/tune_sweep.sh <workload> <tuning> [<tuning> ...]     ("-" = the built-in choice)
wl=$1; shift
for t in "$@"; do
    if [ "$t" = "-" ]; then unset QD_TUNE; else export QD_TUNE=$t; fi
    python bench.py --workload $wl --no-cpu-baseline --steps 6 --warmup 2 2> gpurun_out/tune_err.log | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$t', round(d['ms_per_step'],3), 'ms', round(d['roofline']['achieved'],1), 'GB/s', flush=True)" \
        || { echo "$wl $t FAILED"; tail -3 gpurun_out/tune_err.log; }
done
