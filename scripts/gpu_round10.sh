#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_tests.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p --log2 29 - 1:256:1:8:4:1:1:0 1:256:1:8:4:1:769:0 1:256:1:8:4:1:769:3 > gpurun_out/r2_sweep_cfg3p_d.log 2>&1; echo "sweep rc=$?"; tail -5 gpurun_out/r2_sweep_cfg3p_d.log
step timeout -k 10 300 python scripts/variant_sweep.py cfg2 --reps 40 - 2:256:1:8:4:1:1:0 > gpurun_out/r2_sweep_cfg2_d.log 2>&1; tail -2 gpurun_out/r2_sweep_cfg2_d.log
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_wgtime.so
for tune in "" "1:256:1:8:4:1:769:0"; do
    echo "== tune=$tune"
    QD_TUNE=$tune timeout -k 10 200 python bench.py --workload cfg3p --steps 4 --warmup 1 --no-cpu-baseline --no-others 2>&1 >/dev/null | grep wgtime | tail -2
done 2>&1 | tee gpurun_out/r2_wgtime2.log
unset QD_LIB_PATH
step timeout -k 10 400 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r2_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['hbm']['frac'], {k:(v['ms_per_step'],round(v['hbm_frac'],3),round(v['valu_frac'],3)) for k,v in d['others'].items()})"
