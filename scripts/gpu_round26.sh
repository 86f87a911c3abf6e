#!/bin/bash
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python scripts/variant_sweep.py cfg3p - 1:256:1:8:4:2:3073:0 > gpurun_out/r2_pk_cfg3p.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_pk_cfg3p.log
step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_tests.log
step timeout -k 10 500 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r2_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['hbm']['frac'], d['roofline']['valu']['frac'], d['config']['kernel_kind'], {k:(round(v['ms_per_step'],3),round(v['hbm_frac'],3),round(v['valu_frac'],3)) for k,v in d['others'].items()}); print(d['cpu_baseline']['value'], d['cpu_allcores']['value'], d['end_to_end']['pinned'], d['end_to_end']['pageable'])"
