#!/bin/bash
# The round-end validation in one gpurun call: GPU suite under the three kernel policies, default bench, 2-rank rehearsal.
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_tests.log
step timeout -k 10 500 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r2_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['hbm']['frac'], d['roofline']['valu']['frac'], d['config']['kernel_kind'], {k:(round(v['ms_per_step'],3),round(v['hbm_frac'],3),round(v['valu_frac'],3)) for k,v in d['others'].items()}); print(d['cpu_baseline']['value'], d['cpu_allcores']['value'], d['end_to_end']['pinned'], d['end_to_end']['pageable'])"
step timeout -k 10 300 python bench.py --gpus 2 --rehearse --samples-log2 28 --no-others > gpurun_out/r2_rehearse2.json 2> gpurun_out/r2_rehearse2.err; echo "rehearse rc=$?"; cut -c1-300 gpurun_out/r2_rehearse2.json
QD_NO_FIXED=1 step timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests_nofixed.log 2>&1; echo "pytest nofixed rc=$?"; tail -2 gpurun_out/r2_tests_nofixed.log
QD_JIT=1 step timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests_jit.log 2>&1; echo "pytest jit rc=$?"; tail -2 gpurun_out/r2_tests_jit.log
