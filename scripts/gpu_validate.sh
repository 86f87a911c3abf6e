#!/bin/bash
# The round-end validation in one gpurun call (about 15 GPU-minutes): the GPU suite under the three kernel policies, the default bench,
# a six-rank rehearsal on the one GPU (the box admits at most six processes on the card), the lowpass-free fuzzer and the determinism soak.
mkdir -p gpurun_out
step() { "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "step killed rc=$rc: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/v_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/v_tests.log
step timeout -k 10 500 python bench.py > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/v_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['hbm']['frac'], d['roofline']['valu']['frac'], {k:(round(v['ms_per_step'],3),round(v['hbm_frac'],3),round(v['valu_frac'],3)) for k,v in d['others'].items()}); print({k: round(v['ms'], 3) for k, v in d['no_lowpass']['shapes'].items()})"
step timeout -k 10 300 python bench.py --gpus 6 --rehearse --samples-log2 26 > gpurun_out/v_rehearse6.json 2> gpurun_out/v_rehearse6.err; echo "rehearse rc=$?"; cut -c1-200 gpurun_out/v_rehearse6.json
step timeout -k 10 300 python scripts/fuzz_nofir.py 300 31 > gpurun_out/v_fuzz_nofir.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/v_fuzz_nofir.log
step timeout -k 10 300 python scripts/determinism_soak.py 100 > gpurun_out/v_soak.log 2>&1; echo "soak rc=$?"; grep -c "differing runs: 0" gpurun_out/v_soak.log
QD_NO_FIXED=1 step timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/v_tests_nofixed.log 2>&1; echo "pytest nofixed rc=$?"; tail -2 gpurun_out/v_tests_nofixed.log
QD_JIT=1 step timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/v_tests_jit.log 2>&1; echo "pytest jit rc=$?"; tail -2 gpurun_out/v_tests_jit.log
