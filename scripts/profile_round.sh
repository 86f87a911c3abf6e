#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against:
#   kernel trace + stats (average duration), then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
#   (TCC slots: both do not fit in one pass; never combined with trace domains other than kernel-trace).
#   The counter passes are restricted to the chain kernel (--kernel-include-regex): collecting counters over the thousands of
#   torch kernels that generate the 16 GiB synthetic slab crashed the profiler on cfg3 (round 3).
# usage: scripts/profile_round.sh <tag> [bench.py args...]
set -e
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-others --no-power "$@" > $out/kt.log 2>&1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-others --no-power "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-others --no-power "$@" > $out/write.log 2>&1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-others --no-power "$@" > $out/sq.log 2>&1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-others --no-power "$@" > $out/sq2.log 2>&1
python3 scripts/summarize_profile.py $out $tag "$@"
# the raw rocprofv3 trees are large (gpurun merges at most 64 MiB back): keep the summaries and logs only
rm -rf $out/kt $out/fetch $out/write $out/sq $out/sq2
