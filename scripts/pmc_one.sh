#!/bin/bash
# usage: scripts/pmc_one.sh <workload> [bench.py args...] : SQ/LDS counters of the chain kernel for one workload (QD_TUNE selects a variant)
export TMPDIR=/tmp
wl=$1; shift
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/pmc_one
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_one -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-others "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_one/*/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_chain' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$wl', '${QD_TUNE:-builtin}', {k: '%.4g' % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
done
rm -rf gpurun_out/pmc_one
