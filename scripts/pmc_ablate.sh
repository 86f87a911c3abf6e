#!/bin/bash
# VALU/SALU/LDS instruction counts per ablation setting (timing-only builds of the same binary)
export TMPDIR=/tmp
for skip in 0 1 2 4 8 15; do
  rm -rf gpurun_out/pmc_ab_$skip
  QD_DEBUG_SKIP=$skip rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_ab_$skip -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_ab_$skip/*/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_chain' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('skip=$skip', {k: '%.4g' % (sum(v)/len(v)) for k, v in sorted(agg.items())})
PY
done
