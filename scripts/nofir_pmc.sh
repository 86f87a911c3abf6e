#!/bin/bash
# development: SQ counters of one lowpass-free chain.  usage: scripts/nofir_pmc.sh TAG FMT SHIFT W [S] [log2]
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
python3 scripts/nofir_one.py "$@" > $out/run.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $out/a -- python3 scripts/nofir_one.py "$@" > $out/a.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex 'k_chain|k_spark' --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -- python3 scripts/nofir_one.py "$@" > $out/b.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open(out + "/run.log").read().strip().splitlines()[-1])
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v) / len(v):.4e}  ({len(v)} launches)")
PY
rm -rf $out/a $out/b
