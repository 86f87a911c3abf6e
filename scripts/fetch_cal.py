"""Summarises scripts/ubench_fetch.hip's rocprofv3 --pmc FETCH_SIZE run: counter / true bytes per load width."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_read" in r["Kernel_Name"]:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
true_bytes = 2 << 30
for k, v in sorted(agg.items()):
    m = sum(v) / len(v)
    print(f"{k[:60]:60s} FETCH_SIZE {m:.0f} KiB  -> counter*1024 / bytes = {m * 1024 / true_bytes:.4f}")
