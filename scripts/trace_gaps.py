"""Development: from a rocprofv3 --kernel-trace csv, the durations of the chain kernel and the gaps between consecutive dispatches."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_chain" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
per = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(len(rows) - 1)]
print("last %d launches: kernel %.1f us (min %.1f), gap %.1f us (min %.1f max %.1f), period %.1f us" % (
    len(rows), sum(dur) / len(dur) / 1e3, min(dur) / 1e3, sum(gap) / len(gap) / 1e3, min(gap) / 1e3, max(gap) / 1e3, sum(per) / len(per) / 1e3))
