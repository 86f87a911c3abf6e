"""Copies the summaries scripts/profile_round.sh left under gpurun_out/profiles_out into profiles/<round>/ and
rebuilds profiles/traffic_latest.json (what bench.py reports as roofline.traffic, per workload).
cs8/cu8/cs16 workloads load 8 B per lane: FETCH_SIZE is uncalibrated for that width on gfx950
(MI355X_MICROARCH.md, HBM section), so their traffic stays null."""
import json, os, shutil, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/profiles_out", f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
traffic = {"workloads": {}, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; hbm bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
           "(gfx950 reports half of a 16 B/lane streaming read); per launch, like roofline.achieved"}
for f in sorted(os.listdir(src)):
    if not f.startswith(rnd + "_"):
        continue
    name = f[len(rnd) + 1:]
    shutil.copy(os.path.join(src, f), os.path.join(dst, name.replace(".json", "_summary.json") if f.endswith(".json") else name))
    if f.endswith(".json"):
        d = json.load(open(os.path.join(src, f)))
        wl = d["workload"]
        calibrated = wl in ("cfg2", "cfg3p", "cfg4")          # cf32 sources: 16 B per lane
        traffic["workloads"][wl] = {
            "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch") if calibrated else None,
            "kernel_avg_ns": (d.get("timed_region") or {}).get("kernel_avg_ns") or (float(d["kernel_stats"][0]["AverageNs"]) if d["kernel_stats"] else None),
            "kernel_avg_ns_all_launches": float(d["kernel_stats"][0]["AverageNs"]) if d["kernel_stats"] else None,
            "source": f"{dst}/{name.replace('.json', '_summary.json')}" + ("" if calibrated else " (8 B/lane loads: FETCH_SIZE uncalibrated, traffic not reported)"),
        }
json.dump(traffic, open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
