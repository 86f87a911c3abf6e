"""Copies the summaries scripts/profile_round.sh left under gpurun_out/profiles_out into profiles/<round>/ and
rebuilds profiles/traffic_latest.json (what bench.py reports as roofline.traffic, per workload).
The factor 2 on FETCH_SIZE (MI355X_MICROARCH.md, HBM section, stated there for 16 B/lane) was calibrated here for the widths this
kernel uses: scripts/ubench_fetch.hip streams a known byte count with 16, 8 and 4 bytes per lane and FETCH_SIZE*1024 comes out
at exactly 0.5000 of the bytes for all three (scripts/fetch_cal.py), so the cs8 workload's traffic is reported too."""
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
    if name.startswith("bench_"):                           # bench.py lines of the same box: copied as they are
        shutil.copy(os.path.join(src, f), os.path.join(dst, name))
        continue
    shutil.copy(os.path.join(src, f), os.path.join(dst, name.replace(".json", "_summary.json") if f.endswith(".json") else name))
    if f.endswith(".json"):
        d = json.load(open(os.path.join(src, f)))
        wl = d["workload"]
        # the traffic figure may only describe the kernel bench.py times for this workload: the bench line of the same box
        # (config.kernel_flags, baked-taps bit excluded: it does not change the memory traffic) against the profiled kernel's name
        bline = os.path.join(src, f"{rnd}_bench_{'default' if wl == 'cfg3p' else wl}.json")
        ident = d.get("kernel_identity")
        if os.path.exists(bline) and ident is not None:
            try:
                bj = json.loads(open(bline).read().strip().splitlines()[-1])
                bflags = bj["config"].get("kernel_flags")
                if bj["config"]["workload"].split(":")[0] == wl and bflags is not None and (bflags & ~2) != (ident["flags"] & ~2):
                    print(f"REFUSED {wl}: profiled kernel flags {ident['flags']} != bench kernel flags {bflags}: traffic not published", file=sys.stderr)
                    continue
            except Exception as e:
                print(f"{wl}: cannot compare the profiled kernel with the bench line ({e}): traffic not published", file=sys.stderr)
                continue
        elif ident is None:
            print(f"{wl}: summary carries no kernel identity: traffic not published", file=sys.stderr)
            continue
        calibrated = True                                     # 16 and 8 B per lane both calibrated (see the docstring)
        traffic["workloads"][wl] = {
            "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch") if calibrated else None,
            "kernel_avg_ns": (d.get("timed_region") or {}).get("kernel_avg_ns") or (float(d["kernel_stats"][0]["AverageNs"]) if d["kernel_stats"] else None),
            "kernel_avg_ns_all_launches": float(d["kernel_stats"][0]["AverageNs"]) if d["kernel_stats"] else None,
            "source": f"{dst}/{name.replace('.json', '_summary.json')}" + ("" if calibrated else " (8 B/lane loads: FETCH_SIZE uncalibrated, traffic not reported)"),
        }
json.dump(traffic, open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
