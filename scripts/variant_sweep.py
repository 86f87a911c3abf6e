"""Development: A/B tiling variants of one workload in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
usage: python scripts/variant_sweep.py <workload> [--log2 N] [--rounds R] [--reps K] hint hint ...
  hint = "-" (the library's own choice) or G:NT:FIRR:FIRB:LB:PAD:BATCH:WGPERCU (qd_plan_options.tile_hint)
Every variant's output is compared bit for bit with the first variant's; prints median / min ms per pass and the HBM fraction."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("hints", nargs="+")
ap.add_argument("--log2", type=int, default=None)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
cfg = dict(bench.WORKLOADS[a.workload])
if a.log2:
    cfg["n"] = 1 << a.log2
dev = torch.device("cuda", 0)
if a.workload == "cfg4":
    src = torch.empty(cfg["n"], 2, dtype=torch.float32, device=dev)
    tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
    for off in range(0, cfg["n"], 1 << 28):
        Q.gen_device(tones, cfg["sr"], off, src[off:off + (1 << 28)])
else:
    src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
torch.cuda.synchronize()
plans, outs = [], []
for h in a.hints:
    kw = {} if h == "-" else dict(tile_hint=[int(v) for v in h.split(":")])
    t0 = time.perf_counter()
    try:
        p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"], **kw)
    except Q.QuadrsError as e:
        print(f"{h}: plan failed: {e}", flush=True)
        plans.append(None); outs.append(None)
        continue
    o = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, o)
    torch.cuda.synchronize()
    print(f"{h}: plan {time.perf_counter() - t0:.1f} s, kind {p.info.kernel_kind}, G {p.info.tile_windows}, threads {p.info.threads}, lds {p.info.lds_bytes}", flush=True)
    plans.append(p); outs.append(o)
base = next(o for o in outs if o is not None)
for h, o in zip(a.hints, outs):
    if o is not None and o is not base:
        same = bool(torch.equal(o.view(torch.int32), base.view(torch.int32)))
        print(f"{h}: output == first variant's: {same}", flush=True)
for o in outs[1:]:
    del o
times = {h: [] for h in a.hints}
sensors = bench.GpuSensors.for_torch_device(torch, 0)
power = {h: [] for h in a.hints}
for r in range(a.rounds):
    for h, p, o in zip(a.hints, plans, outs):
        if p is None:
            continue
        for _ in range(2):
            p.run_device(src, o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            p.run_device(src, o)
        e1.record()
        torch.cuda.synchronize()
        times[h].append(e0.elapsed_time(e1) / a.reps)
        if sensors.ok() and r == a.rounds - 1:           # power / clock under this variant: ~0.7 s of back-to-back launches
            t_end = time.perf_counter() + 0.7
            while time.perf_counter() < t_end:
                for _ in range(16):
                    p.run_device(src, o)
                w, mhz = sensors.read()
                torch.cuda.synchronize()
                power[h].append((w, mhz))
bps = bench.BPS[cfg["fmt"]]
for h, p in zip(a.hints, plans):
    if p is None:
        continue
    t = np.array(times[h])
    alg = cfg["n"] * bps + p.n_windows * cfg["W"] * 4
    pw = power[h][len(power[h]) // 2:]
    ptxt = f"  {np.mean([x[0] for x in pw]):6.0f} W {np.mean([x[1] or 0 for x in pw]):5.0f} MHz" if pw else ""
    print(f"{a.workload} {h:28s} median {np.median(t):8.4f} ms  min {t.min():8.4f} ms  hbm_frac(median) {alg / (np.median(t) * 1e-3) / 8e12:.3f}{ptxt}", flush=True)
