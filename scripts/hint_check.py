"""Development: a workload's chain under tile_hint variants against the generic kernels at a small size; reports where
outputs differ.  usage: python scripts/hint_check.py <workload> --log2 N hint hint ..."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("hints", nargs="+")
ap.add_argument("--log2", type=int, default=22)
a = ap.parse_args()
cfg = dict(bench.WORKLOADS[a.workload]); cfg["n"] = 1 << a.log2
dev = torch.device("cuda", 0)
src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
def run(**kw):
    p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"], **kw)
    o = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, o); torch.cuda.synchronize()
    return p, o.cpu().numpy()
pg, ref = run(kernel_policy=Q.KERNEL_GENERIC if hasattr(Q, "KERNEL_GENERIC") else 1)
print("generic: kind", pg.info.kernel_kind, "G", pg.info.tile_windows, "windows", ref.shape[0])
for h in a.hints:
    kw = {} if h == "-" else dict(tile_hint=[int(v) for v in h.split(":")])
    try:
        p, o = run(**kw)
    except Q.QuadrsError as e:
        print(h, "plan failed:", e); continue
    bad = np.nonzero((o.view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0]
    print(f"{h}: kind {p.info.kernel_kind} G {p.info.tile_windows} threads {p.info.threads} lds {p.info.lds_bytes}: {len(bad)} of {ref.shape[0]} windows differ")
    if len(bad):
        G = p.info.tile_windows
        print("   first:", bad[:12], " mod G:", sorted(set((bad % G).tolist()))[:20], " tiles:", sorted(set((bad // G).tolist()))[:10])
        w = bad[0]; k = np.nonzero(o[w].view(np.uint32) != ref[w].view(np.uint32))[0]
        print("   window", w, "bins", k[:10], "got", o[w][k[:4]], "ref", ref[w][k[:4]])
