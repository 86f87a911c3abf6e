"""Plan-time variant selection on shapes WITHOUT a built-in kernel: for each shape an un-hinted plan
(QD_KERNEL_SPECIALISE) over a 16 GiB cf32 stream — which kernel it got (kind, variant flags, tiling), ms per pass, fraction of
the 8 TB/s HBM peak and of the exact-order VALU roof (bench.valu_roof_msamples) — next to the nearest built-in shape, and a
bit-for-bit comparison of sampled window ranges with the generic kernel.
usage: python scripts/shape_sweep.py [--log2 31]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=31)
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
n = 1 << a.log2
dev = torch.device("cuda", 0)
src = bench.synth_slab(torch, 0, 0, n, bench.STREAM_SEED, dev)
torch.cuda.synchronize()
SR, SHIFT, FC = 21_000_000, 280000, 200_000
# (taps, decimate, width, has_shift, nearest built-in)
SHAPES = [(200, 32, 128, True, "cfg3p (built-in)"), (512, 8, 1024, False, "cfg4 (built-in)"),
          (160, 32, 128, True, "cfg3p"), (192, 32, 128, True, "cfg3p"), (256, 32, 128, True, "cfg3p"), (384, 32, 128, True, "cfg3p"),
          (192, 16, 128, True, "cfg3p"), (200, 64, 128, True, "cfg3p"), (192, 32, 64, True, "cfg3p"), (200, 32, 256, True, "cfg3p"),
          (256, 32, 128, True, "cfg3p, forced pair recipe 1:256:1:8:4:2:84994:0"), (384, 32, 128, True, "cfg3p, forced pair recipe 1:256:1:8:4:2:84994:0"),
          (256, 16, 256, True, "cfg3p"), (384, 8, 1024, False, "cfg4"), (256, 8, 512, False, "cfg4"), (512, 16, 512, False, "cfg4"), (160, 16, 512, True, "cfg4")]
print(f"{'taps':>5} {'D':>3} {'W':>5} shift  kind flags   G threads   ms/pass  hbm_frac valu_frac  vs generic   nearest", flush=True)
for T, D, W, sh, near in SHAPES:
    cfg = dict(fmt=0, n=n, sr=SR, shift=SHIFT if sh else None, lp=(FC, D, T), W=W, S=W)
    try:
        kw = dict(kernel_policy=Q.KERNEL_AUTO if "built-in" in near else Q.KERNEL_SPECIALISE)
        if "forced" in near:
            kw = dict(tile_hint=[int(v) for v in near.split()[-1].split(":")])
        p = Q.Plan(0, SR, n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=W, **kw)
    except Q.QuadrsError as e:
        print(f"{T:5d} {D:3d} {W:5d} plan failed: {e}", flush=True)
        continue
    out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
    for _ in range(3):
        p.run_device(src, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        p.run_device(src, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    alg = n * 8 + p.n_windows * W * 4
    nco = 0 if not sh else (2 if abs(p.info.ratio) * n > 268435456.0 else 1)
    valu_roof, _, _ = bench.valu_roof_msamples(cfg, nco)
    ms_samples = p.n_windows * p.info.raw_step / (ms * 1e-3) / 1e6
    # sampled window ranges against the generic kernel, bit for bit
    g = Q.Plan(0, SR, n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=W, kernel_policy=Q.KERNEL_GENERIC)
    same = True
    for w0 in (0, p.n_windows // 3, p.n_windows - 4096):
        k = min(4096, p.n_windows - w0)
        ref = torch.empty(k, W, dtype=torch.float32, device=dev)
        first, count = g.src_range(w0, k)
        g.run_device(src[first:first + count], ref, w0, k, src_first=first, src_count=count)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(ref.view(torch.int32), out[w0:w0 + k].view(torch.int32)))
    print(f"{T:5d} {D:3d} {W:5d} {'yes' if sh else 'no ':>5}  {p.info.kernel_kind:4d} {p.info.kernel_flags:5d} {p.info.tile_windows:3d} {p.info.threads:7d} {ms:9.3f} {alg / (ms * 1e-3) / 8e12:9.3f} "
          f"{ms_samples / valu_roof:9.3f}  {'identical' if same else 'DIFFERENT':>10}   {near}", flush=True)
    p.close(); g.close()
    del out
