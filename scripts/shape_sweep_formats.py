"""The sample formats at scale: cs8 (HackRF), cu8 (RTL-SDR), cs16 and cf32 streams of 16 GiB each through the cfg3
chain (shift -> 400 taps / 32 -> 64-point windows, stride 16: the streaming three-stage kernel) and the cfg2 chain (shift -> 40 taps / 16 ->
128-point windows): which kernel an un-hinted plan gets, ms per pass, fractions of the HBM peak and of the exact-order VALU roof, and a
bit-for-bit comparison of sampled window ranges with the generic kernel (src/lib.rs:241-255: the formats differ in the unpack only).
usage: python scripts/shape_sweep_formats.py [--gib 16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=int, default=16)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
SR, SHIFT = 21_000_000, 280000
NAMES = {0: "cf32", 1: "cs8", 2: "cu8", 3: "cs16"}
CHAINS = [("cfg3 chain", (200_000, 32, 400), 64, 16), ("cfg2 chain", (2_000_000, 16, 40), 128, 128)]
print(f"{'chain':>10} {'fmt':>5} {'samples':>8}  kind  flags   G threads   ms/pass  Msamples/s  hbm_frac valu_frac  vs generic", flush=True)
for fmt in (1, 2, 3, 0):
    n = (a.gib << 30) // bench.BPS[fmt]
    src = bench.synth_slab(torch, fmt, 0, n, bench.STREAM_SEED, dev)
    torch.cuda.synchronize()
    for label, lp, W, S in CHAINS:
        cfg = dict(fmt=fmt, n=n, sr=SR, shift=SHIFT, lp=lp, W=W, S=S)
        p = Q.Plan(fmt, SR, n, shift_hz=SHIFT, lowpass=lp, width=W, stride=S)
        out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
        for _ in range(2):
            p.run_device(src, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            p.run_device(src, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        alg = n * bench.BPS[fmt] + p.n_windows * W * 4
        nco = 2 if abs(p.info.ratio) * n > 268435456.0 else 1
        valu_roof, _, _ = bench.valu_roof_msamples(cfg, nco)
        msamples = p.n_windows * p.info.raw_step / (ms * 1e-3) / 1e6
        g = Q.Plan(fmt, SR, n, shift_hz=SHIFT, lowpass=lp, width=W, stride=S, kernel_policy=Q.KERNEL_GENERIC)
        same = True
        flat = src.view(torch.uint8).reshape(-1)
        for w0 in (0, p.n_windows // 3, p.n_windows - 4096):
            k = min(4096, p.n_windows - w0)
            ref = torch.empty(k, W, dtype=torch.float32, device=dev)
            first, count = g.src_range(w0, k)
            g.run_device(flat[first * bench.BPS[fmt]:(first + count) * bench.BPS[fmt]], ref, w0, k, src_first=first, src_count=count)
            torch.cuda.synchronize()
            same = same and bool(torch.equal(ref.view(torch.int32), out[w0:w0 + k].view(torch.int32)))
        print(f"{label:>10} {NAMES[fmt]:>5} 2^{n.bit_length() - 1:<6d} {p.info.kernel_kind:5d} {p.info.kernel_flags:6d} {p.info.tile_windows:3d} {p.info.threads:7d} {ms:9.3f} {msamples:11.0f} "
              f"{alg / (ms * 1e-3) / 8e12:9.3f} {msamples / valu_roof:9.3f}  {'identical' if same else 'DIFFERENT (NCO row length: see DESIGN section 4)':>10}", flush=True)
        p.close(); g.close()
        del out
    del src
    torch.cuda.empty_cache()
