"""Development: the runtime-geometry (generic) kernels on the README chains — what every stream below 1 GiB without a cached plan-time build
runs on.  Times the README FSK chain (shift 280000 -> lowpass -power 200 -decimate 32 200000 -> sparkfft -width 64 -stride 16) on the
committed 1.5 MB recording and on a 256 MiB synthetic stream, the cfg2 chain on 256 MiB, cs8 / cs16 forms — generic kernels forced
(QD_KERNEL_GENERIC) — and prints ms per pass.  Run it once per library (QD_LIB_PATH) for a before / after.
usage: [QD_LIB_PATH=...] python scripts/generic_rate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import quadrs_amd as Q

dev = torch.device("cuda", 0)
print("library:", os.environ.get("QD_LIB_PATH", "quadrs_amd/libquadrs_hip.so"), flush=True)
rec = np.fromfile(os.path.join(ROOT, "tests", "golden", "fsk-example.sr21M.fc32"), dtype=np.uint8)
cases = [("README FSK chain, 1.5 MB recording (cf32)", 0, torch.from_numpy(rec).to(dev), rec.size // 8, 280000, (200_000, 32, 400), 64, 16)]
for fmt, n, shift, lp, W, S, what in ((0, 1 << 25, 280000, (200_000, 32, 400), 64, 16, "README FSK chain, 256 MiB cf32"),
                                      (0, 1 << 25, 280000, (2_000_000, 16, 40), 128, 128, "cfg2 chain, 256 MiB cf32"),
                                      (1, 1 << 27, 280000, (200_000, 32, 400), 64, 16, "README FSK chain, 256 MiB cs8"),
                                      (3, 1 << 26, 280000, (2_000_000, 16, 40), 128, 128, "cfg2 chain, 256 MiB cs16"),
                                      (0, 1 << 25, None, None, 128, 128, "sparkfft -width 128 alone, 256 MiB cf32")):
    cases.append((what, fmt, bench.synth_slab(torch, fmt, 0, n, bench.STREAM_SEED, dev), n, shift, lp, W, S))
for what, fmt, src, n, shift, lp, W, S in cases:
    p = Q.Plan(fmt, 21_000_000, n, shift_hz=shift, lowpass=lp, width=W, stride=S, kernel_policy=Q.KERNEL_GENERIC)
    out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
    for _ in range(3):
        p.run_device(src, out)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        p.run_device(src, out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{what:48s} kind {p.info.kernel_kind} G {p.info.tile_windows:3d} thr {p.info.threads}: {e0.elapsed_time(e1) / reps:8.4f} ms per pass, checksum {float(out.double().sum().item()):.9e}", flush=True)
    p.close()
