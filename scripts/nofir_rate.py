"""Development: chains WITHOUT a lowpass on a device-resident stream (from -> [shift] -> sparkfft): ms per pass, fraction of the HBM peak
(bytes read + norms written), kernel kind / flags.  usage: python scripts/nofir_rate.py [log2 samples, default 31]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 31
dev = torch.device("cuda", 0)
only_overlap = len(sys.argv) > 2 and sys.argv[2] == "overlap"
for fmt, shift, W, S in ((0, None, 128, 128), (0, 280000, 128, 128), (0, None, 1024, 1024), (0, 280000, 64, 16), (1, 280000, 256, 256), (0, None, 4, 2), (0, 280000, 16, 16), (0, 280000, 64, 64), (0, 280000, 256, 256), (3, 280000, 512, 512),
                        (0, None, 64, 64), (0, None, 16, 16), (0, 280000, 32, 32), (0, None, 8, 8), (1, None, 64, 64), (1, 280000, 16, 16), (3, None, 32, 32),
                        (0, None, 16, 4), (0, None, 8, 4), (0, None, 8, 2), (0, None, 16, 8), (1, None, 4, 2), (1, None, 16, 4), (3, None, 8, 4), (0, None, 64, 16), (0, None, 128, 64), (0, None, 1024, 256), (1, None, 64, 16), (3, None, 256, 128)):
    if only_overlap and (S == W or shift is not None):
        continue
    n = 1 << (log2 + (2 if fmt == 1 else 0) - (3 if S < W else 0))
    src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
    p = Q.Plan(fmt, 21_000_000, n, shift_hz=shift, width=W, stride=S)
    out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
    for _ in range(2):
        p.run_device(src, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p.run_device(src, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = n * bench.BPS[fmt] + out.numel() * 4
    print(f"fmt={fmt} shift={shift} W={W} S={S} n=2^{n.bit_length() - 1}: kind {p.info.kernel_kind} flags {p.info.kernel_flags} G {p.info.tile_windows} thr {p.info.threads}: {ms:.3f} ms, "
          f"{byts / ms / 1e6:.0f} GB/s read+written = {byts / ms / 1e6 / 8000:.3f} of the HBM peak", flush=True)
    p.close(); del src, out
