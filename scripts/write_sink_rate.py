"""Development: rate of the `write` sink (QD_EPI_CF32_BLOCKS: shift -> lowpass -> decimated cf32 in blocks of 0x1000, src/lib.rs:178-213)
on a device-resident stream: ms per pass, input GB/s, output GB/s, kernel kind / flags.
usage: python scripts/write_sink_rate.py [log2 samples, default 31]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 31
dev = torch.device("cuda", 0)
for fmt, D, T in ((0, 32, 200), (0, 16, 40), (0, 32, 400), (1, 32, 400), (0, 8, 512)):
    n = 1 << (log2 + (2 if fmt == 1 else 0))
    src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
    p = Q.Plan(fmt, 21_000_000, n, shift_hz=280000, lowpass=(200_000, D, T), width=4096, epilogue=Q.EPI_CF32_BLOCKS)
    out = torch.empty(p.n_windows * 4096, 2, dtype=torch.float32, device=dev)
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
    in_b = n * bench.BPS[fmt]
    print(f"fmt={fmt} D={D} T={T}: {p.n_windows} blocks, kind {p.info.kernel_kind} flags {p.info.kernel_flags} G {p.info.tile_windows} thr {p.info.threads}: {ms:.3f} ms, "
          f"in {in_b / ms / 1e6:.0f} GB/s, out {out.numel() * 4 / ms / 1e6:.0f} GB/s, {n / ms / 1e3:.0f} Msamples/s", flush=True)
    p.close(); del src, out
