"""Development: the three sinks of a lowpass-free chain on a device-resident 16 GiB cf32 stream (norms f32 / glyph u8 / bucket digit).
usage: python scripts/nofir_sinks.py [W] [shift]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shift = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "none" else None
dev = torch.device("cuda", 0)
n = 1 << 31
src = bench.synth_slab(torch, 0, 0, n, 0x5EED0002, dev)
for epi, name in ((Q.EPI_NORMS_F32, "norms"), (Q.EPI_GLYPH_U8, "glyph"), (Q.EPI_BUCKET2_U8, "bucket")):
    p = Q.Plan(0, 21_000_000, n, shift_hz=shift, width=W, stride=W, epilogue=epi, rng=(0.01, 0.5))
    shape = (p.n_windows,) if epi == Q.EPI_BUCKET2_U8 else (p.n_windows, W)
    out = torch.empty(shape, dtype=torch.float32 if epi == Q.EPI_NORMS_F32 else torch.uint8, device=dev)
    for _ in range(2):
        p.run_device(src, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        p.run_device(src, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = n * 8 + out.numel() * out.element_size()
    print(f"W={W} shift={shift} {name}: {p.kernel_name()[:60]}: {ms:.3f} ms, {byts / ms / 1e6:.0f} GB/s = {byts / ms / 1e6 / 8000:.3f} of the HBM peak", flush=True)
    p.close(); del out
