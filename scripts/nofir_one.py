"""Development: ONE chain without a lowpass on a device-resident stream, a few passes (for rocprofv3 counter runs).
usage: python scripts/nofir_one.py FMT SHIFT|none W [S] [log2 samples] [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

fmt = int(sys.argv[1]); shift = None if sys.argv[2] == "none" else float(sys.argv[2]); W = int(sys.argv[3])
S = int(sys.argv[4]) if len(sys.argv) > 4 else W
log2 = int(sys.argv[5]) if len(sys.argv) > 5 else 31
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda", 0)
n = 1 << log2
src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
p = Q.Plan(fmt, 21_000_000, n, shift_hz=shift, width=W, stride=S)
out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
p.run_device(src, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    p.run_device(src, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
byts = n * bench.BPS[fmt] + out.numel() * 4
print(f"fmt={fmt} shift={shift} W={W} S={S} n=2^{log2}: {p.kernel_name()}: {ms:.3f} ms, {byts / ms / 1e6 / 8000:.3f} of the HBM peak", flush=True)
