import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import quadrs_amd as Q
dev = torch.device("cuda", 0)
for fmt, W in ((0, 2), (0, 4), (0, 8), (1, 4), (1, 8), (3, 4)):
    n = (1 << 34) // bench.BPS[fmt]
    src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
    p = Q.Plan(fmt, 21_000_000, n, width=W, stride=W)
    out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
    for _ in range(2): p.run_device(src, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): p.run_device(src, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = n * bench.BPS[fmt] + out.numel() * 4
    print(f"fmt={fmt} W={W}: {p.kernel_name()[:70]}: {ms:.3f} ms, {byts / ms / 1e6 / 8000:.3f} of the HBM peak", flush=True)
    p.close(); del src, out
