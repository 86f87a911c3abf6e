"""Development: un-hinted plans (QD_KERNEL_SPECIALISE) on overlapping-window, long-filter chains without a built-in kernel, against the
generic kernel of the same process: kernel kind / flags / tiling, ms per pass, outputs compared word for word.
usage: python scripts/overlap_sweep.py [log2 samples, default 29]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 29
dev = torch.device("cuda", 0)
n = 1 << log2
SHAPES = [(0, 400, 16, 64, 16), (0, 400, 32, 64, 16), (0, 256, 16, 128, 32), (0, 200, 8, 64, 32), (0, 512, 32, 128, 64), (1, 400, 16, 64, 16), (3, 256, 16, 128, 32)]
for fmt, T, D, W, S in SHAPES:
    src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
    kw = dict(shift_hz=280000, lowpass=(200_000, D, T), width=W, stride=S)
    res = {}
    for name, pol in (("auto", Q.KERNEL_SPECIALISE), ("generic", Q.KERNEL_GENERIC)):
        p = Q.Plan(fmt, 21_000_000, n, kernel_policy=pol, **kw)
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
        res[name] = (p.info, e0.elapsed_time(e1) / 5, out)
        p.close()
    same = bool(torch.equal(res["auto"][2].view(torch.int32), res["generic"][2].view(torch.int32)))
    a, g = res["auto"], res["generic"]
    print(f"fmt={fmt} T={T} D={D} W={W} S={S}: auto: kind {a[0].kernel_kind} flags {a[0].kernel_flags} G {a[0].tile_windows} thr {a[0].threads} {a[1]:.3f} ms  "
          f"generic: G {g[0].tile_windows} thr {g[0].threads} {g[1]:.3f} ms  identical={same}", flush=True)
    del src, res
