"""Development: race hunt by repetition — the built-in kernels of the BASELINE shapes (deferred FFT, tile queue, LDS arrival
counter) run many times over one input; every output must equal the first run's, byte for byte.
usage: python scripts/determinism_soak.py [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
HINTS = {"pipe": [1, 256, 1, 8, 6, 2, 2 | (1796 << 8), 0], "pipe3": [12, 512, 1, 8, 4, 2, 1 | (32800 << 8), 0], "pipe3s": [12, 512, 1, 8, 4, 2, 1 | (164128 << 8), 0]}      # the role-split experiments
for name, log2, epi, hint in (("cfg3p", 27, 0, None), ("cfg3p", 27, 2, None), ("cfg4", 28, 0, None), ("cfg4", 28, 1, None), ("cfg3", 27, 0, None), ("cfg2", 26, 0, None),
                              ("cfg3p", 27, 0, "pipe"), ("cfg3", 27, 0, "pipe3"), ("cfg3", 27, 2, "pipe3"), ("cfg5", 27, 0, None), ("cfg5", 27, 1, None), ("cfg5", 27, 2, "pipe3s")):
    cfg = dict(bench.WORKLOADS[name]); cfg["n"] = 1 << log2
    if name == "cfg4":
        src = torch.empty(cfg["n"], 2, dtype=torch.float32, device=dev)
        Q.gen_device([(k - 32) * 1_562_500 + 390_625 for k in range(64)], cfg["sr"], 0, src)
    else:
        src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
    kw = dict(rng=(0.001, 0.5)) if epi == 1 else {}
    if hint:
        kw["tile_hint"] = HINTS[hint]
    p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"], epilogue=epi, **kw)
    shape = (p.n_windows,) if epi == 2 else (p.n_windows, cfg["W"])
    dt = torch.float32 if epi == 0 else torch.uint8
    ref = torch.empty(shape, dtype=dt, device=dev)
    out = torch.empty(shape, dtype=dt, device=dev)
    p.run_device(src, ref)
    torch.cuda.synchronize()
    bad = 0
    for r in range(runs):
        out.zero_()
        p.run_device(src, out)
        if not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
            bad += 1
    print(f"{name} epilogue {epi}{' (' + hint + ')' if hint else ''}: kind {p.info.kernel_kind}, flags {p.info.kernel_flags}, {p.n_windows} windows, {runs} runs, differing runs: {bad}", flush=True)
    del src, ref, out

# chains without a lowpass: the wave-local family (plan-time builds; LDS traffic ordered by wave-level fences only)
for fmt, shift, W, S, epi in ((0, None, 128, 128, 0), (0, 280000, 256, 256, 0), (0, None, 1024, 1024, 2), (1, 280000, 256, 256, 1), (0, 280000, 64, 64, 0), (3, None, 16, 16, 0),
                              (0, None, 4, 2, 0), (1, None, 8, 4, 1), (0, None, 64, 16, 0), (0, 280000, 64, 16, 0), (0, None, 512, 128, 0), (0, 280000, 1024, 256, 1)):
    n = 1 << 26
    src = bench.synth_slab(torch, fmt, 0, n, 0x5EED0002, dev)
    kw = dict(rng=(0.001, 0.5)) if epi == 1 else {}
    p = Q.Plan(fmt, 21_000_000, n, shift_hz=shift, width=W, stride=S, epilogue=epi, kernel_policy=Q.KERNEL_SPECIALISE, **kw)
    shape = (p.n_windows,) if epi == 2 else (p.n_windows, W)
    dt = torch.float32 if epi == 0 else torch.uint8
    ref = torch.empty(shape, dtype=dt, device=dev)
    out = torch.empty(shape, dtype=dt, device=dev)
    p.run_device(src, ref)
    torch.cuda.synchronize()
    bad = 0
    for r in range(runs):
        out.zero_()
        p.run_device(src, out)
        if not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
            bad += 1
    print(f"no lowpass fmt {fmt} shift {shift} W {W} S {S} epilogue {epi}: {p.kernel_name()[:48]}, flags {p.info.kernel_flags}, {p.n_windows} windows, {runs} runs, differing runs: {bad}", flush=True)
    p.close()
    del src, ref, out
