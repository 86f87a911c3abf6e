"""Development: what the per-step HIP events in bench.py's timed loop cost (GPU-side gaps between kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
N = 1 << 27
src = (torch.randn(N, 2, device="cuda") * 0.02)
p = Q.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128)
out = torch.empty(p.n_windows, 128, device="cuda")
K = 50
def run(mode):
    for _ in range(5): p.run_device(src, out)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K + 2)]
    t0 = time.perf_counter()
    if mode == "none":
        for i in range(K): p.run_device(src, out)
    elif mode == "pair":
        for i in range(K):
            evs[2 * i].record(); p.run_device(src, out); evs[2 * i + 1].record()
    elif mode == "chain":
        evs[0].record()
        for i in range(K):
            p.run_device(src, out); evs[i + 1].record()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K * 1e3
    if mode == "pair": k = sum(evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(K)) / K
    elif mode == "chain": k = evs[0].elapsed_time(evs[K]) / K
    else: k = float("nan")
    print(f"{mode:6s} wall/step {el:.4f} ms   event-measured kernel {k:.4f} ms")
for m in ("none", "pair", "chain", "none", "pair", "chain"):
    run(m)
