"""Development: per-step device time of back-to-back qd_plan_run launches, plain stream vs one hipGraph of K steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
N = 1 << 27
src = torch.randn(N, 2, device="cuda") * 0.02
p = Q.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128)
out = torch.empty(p.n_windows, 128, device="cuda")
K = 50
for _ in range(1500): p.run_device(src, out)
torch.cuda.synchronize()
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / K * 1e3
def plain():
    for _ in range(K): p.run_device(src, out)
print(f"plain stream: {timed(plain):.4f} ms/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): p.run_device(src, out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): p.run_device(src, out)
torch.cuda.synchronize()
print(f"hipGraph of {K} steps: {timed(g.replay):.4f} ms/step")
print(f"plain stream again: {timed(plain):.4f} ms/step")
