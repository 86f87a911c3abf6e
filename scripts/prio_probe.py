"""Development: wave-priority combinations per phase (phase 1 / FIR / FFT+epilogue), cfg2 and cfg3' shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
def run(N, lp, W, skip):
    os.environ["QD_DEBUG_SKIP"] = str(skip)
    p = Q.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=lp, width=W)
    out = torch.empty(p.n_windows, W, device="cuda")
    for _ in range(max(3, int(2e8 // N))): p.run_device(SRC[:N], out)
    torch.cuda.synchronize()
    K = max(4, int(4e9 // N) // 4)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    e[0].record()
    for i in range(K):
        p.run_device(SRC[:N], out); e[i + 1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[K]) / K
SRC = torch.randn(1 << 30, 2, device="cuda") * 0.02
for name, N, lp, W in (("cfg2", 1 << 27, (2_000_000, 16, 40), 128), ("cfg3p", 1 << 30, (200_000, 32, 200), 128)):
    for p1, fir, fft in ((0, 3, 3), (0, 0, 0), (0, 2, 3), (0, 1, 3), (1, 2, 3), (1, 3, 3), (0, 3, 2), (3, 0, 0), (0, 3, 3)):
        skip = 0x8000 | (p1 << 9) | (fir << 11) | (fft << 13)
        print(f"{name} prio phase1={p1} fir={fir} fft/epi={fft}: {run(N, lp, W, skip):.4f} ms", flush=True)
