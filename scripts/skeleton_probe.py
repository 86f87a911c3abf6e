"""Development: where the chain kernel's skeleton loses bandwidth against scripts/ubench_stream.hip (6.3 TB/s).
Times cfg2-shaped plans with / without the shift stage, with all arithmetic ablated (QD_DEBUG_SKIP=15) or not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
N = 1 << 27
src = torch.randn(N, 2, device="cuda") * 0.02
def run(shift, skip, jit):
    os.environ["QD_DEBUG_SKIP"] = str(skip)
    os.environ["QD_JIT"] = jit
    p = Q.Plan(0, 21_000_000, N, shift_hz=shift, lowpass=(2_000_000, 16, 40), width=128)
    out = torch.empty(p.n_windows, 128, device="cuda")
    for _ in range(300): p.run_device(src, out)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    e[0].record()
    for i in range(40):
        p.run_device(src, out); e[i + 1].record()
    torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[40]) / 40
    print(f"shift={shift} skip={skip} jit={jit} kind={p.info.kernel_kind} G={p.info.tile_windows}: {ms:.4f} ms  {N*8.25/ms/1e6:.0f} GB/s", flush=True)
for skip in (0, 14, 14 + 16, 14 + 32, 14 + 16 + 32, 15 + 16 + 32, 16, 0):
    run(280000, skip, "1")
