"""Generic vs plan-time-specialised (hiprtc) kernel on a shape that has no built-in FixedGeo build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
N = 1 << 27
torch.manual_seed(3)
src = torch.randn(N, 2, device="cuda") * 0.02
for jit in ("0", "1"):
    os.environ["QD_JIT"] = jit
    t0 = time.perf_counter()
    p = Q.Plan(0, 21_000_000, N, shift_hz=-1_250_000, lowpass=(1_500_000, 12, 48), width=256, stride=256)
    t_plan = time.perf_counter() - t0
    out = torch.empty(p.n_windows, 256, device="cuda")
    p.run_device(src, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): p.run_device(src, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"QD_JIT={jit}: plan {t_plan*1e3:.0f} ms, {ms:.3f} ms/pass, {N*8/ms/1e6:.0f} GB/s, checksum {float(out.double().sum()):.6f}")
    if jit == "0": ref = out.clone()
    else: print("bit-identical to generic:", torch.equal(ref, out))
