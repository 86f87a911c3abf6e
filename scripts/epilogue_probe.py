"""Development: cfg2 step time by epilogue (f32 norms 32 MiB out vs glyph codes 8 MiB out vs bucket digits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
N = 1 << 27
src = torch.randn(N, 2, device="cuda") * 0.02
for name, epi, dt, cols in (("norms f32", Q.EPI_NORMS_F32, torch.float32, 128), ("glyph u8", Q.EPI_GLYPH_U8, torch.uint8, 128), ("norms f32", Q.EPI_NORMS_F32, torch.float32, 128), ("glyph u8", Q.EPI_GLYPH_U8, torch.uint8, 128)):
    p = Q.Plan(0, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128, epilogue=epi)
    out = torch.empty(p.n_windows, cols, device="cuda", dtype=dt)
    for _ in range(1500): p.run_device(src, out)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
    e[0].record()
    for i in range(100):
        p.run_device(src, out); e[i + 1].record()
    torch.cuda.synchronize()
    print(f"{name}: {e[0].elapsed_time(e[100]) / 100:.4f} ms", flush=True)
