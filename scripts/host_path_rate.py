"""PCIe-inclusive rate of the host-resident path (qd_plan_run with QD_MEM_HOST buffers) on cfg2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import quadrs_amd as Q
N = 1 << 27
rng = np.random.default_rng(1)
x = (rng.standard_normal((N, 2), dtype=np.float32) * 0.02)
p = Q.Plan(Q.FMT_CF32, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128)
buf = x.view(np.uint8).reshape(-1)
p.run_host(buf, 0, 1024)
t0 = time.perf_counter(); out = p.run_host(buf); dt = time.perf_counter() - t0
print(f"host-resident cfg2: {dt*1e3:.1f} ms  {N/dt/1e6:.0f} Msamples/s  {N*8/dt/1e9:.2f} GB/s (PCIe + pageable memcpy inclusive)")
