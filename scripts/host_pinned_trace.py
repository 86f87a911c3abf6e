"""Development: the host-resident cfg2 run with pinned source and sink (bench.py's end_to_end.pinned), alone, so that
rocprofv3 --memory-copy-trace --kernel-trace shows how the chunk pipeline's copies and kernels actually overlap.
usage: python3 scripts/host_pinned_trace.py [chunk_MiB]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import quadrs_amd as Q
N = 1 << 27
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pin_in = Q.PinnedBuffer(N * 8)
rng = np.random.default_rng(1)
for a in range(0, N * 8, 1 << 26):
    pin_in.array[a:a + (1 << 26)] = rng.integers(0, 255, 1 << 26, dtype=np.uint8) & 0x3f
p = Q.Plan(Q.FMT_CF32, 21_000_000, N, shift_hz=280000, lowpass=(2_000_000, 16, 40), width=128, chunk_bytes=chunk << 20)
pin_out = Q.PinnedBuffer(p.n_windows * 128 * 4)
for rep in range(4):
    t0 = time.perf_counter()
    p.run_host(pin_in.array, pinned=True, out=pin_out.array)
    dt = time.perf_counter() - t0
    st = p.stats()
    print(f"rep {rep}: {dt*1e3:.2f} ms  {N*8/dt/1e9:.1f} GB/s  chunks {st.chunks} stage_ms {st.stage_ms:.2f}", flush=True)
