"""First-contact GPU check: parity on the two recordings + a cfg-2 shaped timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

def ulps(a, b):
    a = a.view(np.int32).astype(np.int64); b = b.view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a); b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)

# cfg 1
data = open(os.path.join(G, "cupboard-superdec.sr400.cf32"), "rb").read()
ch = O.Chain.from_bytes(data, O.FMT_CF32, 400)
n0, c0 = ch.spark_fft(4, 2, (0.001, 0.01))
p = Q.Plan(Q.FMT_CF32, 400, len(data) // 8, width=4, stride=2, epilogue=Q.EPI_NORMS_F32)
n1 = p.run_host(data)
print("cfg1 windows", p.n_windows, n0.shape, "norm max ulp", ulps(n0, n1).max())
p2 = Q.Plan(Q.FMT_CF32, 400, len(data) // 8, width=4, stride=2, epilogue=Q.EPI_GLYPH_U8, rng=(0.001, 0.01))
c1 = p2.run_host(data)
print("cfg1 glyph equal", np.array_equal(c0, c1))

# FSK chain
fsk = open(os.path.join(G, "fsk-example-head65536.sr21M.cf32"), "rb").read()
ch = O.Chain.from_bytes(fsk, O.FMT_CF32, 21000000).shift(280000).lowpass(200000, 32, 400)
n0, _ = ch.spark_fft(64, 16)
p = Q.Plan(Q.FMT_CF32, 21000000, len(fsk) // 8, shift_hz=280000, lowpass=(200000, 32, 400), width=64, stride=16)
n1 = p.run_host(fsk)
u = ulps(n0, n1)
print("fsk windows", p.n_windows, n0.shape, "max ulp", u.max(), "mismatch frac", (u > 0).mean(), "G", p.info.tile_windows, "lds", p.info.lds_bytes)
print("taps equal", np.array_equal(p.taps(), O.taps(200000, 21000000, 400)))

# fine-grained
x = np.frombuffer(fsk, dtype=np.float32).reshape(-1, 2)[:20000]
ratio = Q.shift_ratio(280000, 21000000)
for off in (0, 12345, 2**31 - 777, 2**33 + 5):
    a = O.shift_apply(x, off, ratio); b = Q.shift(x, off, ratio)
    u = ulps(a, b); print("shift off", off, "max ulp", u.max(), "mism", int((u > 0).sum()))
rng = np.random.default_rng(0)
b8 = rng.integers(0, 256, 2 * 70000, dtype=np.uint8).tobytes()
for f in (1, 2, 3):
    print("unpack", f, np.array_equal(O.unpack(f, b8).view(np.uint32), Q.unpack(f, b8).view(np.uint32)))
for W in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096):
    xx = rng.standard_normal((W * 3, 2)).astype(np.float32)
    ref = np.stack([O.norm(O.fft(xx[i * W:(i + 1) * W]))[np.r_[W // 2:W, 0:W // 2]] for i in range(3)])
    got = Q.fft_norm_batch(xx, W, 3, W)
    print("fft", W, "max ulp", ulps(ref, got).max())

# cfg-2 shaped timing, device resident
N = 1 << 27
torch.manual_seed(2)
src = torch.randn(N, 2, device="cuda", dtype=torch.float32) * 0.02
plan = Q.Plan(Q.FMT_CF32, 21000000, N, shift_hz=280000, lowpass=(2000000, 16, 40), width=128)
out = torch.empty(plan.n_windows, 128, device="cuda", dtype=torch.float32)
plan.run_device(src, out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): plan.run_device(src, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"cfg2: {ms:.3f} ms/pass  {N/ms/1e3:.1f} Msamples/s  {(N*8+plan.n_windows*512)/ms/1e6:.1f} GB/s  windows {plan.n_windows}")
# parity subsample on cfg 2: first 64 windows via oracle
nw = 64
a, b = plan.src_range(0, nw)
host = src[a:a + b].cpu().numpy().tobytes()
ch = O.Chain.from_bytes(host, O.FMT_CF32, 21000000).shift(280000).lowpass(2000000, 16, 40)
n0, _ = ch.spark_fft(128, 128, max_windows=nw - 1)
u = ulps(n0, out[:nw - 1].cpu().numpy()); print("cfg2 first windows max ulp", u.max(), "mism frac", (u > 0).mean())
