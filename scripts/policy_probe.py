"""Development: time tiling candidates (QD_TUNE) for FIR-dominated chain shapes without a built-in kernel and
check each against the generic kernel's output bit for bit.  Backs the plan-time tiling policy in
quadrs_hip.hip (plan_init)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # QD_* tuning names -> qd_plan_options (quadrs_amd/engine.py)
import torch
import quadrs_amd as Q

N = 1 << 28
torch.manual_seed(5)
src = torch.randn(N, 2, device="cuda") * 0.02

def run(shape, env):
    W, S, D, T = shape
    for k in ("QD_TUNE", "QD_JIT", "QD_JIT_NOSLP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = Q.Plan(0, 21_000_000, N, shift_hz=-1_250_000, lowpass=(1_500_000, D, T), width=W, stride=S)
    out = torch.empty(p.n_windows, W, device="cuda")
    p.run_device(src, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): p.run_device(src, out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4, out, p.info

shapes = [(64, 16, 16, 400), (128, 32, 8, 256), (256, 256, 32, 512), (32, 8, 64, 800), (512, 512, 16, 256), (128, 128, 64, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for shape in shapes:
    W, S, D, T = shape
    ms0, ref, _ = run(shape, {"QD_JIT": "0"})
    print(f"shape W={W} S={S} D={D} T={T}: generic {ms0:.2f} ms")
    ms1, out, info = run(shape, {"QD_JIT": "1"})
    print(f"   plan-time default: G={info.tile_windows} nt={info.threads} lds={info.lds_bytes}  {ms1:.2f} ms  identical={torch.equal(ref, out)}")
    # candidate: the largest tile with <= 512 FIR outputs that fits LDS, 512 threads, 256-VGPR budget
    outs = lambda g: (g - 1) * S + W if S < W else g * W
    for nt, lb in ((512, 2), (1024, 4), (256, 4)):
        g = 1
        while outs(g + 1) <= max(nt, W):
            g += 1
        while g >= 1:
            for noslp in ("", "1"):
                env = {"QD_TUNE": f"{g}:{nt}:1:8:{lb}"}
                if noslp: env["QD_JIT_NOSLP"] = "1"
                ms, out, info = run(shape, env)
                ok = info.tile_windows == g and info.threads == nt
                if not ok: break
                print(f"   G={g} nt={nt} lb={lb} noslp={bool(noslp)}: lds={info.lds_bytes} {ms:.2f} ms  identical={torch.equal(ref, out)}", flush=True)
            if ok: break
            g = g * 3 // 4 if g > 4 else g - 1
