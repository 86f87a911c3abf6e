"""Development: what the second-order NCO correction costs on the bench workloads (multi-GPU runs: every rank's plan describes the WHOLE
stream, n_total * |ratio| > 2^28 rad from two ranks of 2^31 samples on, so all ranks run the second-order kernel).
usage: python scripts/nco_order_cost.py [workload ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q

dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg3p", "cfg5"]):
    cfg = bench.WORKLOADS[name]
    src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
    res = {}
    for order in (1, 2):
        p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"], nco_order=order)
        out = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
        for _ in range(5):
            p.run_device(src, out)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                p.run_device(src, out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res[order] = (best, p.info.kernel_kind, p.info.kernel_flags)
        p.close(); del out
    print(f"{name}: first-order NCO {res[1][0]:.4f} ms (kind {res[1][1]}, flags {res[1][2]}), second-order {res[2][0]:.4f} ms (kind {res[2][1]}, flags {res[2][2]}): +{(res[2][0] / res[1][0] - 1) * 100:.1f} %", flush=True)
    del src
