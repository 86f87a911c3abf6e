#!/usr/bin/env python3
"""bench.py — Msamples/s through the fused shift -> FIR -> FFT chain on MI355X.

One "step" = one pass of the hot path over this rank's slab of a synthetic IQ stream that is
already resident in HBM.  Default workload = BASELINE.json configs[1] ("cfg2"):
1 GiB cf32 @21 Msps per GPU, shift 280000 -> lowpass -power 20 -decimate 16 2000000 ->
sparkfft -width 128.  With --gpus N the stream is N slabs (weak scaling), sharded by contiguous
window ranges with a (W-S)*D+T-sample halo fetched once from the next rank over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — algorithmic HBM bytes per launch / mean kernel time from HIP events recorded on
                 the launch stream (torch's current stream is handed to the C ABI);
  cpu_baseline — the CPU oracle in reference-literal mode (single thread, the reference CLI is
                 single-threaded) on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: fmt, samples per GPU, sample_rate, shift, (fc, decimate, taps), width, stride
    "cfg2": dict(fmt=0, n=1 << 27, sr=21_000_000, shift=280000, lp=(2_000_000, 16, 40), W=128, S=128,
                 desc="1 GiB cf32 @21Msps: shift 280000 -> lowpass -power 20 -decimate 16 2000000 -> sparkfft -width 128"),
    "cfg3p": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=280000, lp=(200_000, 32, 200), W=128, S=128,
                  desc="16 GiB cf32: shift -> 200-tap FIR decimate 32 -> 128-pt FFT (north_star target sentence)"),
    "cfg3": dict(fmt=1, n=1 << 33, sr=21_000_000, shift=280000, lp=(200_000, 32, 400), W=64, S=16,
                 desc="16 GiB cs8: unpack -> shift -> lowpass -power 200 -decimate 32 200000 -> sparkfft -width 64 -stride 16"),
    "cfg4": dict(fmt=0, n=1 << 32, sr=100_000_000, shift=None, lp=(5_000_000, 8, 512), W=1024, S=1024,
                 desc="gen 64 cosines @100 Msps, 32 GiB cf32: 512-tap FIR decimate 8 -> 1024-pt FFT"),
}
BPS = {0: 8, 1: 2, 2: 2, 3: 4}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E data-sheet peak (MI355X_MICROARCH.md); ~6300 measured copy


def synth_slab(torch, fmt, first, count, seed, device):
    """Deterministic synthetic IQ (tone at -shift + FSK-ish sign flips + noise + DC), generated on
    the device in chunks so nothing large crosses PCIe."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = 1 << 24
    if fmt == 0:
        out = torch.empty(count, 2, dtype=torch.float32, device=device)
    elif fmt == 3:
        out = torch.empty(count, 2, dtype=torch.int16, device=device)
    else:
        out = torch.empty(count, 2, dtype=torch.uint8 if fmt == 2 else torch.int8, device=device)
    for a in range(0, count, chunk):
        b = min(count, a + chunk)
        n = torch.arange(first + a, first + b, device=device, dtype=torch.float64)
        ph = (n * (-280000.0 / 21e6)) % 1.0 * (2 * np.pi)
        sign = torch.sign(torch.sin(n * (2 * np.pi * 9600.0 / 21e6)) + 1e-9)
        re = (0.02 * torch.cos(ph) * sign).float()
        im = (0.02 * torch.sin(ph) * sign).float()
        noise = torch.randn(b - a, 2, generator=g, device=device, dtype=torch.float32) * 0.002
        x = torch.stack([re + 0.005, im - 0.024], dim=1) + noise
        if fmt == 0:
            out[a:b] = x
        elif fmt == 1:
            out[a:b] = torch.clamp(torch.round(x * (127 * 20)), -128, 127).to(torch.int8)
        elif fmt == 2:
            out[a:b] = torch.clamp(torch.round(x * (127 * 20) + 127.5), 0, 255).to(torch.uint8)
        else:
            out[a:b] = torch.clamp(torch.round(x * (32767 * 20)), -32768, 32767).to(torch.int16)
    return out


def cpu_baseline(cfg, slab_bytes, target_s=12.0):
    """Times the CPU oracle (reference-literal cost class) on a bounded number of windows."""
    from oracle import oracle as O
    O.lib().qo_set_lowpass_closed_form(0)      # complex_convolve over every non-decimated position
    try:
        def build():
            ch = O.Chain.from_bytes(slab_bytes, cfg["fmt"], cfg["sr"])
            if cfg["shift"] is not None:
                ch = ch.shift(cfg["shift"])
            if cfg["lp"] is not None:
                ch = ch.lowpass(*cfg["lp"])
            return ch
        ch = build()
        total = O.lib().qo_spark_window_count(ch.len(), cfg["W"], cfg["S"])
        probe = min(total, 16)
        t0 = time.perf_counter()
        ch.spark_fft(cfg["W"], cfg["S"], max_windows=probe, want_codes=False)
        dt = time.perf_counter() - t0
        n = int(min(total, max(probe, target_s / max(dt / probe, 1e-9))))
        t0 = time.perf_counter()
        ch.spark_fft(cfg["W"], cfg["S"], max_windows=n, want_codes=False)
        dt = time.perf_counter() - t0
    finally:
        O.lib().qo_set_lowpass_closed_form(1)
    D = cfg["lp"][1] if cfg["lp"] else 1
    samples = n * cfg["S"] * D
    return dict(value=samples / dt / 1e6, unit="Msamples/s", cores=1, kind="port",
                sample=f"{n} consecutive windows ({samples} input samples, {dt:.1f} s) of the same workload, "
                       "oracle in reference-literal mode (per-window fetch, f64 sin/cos per sample, "
                       "complex_convolve over every non-decimated position), 1 thread")


def cpu_allcores(cfg, slab_bytes, target_s=8.0):
    """The same oracle in its cheapest exact form (FIR only at the decimated positions, same rounding order) on every
    host core the process may use: windows are independent, so threads take window ranges (ctypes releases the GIL).
    A second, friendlier CPU figure next to `cpu_baseline` (which keeps the reference's own cost class and thread count)."""
    import concurrent.futures as cf
    from oracle import oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    ch = O.Chain.from_bytes(slab_bytes, cfg["fmt"], cfg["sr"])
    if cfg["shift"] is not None:
        ch = ch.shift(cfg["shift"])
    if cfg["lp"] is not None:
        ch = ch.lowpass(*cfg["lp"])
    total = O.lib().qo_spark_window_count(ch.len(), cfg["W"], cfg["S"])
    probe = min(total, 64)
    t0 = time.perf_counter()
    ch.spark_fft(cfg["W"], cfg["S"], max_windows=probe, want_codes=False)
    per_win = max((time.perf_counter() - t0) / probe, 1e-9)
    per_thread = int(max(1, min(total // cores, target_s / per_win)))
    def work(k):
        ch.spark_fft(cfg["W"], cfg["S"], first_window=k * per_thread, max_windows=per_thread, want_codes=False)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    D = cfg["lp"][1] if cfg["lp"] else 1
    samples = per_thread * cores * cfg["S"] * D
    return dict(value=samples / dt / 1e6, unit="Msamples/s", cores=cores, kind="port",
                sample=f"{per_thread * cores} windows ({samples} input samples, {dt:.1f} s), oracle with the FIR evaluated at the decimated "
                       f"positions only (same products, same order), {cores} threads over window ranges")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of untimed launches before the warmup steps (clock ramp)")
    ap.add_argument("--samples-log2", type=int, default=None, help="override samples per GPU (2^k); rehearsals only")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on cuda:0, halo staged through the host")
    args = ap.parse_args()

    import torch
    import quadrs_amd as Q
    from quadrs_amd import shard as SH

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev_index = 0 if args.rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI

    cfg = dict(WORKLOADS[args.workload])
    if args.samples_log2 is not None:
        cfg["n"] = 1 << args.samples_log2
    fmt, bps = cfg["fmt"], BPS[cfg["fmt"]]
    n_total = cfg["n"] * world                       # weak scaling: one slab per GPU
    plan = Q.Plan(fmt, cfg["sr"], n_total, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    info = plan.info
    shards = SH.partition(plan.n_windows, world, info.raw_step, info.raw_per_window, info.tile_windows)
    me = shards[rank]

    if args.workload == "cfg4":
        # configs[3]: `gen` with 64 cosines @100 Msps (f_k = (k-32)*1 562 500 + 390 625 Hz), generated on
        # the device by the engine's own Gen kernel, outside the timed region
        own = torch.empty(me.own_count, 2, dtype=torch.float32, device=device)
        tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
        piece = 1 << 28
        for a in range(0, me.own_count, piece):
            Q.gen_device(tones, cfg["sr"], me.own_first + a, own[a:a + piece])
    else:
        own = synth_slab(torch, fmt, me.own_first, me.own_count, 0x5EED0002 + rank, device)
    own_u8 = own.view(torch.uint8).reshape(-1)
    if world == 1:
        slab = own_u8
    elif args.rehearse:
        slab = SH.exchange(own_u8.cpu(), shards, rank, bps, dist).to(device)         # gloo: staged through the host
    else:
        slab = SH.exchange(own_u8, shards, rank, bps, dist)                          # halo over RCCL/xGMI, once
    del own
    nw = me.w1 - me.w0
    out = torch.empty(nw, cfg["W"], dtype=torch.float32, device=device)

    def step():
        plan.run_device(slab, out, me.w0, nw, src_first=me.need_first, src_count=me.need_count)

    # settle: the first launches of a process run at ramping clocks (a cfg2 step is ~0.27 ms, so W warmup steps
    # alone can end before the GPU leaves its idle state); untimed, before the W warmup steps of the contract
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream bracket every 4th launch of the timed region (the kernel's own duration, the number
    # rocprofv3's kernel trace reports); the steps in between run back to back.  A marker between two kernels costs
    # ~2.5 us of GPU-side gap (scripts/launch_gap.py), so sampling keeps `value` within 0.3 % of an unmarked loop.
    ev = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % 4 == 0:
            ev[i] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[i][0].record()       # same stream the ABI launches on (torch's current stream)
            step()
            ev[i][1].record()
        else:
            step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev.values()]))

    finite = bool(torch.isfinite(out).all().item())
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # input samples every rank pushed through the chain per step (window starts advance S*D)
        samples_total = sum((s.w1 - s.w0) * info.raw_step for s in shards)
        alg_bytes = me.need_count * bps + nw * cfg["W"] * 4        # per launch on this rank: read once + norms written
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("workloads", {}).get(args.workload)        # per-workload PMC result of the latest profile round
                if ent and args.samples_log2 is None and world == 1:
                    traffic = ent.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s through shift->FIR->FFT chain",
            "value": samples_total / (elapsed / args.steps) / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['desc']}", "samples_per_gpu": cfg["n"], "taps": cfg["lp"][2],
                       "decimate": cfg["lp"][1], "width": cfg["W"], "stride": cfg["S"],
                       "parallelism": f"window-range shards x{world}, halo {(cfg['W'] - cfg['S']) * cfg['lp'][1] + cfg['lp'][2]} samples",
                       "outputs_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "qd::k_chain", "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            nwin_cpu = min(nw, 1 << 18)
            first, count = plan.src_range(me.w0 + (nw - nwin_cpu) // 2, nwin_cpu)
            host = slab[(first - me.need_first) * bps:(first - me.need_first + count) * bps].cpu().numpy().tobytes()
            line["cpu_baseline"] = cpu_baseline(cfg, host, args.cpu_seconds)
            line["cpu_allcores"] = cpu_allcores(cfg, host)        # informational: cheapest exact CPU form on every host core
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
