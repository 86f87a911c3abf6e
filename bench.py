#!/usr/bin/env python3
"""bench.py — Msamples/s through the fused shift -> FIR -> FFT chain on MI355X.

One "step" = one pass of the hot path over this rank's slab of a synthetic IQ stream that is
already resident in HBM.  Default workload = the configuration BASELINE.json's north_star target is
quoted on ("cfg3p"): a 16 GiB cf32 stream per GPU, shift -> 200-tap FIR decimate 32 -> 128-pt FFT.
The other BASELINE configs (cfg2 / cfg3 / cfg4) are `--workload` choices and, in the default
one-GPU run, are also timed briefly and reported under "others".

With --gpus N the stream is N slabs (weak scaling), sharded by contiguous window ranges with a
(W-S)*D+T-sample halo fetched once from the next rank over RCCL.  `python bench.py --gpus N` with no
launcher starts its own N rank processes (before anything touches the GPU); under
`torch.distributed.run` it is one rank of the job.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — both roofs of the chain kernel: HBM (algorithmic bytes per launch / mean kernel time
                 from HIP events on the launch stream, peak 8 TB/s) and VALU (the chain's arithmetic at
                 the issue rates measured by scripts/ubench_valu.hip); "bound" names the lower one;
  cpu_baseline — the CPU oracle in reference-literal mode (single thread, the reference CLI is
                 single-threaded) on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("QUADRS_AMD_HARNESS_ENV", "1")      # harness: QD_TUNE / QD_JIT / QD_NO_FIXED ... are translated into qd_plan_options (quadrs_amd/engine.py)

WORKLOADS = {
    # name: fmt, samples per GPU, sample_rate, shift, (fc, decimate, taps), width, stride; {size} = the slab's size per GPU
    "cfg3p": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=280000, lp=(200_000, 32, 200), W=128, S=128,
                  desc="{size} cf32: shift 280000 -> 200-tap FIR (lowpass -power 100) decimate 32 -> sparkfft -width 128 (north_star target chain)"),
    "cfg2": dict(fmt=0, n=1 << 27, sr=21_000_000, shift=280000, lp=(2_000_000, 16, 40), W=128, S=128,
                 desc="{size} cf32 @21Msps: shift 280000 -> lowpass -power 20 -decimate 16 2000000 -> sparkfft -width 128"),
    "cfg3": dict(fmt=1, n=1 << 33, sr=21_000_000, shift=280000, lp=(200_000, 32, 400), W=64, S=16,
                 desc="{size} cs8: unpack -> shift -> lowpass -power 200 -decimate 32 200000 -> sparkfft -width 64 -stride 16"),
    "cfg4": dict(fmt=0, n=1 << 32, sr=100_000_000, shift=None, lp=(5_000_000, 8, 512), W=1024, S=1024,
                 desc="gen 64 cosines @100 Msps, {size} cf32: 512-tap FIR decimate 8 -> 1024-pt FFT"),
    # BASELINE.json configs[4]: the cfg3 chain on a cf32 stream sharded over the node, 16 GiB per GPU (128 GiB at 8 GPUs)
    "cfg5": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=280000, lp=(200_000, 32, 400), W=64, S=16,
                 desc="{size} cf32 per GPU: shift 280000 -> lowpass -power 200 -decimate 32 200000 -> sparkfft -width 64 -stride 16 (configs[4])"),
    # chains WITHOUT a lowpass (README example 1 / configs[0]'s chain at scale; the default run reports them under "no_lowpass"): here as
    # workloads so that the profile scripts can put rocprofv3 around them
    "nolp": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=None, lp=None, W=128, S=128, desc="{size} cf32: sparkfft -width 128 (no shift, no lowpass)"),
    "nolp_shift": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=280000, lp=None, W=128, S=128, desc="{size} cf32: shift 280000 -> sparkfft -width 128 (no lowpass)"),
    "nolp1024": dict(fmt=0, n=1 << 31, sr=21_000_000, shift=None, lp=None, W=1024, S=1024, desc="{size} cf32: sparkfft -width 1024 (no shift, no lowpass)"),
}


def workload_desc(cfg):
    size = cfg["n"] * BPS[cfg["fmt"]]
    txt = f"{size / (1 << 30):g} GiB" if size >= (1 << 30) else f"{size / (1 << 20):g} MiB"
    return cfg["desc"].format(size=txt)


DEFAULT_WORKLOAD = "cfg3p"
BPS = {0: 8, 1: 2, 2: 2, 3: 4}
FMT_NAMES = {0: "cf32", 1: "cs8", 2: "cu8", 3: "cs16"}
STREAM_SEED = 0x5EED0002     # ONE seed for the whole stream: the generator is keyed by absolute sample index, not by rank
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E data-sheet peak (MI355X_MICROARCH.md); ~6300 measured copy
HBM_ACHIEVABLE_GBPS = 6290.0  # what a float4 copy streams on this part (MI355X_MICROARCH.md, chip-level parameters)
SCLK_MAX_MHZ = 2400.0
# VALU issue rates measured on MI355X at four waves per SIMD, chip unthrottled at 2.4 GHz (scripts/ubench_pk.hip,
# profiles/r02/ubench_pk.log): ns per wave-instruction per SIMD.  A packed f32 op (two lane-operations) costs what an f64 op
# costs and 5 % less than two scalar f32 ops, so the roof prices the FIR and the complex multiply as packed ops.
T_F32_NS, T_F64_NS, T_PK_NS = 0.925, 1.76, 1.765
LANES_PER_CHIP = 64 * 4 * 256          # 64 lanes x 4 SIMDs x 256 CUs


def valu_ops_per_sample(cfg, nco_order):
    """Arithmetic of the exact-order chain per INPUT sample (SURVEY 8(d) formula, refined to instruction classes): scalar f32
    lane-operations, packed f32 operations (two lane-operations each) and f64-rate operations.  FIR: T/D taps per input sample,
    one packed multiply and one packed add per tap (re and im, separately rounded); NCO: 9 (first order)
    / 12 (second order) f64 ops + 2 f64->f32 converts, complex multiply 3 packed ops; FFT ~5 W log2 W flops per window; |X| per
    bin: the short exact form (see below); unpack per sample (two components):
    cs8 two converts (signed byte -> f32 in one SDWA instruction) + 2 packed ops (the two-operation exact division of qd_device.h on both
    components), cu8 / cs16 one more packed op (the bias subtraction).  Round 3: 10 / 12 scalar operations."""
    fc, D, T = cfg["lp"] if cfg["lp"] else (0, 1, 0)
    W, S = cfg["W"], cfg["S"]
    f32 = 5.0 * W * math.log2(W) / (S * D)
    pk = 2.0 * T / D
    # |X| per bin (round 4, qd_device.h norm_fast): 10 f64-rate slots (3 converts up, multiply + fma, convert down, 2 converts of the
    # seed, 2 fma of the Newton step, convert) and ~11 f32 slots (rsq = 4, two multiplies, five integer / compare operations for the
    # IEEE-path test); round 3 priced the IEEE f64 square root at 23 f64-rate slots, which made the roof lower and the fraction higher
    f64 = 10.0 * W / (S * D)
    f32 += 11.0 * W / (S * D)
    if cfg["shift"] is not None:
        pk += 3.0
        f64 += (12.0 if nco_order == 2 else 9.0) + 2.0
    f32 += {0: 0.0, 1: 2.0, 2: 2.0, 3: 2.0}[cfg["fmt"]]
    pk += {0: 0.0, 1: 2.0, 2: 3.0, 3: 3.0}[cfg["fmt"]]
    return f32, f64, pk


def valu_roof_msamples(cfg, nco_order):
    f32, f64, pk = valu_ops_per_sample(cfg, nco_order)
    ns_per_lane_sample = f32 * T_F32_NS + f64 * T_F64_NS + pk * T_PK_NS
    return LANES_PER_CHIP / (ns_per_lane_sample * 1e-9) / 1e6, f32 + 2.0 * pk, f64


def _splitmix64(torch, z):
    """splitmix64 finaliser on an int64 tensor (two's-complement wrap-around is what the generator wants)."""
    def lsr(v, k):                                   # logical shift right on int64
        return (v >> k) & ((1 << (64 - k)) - 1)
    z = (z ^ lsr(z, 30)) * -4658895280553007687      # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * -7723592293110705685      # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def synth_slab(torch, fmt, first, count, seed, device, out=None):
    """Deterministic synthetic IQ (tone at -shift + FSK-ish sign flips + noise + DC), generated on the device in chunks so
    nothing large crosses PCIe.  COUNTER-BASED: every value is a pure function of (seed, absolute sample
    index) — the noise is a splitmix64 hash of the index pushed through Box-Muller — so any rank of any world size
    regenerates exactly the bytes of the one-rank stream, halo included.  `out`: write into this (uint8 view of the) slab
    instead of allocating."""
    import numpy as np
    chunk = 1 << 24
    dt = {0: torch.float32, 3: torch.int16, 1: torch.int8, 2: torch.uint8}[fmt]
    if out is None:
        out = torch.empty(count, 2, dtype=dt, device=device)
    else:
        out = out.view(dt).reshape(-1, 2)
        assert out.shape[0] == count
    gold = -7046029254386353131                      # 0x9E3779B97F4A7C15
    for a in range(0, count, chunk):
        b = min(count, a + chunk)
        idx = torch.arange(first + a, first + b, device=device, dtype=torch.int64)
        n = idx.to(torch.float64)
        ph = (n * (-280000.0 / 21e6)) % 1.0 * (2 * np.pi)
        sign = torch.sign(torch.sin(n * (2 * np.pi * 9600.0 / 21e6)) + 1e-9)
        re = (0.02 * torch.cos(ph) * sign).float()
        im = (0.02 * torch.sin(ph) * sign).float()
        h = _splitmix64(torch, (idx + int(seed)) * gold)
        u1 = (((h >> 40) & 0xFFFFFF).to(torch.float64) + 0.5) * (1.0 / 16777216.0)        # (0, 1): 24 bits each
        u2 = (((h >> 8) & 0xFFFFFF).to(torch.float64) + 0.5) * (1.0 / 16777216.0)
        r = torch.sqrt(-2.0 * torch.log(u1)) * 0.002
        th = u2 * (2 * np.pi)
        x = torch.stack([re + 0.005 + (r * torch.cos(th)).float(), im - 0.024 + (r * torch.sin(th)).float()], dim=1)
        if fmt == 0:
            out[a:b] = x
        elif fmt == 1:
            out[a:b] = torch.clamp(torch.round(x * (127 * 20)), -128, 127).to(torch.int8)
        elif fmt == 2:
            out[a:b] = torch.clamp(torch.round(x * (127 * 20) + 127.5), 0, 255).to(torch.uint8)
        else:
            out[a:b] = torch.clamp(torch.round(x * (32767 * 20)), -32768, 32767).to(torch.int16)
    return out


class GpuSensors:
    """Board power / shader clock of ONE GPU, read from the amdgpu hwmon files in sysfs (power1_average or power1_input in
    microwatts, freq1_input in Hz, power1_cap in microwatts).  Plain file reads: the process that holds the GPU never forks or
    execs anything to learn its own power (rocm-smi is a Python script; starting it from a process that has initialised HIP is
    an exec-after-GPU-init, which this pool refuses — VERDICT r02 item 9)."""

    def __init__(self, pci_bdf=None):
        import glob
        self.power = self.freq = self.cap = None
        cands = []
        if pci_bdf:
            cands = glob.glob(f"/sys/bus/pci/devices/{pci_bdf}/hwmon/hwmon*")
        if not cands and pci_bdf is None:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        for d in cands:
            for f in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(d, f)):
                    self.power = os.path.join(d, f)
                    break
            if self.power:
                self.freq = os.path.join(d, "freq1_input") if os.path.exists(os.path.join(d, "freq1_input")) else None
                self.cap = os.path.join(d, "power1_cap") if os.path.exists(os.path.join(d, "power1_cap")) else None
                self.dir = d
                break

    @staticmethod
    def for_torch_device(torch, index=0):
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            bdf = None
        s = GpuSensors(bdf)
        s.bdf = bdf
        return s

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def ok(self):
        return self.power is not None and self._num(self.power) is not None

    def read(self):
        """(watts, sclk MHz or None)"""
        w = self._num(self.power) if self.power else None
        f = self._num(self.freq) if self.freq else None
        return (w * 1e-6 if w is not None else None, f * 1e-6 if f is not None else None)

    def cap_watts(self):
        c = self._num(self.cap) if self.cap else None
        return c * 1e-6 if c is not None else None


def sample_power(step, torch, seconds, sensors=None):
    """AFTER the timed region: keep the kernel running back to back for a few seconds and read the board's package power and
    shader clock meanwhile (informational: says whether the kernel runs at the power cap — DESIGN.md section 7).  The readings
    come from sysfs (GpuSensors), in this process, between batches of launches.  Returns None where the files are missing."""
    sensors = sensors or GpuSensors.for_torch_device(torch, torch.cuda.current_device())
    if not sensors.ok():
        return None
    samples = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(16):
            step()
        w, mhz = sensors.read()            # read while the queue is still full of launches, then drain
        torch.cuda.synchronize()
        if w is not None:
            samples.append((w, mhz))
    samples = samples[2:] if len(samples) > 4 else samples          # the first readings may predate the load
    if not samples:
        return None
    clocks = [s[1] for s in samples if s[1]]
    return {"watts": max(s[0] for s in samples), "watts_mean": sum(s[0] for s in samples) / len(samples),
            "sclk_mhz": min(clocks) if clocks else None, "sclk_mhz_mean": sum(clocks) / len(clocks) if clocks else None,
            "cap_watts": sensors.cap_watts(), "samples": len(samples), "card": getattr(sensors, "bdf", None),
            "note": "amdgpu hwmon (sysfs) while the kernel runs back to back, after the timed region: highest package power, lowest shader clock seen"}


def cpu_baseline(cfg, slab_bytes, target_s=12.0):
    """Times the CPU oracle (reference-literal cost class) on a bounded number of windows."""
    from oracle import oracle as O
    O.lib().qo_set_lowpass_closed_form(0)      # complex_convolve over every non-decimated position
    try:
        ch = O.Chain.from_bytes(slab_bytes, cfg["fmt"], cfg["sr"])
        if cfg["shift"] is not None:
            ch = ch.shift(cfg["shift"])
        if cfg["lp"] is not None:
            ch = ch.lowpass(*cfg["lp"])
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
                sample=f"{n} consecutive windows ({samples} input samples, {dt:.1f} s) from the middle of the same stream, "
                       "oracle in reference-literal mode (per-window fetch, f64 sin/cos per sample, "
                       "complex_convolve over every non-decimated position), 1 thread")


def cpu_allcores(cfg, slab_bytes, target_s=6.0):
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


# ------------------------------------------------------------------ self-launch (python bench.py --gpus N, no launcher)

def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Parent of a self-launched multi-rank run.  Nothing here imports torch or touches the GPU: the parent only starts
    N fresh rank processes (one per GPU; every rank on cuda:0 with --rehearse / on the CPU with --stub), relays rank 0's
    JSON line and fails if any rank fails.  Never re-execs itself."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in (out0 or "").splitlines():              # rank 0's JSON line; library chatter on its stdout (gloo) goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    return 0


# ------------------------------------------------------------------ one measurement

def measure(args, name, cfg, rank, world, device, dist, steps, warmup, with_cpu):
    import numpy as np
    import torch
    import quadrs_amd as Q
    from quadrs_amd import shard as SH

    fmt, bps = cfg["fmt"], BPS[cfg["fmt"]]
    n_total = cfg["n"] * world                       # weak scaling: one slab per GPU
    plan = Q.Plan(fmt, cfg["sr"], n_total, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    info = plan.info
    # one tiling for every rank's shard table: rank 0's (a rank whose plan obtained another kernel fails here, not in the exchange)
    tile_w = SH.agree_tile_windows(int(info.tile_windows), dist, torch, "cpu" if (dist is None or args.rehearse) else device)
    shards = SH.partition(plan.n_windows, world, info.raw_step, info.raw_per_window, tile_w)
    me = shards[rank]

    # ONE pre-sized buffer per rank: own samples first, room for the halo behind them; the generator writes into it and the
    # halo is received into its tail (no second slab, no concatenation)
    slab = SH.alloc_slab(me, bps, device, torch)
    own_b = me.own_count * bps

    def generate(first, count, dst_u8):
        """samples [first, first + count) of the stream into dst_u8 (device, uint8): pure function of the absolute index"""
        if count == 0:
            return
        if name == "cfg4":
            # configs[3]: `gen` with 64 cosines @100 Msps (f_k = (k-32)*1 562 500 + 390 625 Hz), generated on
            # the device by the engine's own Gen kernel, outside the timed region
            tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
            dst = dst_u8.view(torch.float32).reshape(-1, 2)
            piece = 1 << 28
            for a in range(0, count, piece):
                Q.gen_device(tones, cfg["sr"], first + a, dst[a:a + piece])
            torch.cuda.synchronize()
        else:
            synth_slab(torch, fmt, first, count, STREAM_SEED, device, out=dst_u8)

    generate(me.own_first, me.own_count, slab[:own_b])
    if world > 1:
        if args.rehearse:
            host = slab.cpu()
            SH.exchange(host, shards, rank, bps, dist)                                # gloo: staged through the host
            slab[own_b:] = host[own_b:].to(device)
            del host
        else:
            SH.exchange(slab, shards, rank, bps, dist)                                # halo over RCCL/xGMI, once, in place
        # the exchange step priced on its own (informational): the same exchange once more — it lands the same bytes — between
        # barriers; the first call above also built the communicator's point-to-point channels
        torch.cuda.synchronize()
        dist.barrier()
        t_x = time.perf_counter()
        if not args.rehearse:
            SH.exchange(slab, shards, rank, bps, dist)
            torch.cuda.synchronize()
        dist.barrier()
        exchange_ms = (time.perf_counter() - t_x) * 1e3
    nw = me.w1 - me.w0
    out = torch.empty(nw, cfg["W"], dtype=torch.float32, device=device)

    def step():
        plan.run_device(slab, out, me.w0, nw, src_first=me.need_first, src_count=me.need_count)

    # ---- seam verification (multi-rank runs; outside the timed region).  (1) The halo this rank RECEIVED is compared, byte for
    # byte, with the same samples regenerated locally (the generator is keyed by absolute index).  (2) The rank's first and last
    # four windows — the last ones read the halo — are recomputed by a FRESH plan from a small slab generated from scratch, and
    # compared with the rows of the sharded run.  Every rank's verdict is AND-reduced into "seams_verified".
    seams = None
    if world == 1:
        exchange_ms = None
    if world > 1:
        ok = True
        if me.halo:
            regen = torch.empty(me.halo * bps, dtype=torch.uint8, device=device)
            generate(me.own_first + me.own_count, me.halo, regen)
            ok = ok and bool(torch.equal(regen, slab[own_b:]))
            del regen
        step()
        torch.cuda.synchronize()
        if nw:
            chk = Q.Plan(fmt, cfg["sr"], n_total, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
            k = int(min(4, nw))
            for w_first in sorted({me.w0, me.w1 - k}):
                first, count = chk.src_range(w_first, k)
                small = torch.empty(count * bps, dtype=torch.uint8, device=device)
                generate(first, count, small)
                rows = torch.empty(k, cfg["W"], dtype=torch.float32, device=device)
                chk.run_device(small, rows, w_first, k, src_first=first, src_count=count)
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(rows.view(torch.int32), out[w_first - me.w0:w_first - me.w0 + k].view(torch.int32)))
                del small, rows
            chk.close()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if args.rehearse else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        seams = bool(flag.item())

    # settle: the first launches of a process run at ramping clocks; untimed, before the W warmup steps of the contract
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        step()
        torch.cuda.synchronize()
    for _ in range(warmup):
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
    for i in range(steps):
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
    power = None
    if rank == 0 and world == 1 and getattr(args, "power_sample", False):
        power = sample_power(step, torch, getattr(args, "power_seconds", 2.5))

    res = None
    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        # input samples every rank pushed through the chain per step (window starts advance S*D)
        samples_total = sum((s.w1 - s.w0) * info.raw_step for s in shards)
        samples_rank = nw * info.raw_step
        alg_bytes = me.need_count * bps + nw * cfg["W"] * 4        # per launch on this rank: read once + norms written
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        nco_order = 0 if cfg["shift"] is None else (2 if abs(info.ratio) * n_total > 268435456.0 else 1)
        valu_ms_roof, f32_ops, f64_ops = valu_roof_msamples(cfg, nco_order)
        kernel_msamples = samples_rank / (kernel_ms * 1e-3) / 1e6
        hbm_roof_msamples = HBM_PEAK_GBPS * 1e9 / (alg_bytes / samples_rank) / 1e6
        hbm_frac, valu_frac = achieved / HBM_PEAK_GBPS, kernel_msamples / valu_ms_roof
        bound = "hbm" if hbm_roof_msamples <= valu_ms_roof else "valu"
        # the figure rocprofv3 measured for this workload in the latest committed profile round (profiles/): NOT measured in
        # this run, hence its own key; `traffic` (in-run PMC) stays null
        traffic_profiled = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("workloads", {}).get(name)
                if ent and args.samples_log2 is None and world == 1:
                    traffic_profiled = {"hbm_bytes_per_launch": ent.get("hbm_bytes_per_launch"), "source": ent.get("source", "profiles/")}
            except Exception:
                traffic_profiled = None
        roof = {"bound": bound, "kernel": plan.kernel_name(), "kernel_ms": kernel_ms, "algorithmic_bytes": alg_bytes,
                "traffic": None, "traffic_profiled": traffic_profiled,
                # frac_of_achievable: against what a copy kernel streams on this part (MI355X_MICROARCH.md: 6.29 TB/s), beside — never
                # instead of — the data-sheet fraction
                "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac,
                        "achievable_peak": HBM_ACHIEVABLE_GBPS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS},
                "valu": {"achieved": kernel_msamples, "peak": valu_ms_roof, "unit": "Msamples/s", "frac": valu_frac,
                         "f32_ops_per_sample": f32_ops, "f64_ops_per_sample": f64_ops,
                         "issue_ns_per_wave_instr": {"f32": T_F32_NS, "f64": T_F64_NS, "packed_f32": T_PK_NS},
                         "note": "exact-order arithmetic at the issue rates scripts/ubench_pk.hip measured (4 waves / SIMD, 2.4 GHz); FIR and complex multiply priced as packed ops"}}
        if bound == "hbm":
            roof.update(achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s", frac=hbm_frac)
        else:
            # the VALU roof in TFLOP/s: every lane-operation of the count above is one flop (separately rounded mul / add; f32 and
            # f64 alike); `peak` is the same count at the roof's sample rate, i.e. what this instruction mix can reach at the
            # measured issue rates — not the data-sheet FMA peak, which unfused exact-order arithmetic cannot approach
            flops = f32_ops + f64_ops
            roof.update(achieved=kernel_msamples * 1e6 * flops / 1e12, peak=valu_ms_roof * 1e6 * flops / 1e12, unit="TFLOP/s", frac=valu_frac)
        if power:
            roof["power"] = power
            # the VALU roof at the shader clock the board actually held under this kernel (the issue rates above are per-instruction times at
            # an unthrottled 2.4 GHz; at the package power cap the governor runs the kernel well below that): peak_at_sclk = peak x sclk / 2400
            if power.get("sclk_mhz_mean"):
                scale = power["sclk_mhz_mean"] / SCLK_MAX_MHZ
                roof["valu"]["sclk_mhz"] = power["sclk_mhz_mean"]
                roof["valu"]["peak_at_sclk"] = valu_ms_roof * scale
                roof["valu"]["frac_at_sclk"] = kernel_msamples / (valu_ms_roof * scale)
        res = {"workload": name, "value": samples_total / (elapsed / steps) / 1e6, "ms_per_step": ms_per_step, "roofline": roof,
               "outputs_finite": finite, "seams_verified": seams,
               "halo_exchange": None if world == 1 or args.rehearse else {"bytes_per_rank": me.halo * bps, "ms": exchange_ms, "note": "one neighbour send/recv, once per resident slab, outside the timed steps"},
               "kernel_kind": int(info.kernel_kind), "kernel_flags": int(info.kernel_flags), "tile_windows": int(info.tile_windows), "threads": int(info.threads),
               "nco_order": int(nco_order)}
        if with_cpu:
            nwin_cpu = int(min(nw, max(64, (1 << 30) // (info.raw_step * bps))))     # at most 1 GiB of the stream goes to the host
            first, count = plan.src_range(me.w0 + (nw - nwin_cpu) // 2, nwin_cpu)
            host = slab[(first - me.need_first) * bps:(first - me.need_first + count) * bps].cpu().numpy().tobytes()
            res["cpu_baseline"] = cpu_baseline(cfg, host, args.cpu_seconds)
            res["cpu_allcores"] = cpu_allcores(cfg, host)        # informational: cheapest exact CPU form on every host core
    plan.close()
    del slab, out
    torch.cuda.empty_cache()
    return res


def fast_mode_leg(cfg, device, steps=8):
    """QD_MODE_FAST on the default workload (SURVEY 8(b) mode{EXACT_ORDER / FAST}; informational, NEVER `value`): the same chain
    with the FIR's multiply-adds fused (one rounding per tap).  Reported with its deviation from the exact run in ulp of the
    window maximum over sampled window ranges."""
    import torch
    import quadrs_amd as Q
    n = cfg["n"]
    slab = synth_slab(torch, cfg["fmt"], 0, n, STREAM_SEED, device)
    kw = dict(shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    fast = Q.Plan(cfg["fmt"], cfg["sr"], n, mode=Q.MODE_FAST, **kw)
    exact = Q.Plan(cfg["fmt"], cfg["sr"], n, **kw)
    a = torch.empty(fast.n_windows, cfg["W"], dtype=torch.float32, device=device)
    for _ in range(3):
        fast.run_device(slab, a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fast.run_device(slab, a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    dev_max = 0.0
    k = min(16384, fast.n_windows)
    for w0 in (0, (fast.n_windows - k) // 2, fast.n_windows - k):
        b = torch.empty(k, cfg["W"], dtype=torch.float32, device=device)
        first, count = exact.src_range(w0, k)
        exact.run_device(slab.view(torch.uint8).reshape(-1)[first * BPS[cfg["fmt"]]:(first + count) * BPS[cfg["fmt"]]], b, w0, k, src_first=first, src_count=count)
        torch.cuda.synchronize()
        aa, bb = a[w0:w0 + k].double(), b.double()
        mx = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        ulp = torch.pow(2.0, torch.floor(torch.log2(mx.double())) - 23)
        dev_max = max(dev_max, float(((aa - bb).abs() / ulp).max().item()))
    alg = n * BPS[cfg["fmt"]] + fast.n_windows * cfg["W"] * 4
    res = {"mode": "QD_MODE_FAST: FIR multiply-adds fused (v_pk_fma_f32), everything else as in the exact mode; not the reference's bits",
           "kernel_flags": int(fast.info.kernel_flags), "fused": bool(fast.info.kernel_flags & 16384), "ms_per_step": ms, "steps": steps,
           "value": fast.n_windows * fast.info.raw_step / (ms * 1e-3) / 1e6, "unit": "Msamples/s", "hbm_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "max_deviation_from_exact_ulp_of_window_max": dev_max}
    fast.close(); exact.close()
    del slab, a
    torch.cuda.empty_cache()
    return res


def write_sink_leg(cfg, device, steps=8):
    """The `write` sink on the default workload's chain (shift -> lowpass -> decimated cf32 in read_at blocks of 0x1000,
    src/lib.rs:178-213; informational, never `value`): the streaming kernel without an FFT stage, 8 N / D bytes written."""
    import torch
    import quadrs_amd as Q
    n = cfg["n"]
    slab = synth_slab(torch, cfg["fmt"], 0, n, STREAM_SEED, device)
    p = Q.Plan(cfg["fmt"], cfg["sr"], n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=4096, epilogue=Q.EPI_CF32_BLOCKS)
    out = torch.empty(p.n_windows * 4096, 2, dtype=torch.float32, device=device)
    for _ in range(3):
        p.run_device(slab, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        p.run_device(slab, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    in_b, out_b = n * BPS[cfg["fmt"]], out.numel() * 4
    res = {"chain": f"shift {cfg['shift']} -> lowpass -power {cfg['lp'][2] // 2} -decimate {cfg['lp'][1]} {cfg['lp'][0]} -> write (blocks of 4096)",
           "kernel_kind": int(p.info.kernel_kind), "kernel_flags": int(p.info.kernel_flags), "threads": int(p.info.threads), "ms_per_step": ms, "steps": steps,
           "value": n / (ms * 1e-3) / 1e6, "unit": "Msamples/s", "read_GBps": in_b / (ms * 1e-3) / 1e9, "written_GBps": out_b / (ms * 1e-3) / 1e9,
           "hbm_frac": (in_b + out_b) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "outputs_finite": bool(torch.isfinite(out).all().item())}
    p.close()
    del slab, out
    torch.cuda.empty_cache()
    return res


def no_lowpass_leg(device, steps=6):
    """Chains WITHOUT a lowpass (`from F [shift] sparkfft`, README example 1 / BASELINE configs[0]'s chain at scale; informational, never
    `value`) on the wave-local kernels (k_spark / k_spark2).  Every input sample is an FFT input and every window position yields W f32
    norms, so the algorithmic bytes are the stream read once + 4 W per window written; fractions are of the 8 TB/s data-sheet peak and of
    the 6.29 TB/s a copy kernel streams."""
    import torch
    import quadrs_amd as Q
    res = {"stream": "16 GiB @21 Msps (cf32: 2^31 samples; cs8: 2^33), stride == width unless the label says otherwise", "unit": "ms per pass", "shapes": {}}
    slab, slab_fmt = None, None
    for label, fmt, shift, W, S in (("w128", 0, None, 128, 128), ("w128_glyph", 0, None, 128, 128), ("shift_w128", 0, 280000, 128, 128), ("w1024", 0, None, 1024, 1024),
                                    ("w64", 0, None, 64, 64), ("w128_stride64_4GiB", 0, None, 128, 64), ("cs8_w256", 1, None, 256, 256)):
        glyph = label.endswith("_glyph")                                     # the reference's own sink: one u8 cell per bin (src/fft.rs:54-60)
        n = (1 << 34) // BPS[fmt] if S == W else (1 << 32) // BPS[fmt]       # (overlapping windows: a quarter of the stream, the output is W / S times the input)
        if slab is None or slab_fmt != fmt or slab.numel() * slab.element_size() != n * BPS[fmt]:
            del slab
            torch.cuda.empty_cache()
            slab, slab_fmt = synth_slab(torch, fmt, 0, n, STREAM_SEED, device), fmt
        p = Q.Plan(fmt, 21_000_000, n, shift_hz=shift, width=W, stride=S, **(dict(epilogue=Q.EPI_GLYPH_U8, rng=(0.01, 0.5)) if glyph else {}))
        out = torch.empty(p.n_windows, W, dtype=torch.uint8 if glyph else torch.float32, device=device)
        for _ in range(2):
            p.run_device(slab, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            p.run_device(slab, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        byts = n * BPS[fmt] + out.numel() * out.element_size()
        res["shapes"][label] = {"chain": f"from {FMT_NAMES[fmt]} {'shift 280000 -> ' if shift else ''}sparkfft -width {W}{'' if S == W else f' -stride {S}'}{' (glyph cells)' if glyph else ''}", "samples": n,
                                "ms": ms, "Msamples_per_s": n / (ms * 1e-3) / 1e6,
                                "GBps_read_plus_written": byts / (ms * 1e-3) / 1e9, "hbm_frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "frac_of_achievable": byts / (ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS, "kernel": p.kernel_name(),
                                "kernel_kind": int(p.info.kernel_kind), "kernel_flags": int(p.info.kernel_flags), "outputs_finite": True if glyph else bool(torch.isfinite(out).all().item())}
        p.close()
        del out
    del slab
    torch.cuda.empty_cache()
    return res


def end_to_end_host(device):
    """Host-resident cfg2 (1 GiB cf32 in host memory -> norms in host memory) through qd_plan_run's chunked, double-buffered
    path: PCIe-inclusive, reported beside the kernel figures, never as `value`.  Pinned buffers (QD_MEM_HOST_PINNED) are the
    DMA source / target themselves; the pageable figure stages through a pinned ring with a multi-threaded memcpy."""
    import numpy as np
    import torch
    import quadrs_amd as Q
    cfg = WORKLOADS["cfg2"]
    n = cfg["n"]
    slab = synth_slab(torch, 0, 0, n, 0x5EED0002, device)
    pin_in = Q.PinnedBuffer(n * 8)
    torch.cuda.synchronize()
    pin_in.array[:] = slab.view(torch.uint8).reshape(-1).cpu().numpy()
    del slab
    plan = Q.Plan(0, cfg["sr"], n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    out_bytes = plan.n_windows * cfg["W"] * 4
    pin_out = Q.PinnedBuffer(out_bytes)
    res = {"workload": "cfg2, host-resident: 1 GiB cf32 in host memory -> fused chain -> norms in host memory", "unit": "GB/s (input bytes / wall)"}

    def best_of(p, src, pinned, reps=5):      # the host is shared with seven other GPUs' tenants: repeats of one box range 19.6 - 24 ms
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            if pinned:
                p.run_host(src, pinned=True, out=pin_out.array)
            else:
                p.run_host(src)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best
    for name, pinned in (("pinned", True), ("pageable", False)):
        src = pin_in.array if pinned else np.array(pin_in.array, copy=True)
        best = best_of(plan, src, pinned)
        res[name] = {"GBps": n * 8 / best / 1e9, "ms": best * 1e3, "Msamples_per_s": n / best / 1e6, "chunk_MiB": 64}
    sweep = {}
    for mib in (8, 16, 32, 128):                 # the library's default chunk is 64 MiB: how the pinned path moves with it
        p2 = Q.Plan(0, cfg["sr"], n, shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"], chunk_bytes=mib << 20)
        sweep[str(mib)] = n * 8 / best_of(p2, pin_in.array, True, reps=2) / 1e9
        p2.close()
    res["pinned_GBps_by_chunk_MiB"] = sweep
    st = plan.stats()
    res["chunks"] = int(st.chunks)
    plan.close(); pin_in.close(); pin_out.close()
    return res


def stub_rank(args, rank, world):
    """--stub: the launch / rendezvous / barrier / max-over-ranks skeleton with a sleep for a step, on the CPU (gloo).
    Covers the self-spawn path in the CPU test suite; prints a line that says so."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub (no GPU work: launch-path rehearsal)",
                          "config": {"workload": "stub"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == args.stub_fail_rank:
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other BASELINE configs (default one-GPU run only)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock reading after the timed region (one-GPU runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of untimed launches before the warmup steps (clock ramp)")
    ap.add_argument("--samples-log2", type=int, default=None, help="override samples per GPU (2^k); rehearsals only")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on ONE GPU: gloo backend, every rank on cuda:0, halo staged through the host")
    ap.add_argument("--stub", action="store_true", help="launch-path rehearsal on the CPU: no GPU work, gloo, a sleep per step")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="(tests) this rank of a --stub run exits non-zero at the end")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # no launcher: start the ranks ourselves, before torch / the GPU are touched in this process
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.stub:
        return stub_rank(args, rank, world)

    import torch
    dev_index = 0 if args.rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI

    cfg = dict(WORKLOADS[args.workload])
    if args.samples_log2 is not None:
        cfg["n"] = 1 << args.samples_log2
    args.power_sample = world == 1 and not args.no_power and not args.stub
    main_res = measure(args, args.workload, cfg, rank, world, device, dist, args.steps, args.warmup,
                       with_cpu=(world == 1 and not args.no_cpu_baseline))
    args.power_sample = False
    bad = rank == 0 and (main_res.get("seams_verified") is False or not main_res["outputs_finite"])
    others = None
    if args.workload == DEFAULT_WORKLOAD and not args.no_others and (world > 1 or args.samples_log2 is None):
        # the other BASELINE configs, briefly (same protocol, fewer steps, no CPU leg): parity-test cases first, bench lines second.
        # One GPU: configs[1..3]; several GPUs: configs[4] (the cfg3 chain on a cf32 stream sharded over the ranks).
        others = {}
        args.power_sample = world == 1 and not args.no_power
        args.power_seconds = 1.0
        for name in (("cfg2", "cfg3", "cfg4", "cfg5") if world == 1 else ("cfg5",)):
            ocfg = dict(WORKLOADS[name])
            if args.samples_log2 is not None:
                ocfg["n"] = 1 << args.samples_log2
            r = measure(args, name, ocfg, rank, world, device, dist, 8, 2, with_cpu=False)
            if rank != 0:
                continue
            bad = bad or r.get("seams_verified") is False or not r["outputs_finite"]
            pw = r["roofline"].get("power") or {}
            others[name] = {"value": r["value"], "unit": "Msamples/s", "ms_per_step": r["ms_per_step"], "steps": 8, "warmup": 2,
                            "bound": r["roofline"]["bound"], "hbm_frac": r["roofline"]["hbm"]["frac"], "valu_frac": r["roofline"]["valu"]["frac"],
                            "valu_frac_at_sclk": r["roofline"]["valu"].get("frac_at_sclk"), "sclk_mhz": pw.get("sclk_mhz_mean"), "watts": pw.get("watts_mean"),
                            "kernel": r["roofline"]["kernel"], "kernel_ms": r["roofline"]["kernel_ms"], "config": workload_desc(ocfg),
                            **({"seams_verified": r["seams_verified"]} if r.get("seams_verified") is not None else {})}
        args.power_sample = False
    if rank == 0:
        line = {
            "metric": "Msamples/s through shift->FIR->FFT chain",
            "value": main_res["value"],
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {workload_desc(cfg)}", "samples_per_gpu": cfg["n"], "taps": cfg["lp"][2] if cfg["lp"] else 0,
                       "decimate": cfg["lp"][1] if cfg["lp"] else 1, "width": cfg["W"], "stride": cfg["S"],
                       "parallelism": f"window-range shards x{world}, halo {(cfg['W'] - cfg['S']) * (cfg['lp'][1] if cfg['lp'] else 1) + (cfg['lp'][2] if cfg['lp'] else 0)} samples",
                       "outputs_finite": main_res["outputs_finite"], "kernel_kind": main_res["kernel_kind"], "kernel_flags": main_res["kernel_flags"],
                       "tile_windows": main_res["tile_windows"], "threads": main_res["threads"],
                       # 1: first-order NCO correction, 2: second-order (streams whose phase n*|ratio| passes 2^28 rad: every multi-rank run of
                       # 2^31 samples per GPU — the plan describes the WHOLE stream — costs cfg3' 5.6 % per GPU against the one-rank run)
                       "nco_order": main_res["nco_order"]},
            "roofline": main_res["roofline"],
        }
        if main_res.get("seams_verified") is not None:
            if main_res.get("halo_exchange"):
                line["halo_exchange"] = main_res["halo_exchange"]
            line["seams_verified"] = main_res["seams_verified"]     # multi-rank: halo bytes + first / last windows of every rank re-derived locally
        for k in ("cpu_baseline", "cpu_allcores"):
            if k in main_res:
                line[k] = main_res[k]
        if others:
            line["others"] = others
        if world == 1 and args.workload == DEFAULT_WORKLOAD and not args.no_others:
            try:
                line["fast_mode"] = fast_mode_leg(cfg, device)
            except Exception as e:                       # informational leg: never takes the bench line down
                line["fast_mode"] = {"error": str(e)[:200]}
        if world == 1 and args.workload == DEFAULT_WORKLOAD and not args.no_others:
            try:
                line["write_sink"] = write_sink_leg(cfg, device)
            except Exception as e:                       # informational leg
                line["write_sink"] = {"error": str(e)[:200]}
        if world == 1 and args.workload == DEFAULT_WORKLOAD and args.samples_log2 is None and not args.no_others:
            try:
                line["no_lowpass"] = no_lowpass_leg(device)
            except Exception as e:                       # informational leg
                line["no_lowpass"] = {"error": str(e)[:200]}
        if world == 1 and args.workload == DEFAULT_WORKLOAD and args.samples_log2 is None and not args.no_others:
            try:
                line["end_to_end"] = end_to_end_host(device)
            except Exception as e:                       # the PCIe leg never takes the bench line down
                line["end_to_end"] = {"error": str(e)[:200]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        # a line whose seams did not verify or whose outputs are not finite is printed (it says so) but the run FAILS
        sys.stderr.write("bench.py: seams_verified false or non-finite outputs: failing the run\n")
        sys.exit(4)


if __name__ == "__main__":
    main()
