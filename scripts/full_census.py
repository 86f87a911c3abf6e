"""Full-size parity census: EVERY window of a BASELINE workload at its full size, HIP chain kernel against the CPU oracle (the
cheapest exact form of the oracle, all host cores: FIR at the decimated positions only, same products, same order).  The GPU
suite compares a sample of windows at full size (tests/test_gpu_parity.py::test_full_size_chains); this compares all of them.
Test infrastructure (uses oracle/): never imported by the product.   usage: python scripts/full_census.py cfg2 cfg3p cfg3 cfg4"""
import concurrent.futures as cf
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import quadrs_amd as Q
from oracle import oracle as O

dev = torch.device("cuda", 0)
cores = max(1, min(len(os.sched_getaffinity(0)), 32))
for name in sys.argv[1:]:
    cfg = bench.WORKLOADS[name]
    t0 = time.perf_counter()
    if name == "cfg4":
        src = torch.empty(cfg["n"], 2, dtype=torch.float32, device=dev)
        tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
        for a in range(0, cfg["n"], 1 << 28):
            Q.gen_device(tones, cfg["sr"], a, src[a:a + (1 << 28)])
    else:
        src = bench.synth_slab(torch, cfg["fmt"], 0, cfg["n"], 0x5EED0002, dev)
    p = Q.Plan(cfg["fmt"], cfg["sr"], cfg["n"], shift_hz=cfg["shift"], lowpass=cfg["lp"], width=cfg["W"], stride=cfg["S"])
    out = torch.empty(p.n_windows, cfg["W"], dtype=torch.float32, device=dev)
    p.run_device(src, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    host = src.view(torch.uint8).reshape(-1).cpu().numpy()
    del src, out
    torch.cuda.empty_cache()
    t_gpu = time.perf_counter() - t0
    # the oracle on the very same bytes, absolute sample indices; no copy of the slab (the C side reads through the pointer)
    node = O.lib().qo_source_mem(O._p(host), host.size, cfg["fmt"], cfg["sr"])
    ch = O.Chain(); ch._node = node; ch._keep.append(host)
    if cfg["shift"] is not None:
        ch = ch.shift(cfg["shift"])
    ch = ch.lowpass(*cfg["lp"])
    total = O.lib().qo_spark_window_count(ch.len(), cfg["W"], cfg["S"])
    assert total == p.n_windows, (total, p.n_windows)
    piece = 8192
    jobs = [(a, min(piece, total - a)) for a in range(0, total, piece)]
    stats = dict(windows_differ=0, bins_differ=0, worst_ulp=0.0, first=None)

    def work(job):
        a, n = job
        ref, _ = ch.spark_fft(cfg["W"], cfg["S"], first_window=a, max_windows=n, want_codes=False)
        g = got[a:a + n]
        ne = ref.view(np.uint32) != g.view(np.uint32)
        if not ne.any():
            return 0, 0, 0.0, None
        # ulp distance measured against the window maximum (the tolerance the tests state for chains with a shift stage)
        scale = np.spacing(np.abs(ref).max(axis=1, keepdims=True).astype(np.float32)).astype(np.float64)
        err = (np.abs(ref.astype(np.float64) - g.astype(np.float64)) / scale).max()
        rows = np.nonzero(ne.any(axis=1))[0]
        return len(rows), int(ne.sum()), float(err), int(a + rows[0])

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        for nw, nb, err, first in ex.map(work, jobs):
            stats["windows_differ"] += nw; stats["bins_differ"] += nb; stats["worst_ulp"] = max(stats["worst_ulp"], err)
            if first is not None and stats["first"] is None:
                stats["first"] = first
    t_cpu = time.perf_counter() - t0
    print(f"{name}: kernel kind {p.info.kernel_kind}, {total} windows x {cfg['W']} bins ({cfg['n']} samples): windows differing from the oracle "
          f"{stats['windows_differ']}, bins {stats['bins_differ']}, worst deviation {stats['worst_ulp']:.2f} ulp of the window maximum"
          f"{'' if stats['first'] is None else ', first at window %d' % stats['first']}   [gpu side {t_gpu:.0f} s, oracle {t_cpu:.0f} s on {cores} threads]", flush=True)
    del ch, host, got
