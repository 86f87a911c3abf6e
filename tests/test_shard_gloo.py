"""world_size-2/3 gloo tests of the multi-rank path on CPU: partition + neighbour halo exchange
(the same torch.distributed calls bench.py issues over RCCL), checked with the oracle: the
concatenation of per-rank outputs must equal the single-rank result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, cfg, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from quadrs_amd import shard as SH
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shift, (fc, D, T), W, S = cfg
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((n, 2)) * 0.05).astype(np.float32)       # every rank can rebuild the stream
    dec_len = 1 + (n - T) // D
    n_windows = O.lib().qo_spark_window_count(dec_len, W, S)
    shards = SH.partition(n_windows, world, S * D, W * D + T, tile_windows=4)
    me = shards[rank]
    own = torch.from_numpy(x[me.own_first:me.own_first + me.own_count].copy()).view(torch.uint8).reshape(-1)
    full = SH.exchange(own, shards, rank, 8, dist)
    slab = full.numpy().view(np.float32).reshape(-1, 2)
    assert slab.shape[0] == me.need_count
    assert np.array_equal(slab, x[me.need_first:me.need_first + me.need_count])     # halo bytes are right
    # per-rank result with absolute indices: shift at the absolute offset, then per-window FIR + FFT
    ratio = O.shift_ratio(shift, 21_000_000)
    taps = O.taps(fc, 21_000_000, T)
    rows = []
    for w in range(me.w0, me.w1):
        a = w * S * D - me.need_first
        raw = O.shift_apply(slab[a:a + W * D + T], w * S * D, ratio)
        k, dec = O.lowpass_block(taps, D, raw)
        assert k == W
        rows.append(O.norm(O.fft(dec))[np.r_[W // 2:W, 0:W // 2]])
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array(rows, dtype=np.float32).reshape(-1, W))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg", [
    (2, (280000, (200_000, 32, 400), 64, 16)),     # cfg 5's chain: halo (W-S)*D + T = 1936 samples
    (3, (280000, (2_000_000, 16, 40), 128, 128)),  # cfg 2's chain: halo T = 40
])
def test_sharded_equals_whole(tmp_path, oracle, world, cfg):
    import torch.multiprocessing as mp
    n = 120_000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, cfg, str(tmp_path)), nprocs=world, join=True)
    shift, (fc, D, T), W, S = cfg
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((n, 2)) * 0.05).astype(np.float32)
    whole, _ = oracle.Chain.from_bytes(x.tobytes(), 0, 21_000_000).shift(shift).lowpass(fc, D, T).spark_fft(W, S)
    parts = np.concatenate([np.load(os.path.join(str(tmp_path), f"rank{r}.npy")) for r in range(world)])
    assert parts.shape == whole.shape
    assert np.array_equal(parts.view(np.uint32), whole.view(np.uint32))


def _hip_worker(rank, world, port, n, cfg, out_dir):
    """One rank of the HIP-path test: both ranks on cuda:0, gloo between them, the halo exchanged in place into the pre-sized
    slab (as bench.py does over RCCL), the rank's window range through the HIP chain kernel."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import quadrs_amd as Q
    from quadrs_amd import shard as SH
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    shift, (fc, D, T), W, S = cfg
    rng = np.random.default_rng(4321)
    x = (rng.standard_normal((n, 2)) * 0.05).astype(np.float32)       # every rank can rebuild the stream
    plan = Q.Plan(0, 21_000_000, n, shift_hz=shift, lowpass=(fc, D, T), width=W, stride=S)
    if rank == world:
        # the extra process: the single-rank reference run of the whole stream, also a child — the pytest process itself never
        # initialises HIP in this test, so the next parametrised case still forks its ranks from a GPU-free parent
        np.save(os.path.join(out_dir, "hip_whole.npy"), plan.run_host(x.tobytes()))
        plan.close()
        return
    dist.init_process_group("gloo", rank=rank, world_size=world)
    info = plan.info
    tile = SH.agree_tile_windows(int(info.tile_windows), dist, torch)       # one tiling for every rank's shard table (rank 0's; differing plans fail here)
    shards = SH.partition(plan.n_windows, world, info.raw_step, info.raw_per_window, tile)
    me = shards[rank]
    host = SH.alloc_slab(me, 8, "cpu", torch)
    host[:me.own_count * 8] = torch.from_numpy(x[me.own_first:me.own_first + me.own_count].copy()).view(torch.uint8).reshape(-1)
    got = SH.exchange(host, shards, rank, 8, dist)
    assert got.data_ptr() == host.data_ptr()                           # received in place: no second slab
    assert np.array_equal(host.numpy().view(np.float32).reshape(-1, 2), x[me.need_first:me.need_first + me.need_count])
    slab = host.cuda()
    nw = me.w1 - me.w0
    out = torch.empty(nw, W, dtype=torch.float32, device="cuda")
    plan.run_device(slab, out, me.w0, nw, src_first=me.need_first, src_count=me.need_count)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"hip_rank{rank}.npy"), out.cpu().numpy())
    np.save(os.path.join(out_dir, f"hip_kind{rank}.npy"), np.array([int(info.kernel_kind), int(info.tile_windows)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.spawns_gpu_ranks
@pytest.mark.parametrize("world,cfg,n", [
    (2, (280000, (200_000, 32, 400), 64, 16), 3_000_000),      # cfg 5's chain (built-in long-filter kernel, 27-window tiles): halo 1936 samples
    (2, (280000, (200_000, 32, 200), 128, 128), 6_000_000),    # the north_star chain (built-in kernel): halo 200 samples
    (3, (280000, (2_000_000, 16, 40), 128, 128), 2_000_000),   # cfg 2's chain, three ranks
])
def test_sharded_hip_ranks_equal_single_rank(tmp_path, world, cfg, n):
    """The multi-rank path with the HIP kernels: `world` rank processes (gloo, all on cuda:0) each run their window range from
    their own slab + exchanged halo; the concatenation must equal the single-rank HIP run (made by one more child process) bit for
    bit.  Scheduled before any test that initialises the GPU in this process (conftest.py: spawns_gpu_ranks), and the test itself
    makes no HIP call in the pytest process, so every parametrised case starts its ranks from a GPU-free parent."""
    import torch.multiprocessing as mp
    port = _free_port()
    # `world` ranks + one process for the single-rank reference run: the parent only compares files (no HIP call in this process)
    mp.spawn(_hip_worker, args=(world, port, n, cfg, str(tmp_path)), nprocs=world + 1, join=True)
    whole = np.load(os.path.join(str(tmp_path), "hip_whole.npy"))
    parts = np.concatenate([np.load(os.path.join(str(tmp_path), f"hip_rank{r}.npy")) for r in range(world)])
    assert parts.shape == whole.shape
    assert np.array_equal(parts.view(np.uint32), whole.view(np.uint32))


def test_partition_properties():
    from quadrs_amd import shard as SH
    for n_windows, world, step, rpw, tile in ((65535, 8, 2048, 2088, 2), (16777212, 8, 512, 2448, 4), (10, 4, 8, 40, 1),
                                              (3, 8, 512, 2448, 4), (995, 2, 2, 4, 64)):
        sh = SH.partition(n_windows, world, step, rpw, tile)
        assert sh[0].w0 == 0 and sh[-1].w1 == n_windows
        assert all(a.w1 == b.w0 for a, b in zip(sh, sh[1:]))
        total_end = (n_windows - 1) * step + rpw
        live = [s for s in sh if s.w1 > s.w0]
        assert sum(s.own_count for s in live) == total_end                  # disjoint cover
        for s in live:
            assert s.need_first == s.w0 * step
            assert s.need_first + s.need_count == (s.w1 - 1) * step + rpw
        assert live[-1].halo == 0
    sh = SH.partition(16777212 * 2 + 4, 8, 512, 2448, 4)                     # cfg 5: 1936-sample halo
    assert {s.halo for s in sh[:-1]} == {1936}
