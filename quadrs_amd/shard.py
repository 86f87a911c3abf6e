"""Multi-GPU sharding of the stream: one process per GPU, contiguous window ranges.

The unit of work is an FFT window; window w is a pure function of source samples
[w*S*D, w*S*D + W*D + T) and of absolute sample indices (Shift is stateless in n, src/shift.rs:49),
so ranks need no data-path collective.  When the stream is device-resident and pre-split into
disjoint slabs, rank g additionally needs the first H = (W-S)*D + T samples of rank g+1's slab:
one neighbour send/recv (RCCL over xGMI with the `nccl` backend, gloo on CPU), a few KiB.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    w0: int            # first window of this rank
    w1: int            # one past the last window
    own_first: int     # first source sample this rank owns
    own_count: int     # samples it owns (disjoint across ranks, union = everything any window reads)
    halo: int          # samples it must fetch from rank+1 (0 on the last rank)

    @property
    def need_first(self):
        return self.own_first

    @property
    def need_count(self):
        return self.own_count + self.halo


def partition(n_windows, world, raw_step, raw_per_window, tile_windows=1):
    """Equal, contiguous, tile-aligned window ranges; returns [Shard] for every rank."""
    per = -(-n_windows // world)
    per = -(-per // tile_windows) * tile_windows
    bounds = [min(n_windows, g * per) for g in range(world + 1)]
    total_end = (n_windows - 1) * raw_step + raw_per_window if n_windows else 0
    shards = []
    for g in range(world):
        w0, w1 = bounds[g], bounds[g + 1]
        own_first = w0 * raw_step
        own_end = bounds[g + 1] * raw_step if g + 1 < world and bounds[g + 1] < n_windows else total_end
        if w1 == w0:
            own_first, own_end = total_end, total_end
        need_end = (w1 - 1) * raw_step + raw_per_window if w1 > w0 else own_first
        halo = max(0, need_end - own_end)
        shards.append(Shard(g, world, w0, w1, own_first, max(0, own_end - own_first), halo))
    return shards


def agree_tile_windows(tile_windows, dist, torch, device="cpu"):
    """Every rank's shard table must come from ONE tiling.  A plan's tile_windows depends on which kernel the plan obtained (a
    plan-time build that is cached on one rank and not on another, or fails on one, gives a different G), and ranks partitioning
    with different tilings would disagree about window ranges and halo sizes — a hang or a gap.  Rank 0's value is broadcast;
    a rank whose own plan tiles differently fails loudly unless rank 0's tiling is a multiple of its own (then its launches still
    start on its tile grid).  `dist` None (one rank): the value itself."""
    if dist is None:
        return int(tile_windows)
    t = torch.tensor([int(tile_windows)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    agreed = int(t.item())
    if agreed % int(tile_windows) != 0:
        raise RuntimeError(f"rank tiles by {tile_windows} windows, rank 0 by {agreed}: plans differ across ranks (plan-time kernel cache?)")
    return agreed


def alloc_slab(shard, bytes_per_sample, device, torch):
    """One buffer for everything `shard` reads: its own samples first, room for the halo behind them (filled by exchange)."""
    return torch.empty(shard.need_count * bytes_per_sample, dtype=torch.uint8, device=device)


def exchange(buf, shards, rank, bytes_per_sample, dist):
    """Performs the neighbour halo exchange for `rank`; every rank must call it.

    `buf` is either the rank's pre-sized slab (alloc_slab: need_count samples, the own part already filled) — the halo is
    then received straight into its tail and `buf` itself is returned, nothing is copied — or just the own samples, in
    which case a slab is allocated and the own part copied once (small streams, tests)."""
    import torch
    me = shards[rank]
    own_b, need_b = me.own_count * bytes_per_sample, me.need_count * bytes_per_sample
    if buf.numel() == need_b:
        slab = buf
    elif buf.numel() == own_b:
        slab = buf
        if me.halo and rank + 1 < len(shards):
            slab = torch.empty(need_b, dtype=torch.uint8, device=buf.device)
            slab[:own_b] = buf
    else:
        raise ValueError(f"buffer of {buf.numel()} bytes is neither the own part ({own_b}) nor the whole slab ({need_b})")
    ops = []
    if me.halo and rank + 1 < len(shards):
        ops.append(dist.P2POp(dist.irecv, slab[own_b:need_b], rank + 1))          # a contiguous view: lands in place
    if rank > 0 and shards[rank - 1].halo:
        h = shards[rank - 1].halo
        if h > me.own_count:
            raise ValueError("halo larger than the neighbour's slab: use fewer ranks or a bigger stream")
        ops.append(dist.P2POp(dist.isend, slab[: h * bytes_per_sample], rank - 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return slab
