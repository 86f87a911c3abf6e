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


def exchange(own, shards, rank, bytes_per_sample, dist):
    """Performs the neighbour halo exchange for `rank`; every rank must call it."""
    import torch
    me = shards[rank]
    ops, recv_buf = [], None
    if me.halo and rank + 1 < len(shards):
        recv_buf = torch.empty(me.halo * bytes_per_sample, dtype=torch.uint8, device=own.device)
        ops.append(dist.P2POp(dist.irecv, recv_buf, rank + 1))
    if rank > 0 and shards[rank - 1].halo:
        h = shards[rank - 1].halo
        if h > me.own_count:
            raise ValueError("halo larger than the neighbour's slab: use fewer ranks or a bigger stream")
        send_buf = own[: h * bytes_per_sample].contiguous()
        ops.append(dist.P2POp(dist.isend, send_buf, rank - 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if recv_buf is None:
        return own
    return torch.cat([own, recv_buf])
