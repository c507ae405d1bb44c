"""Multi-GPU sharding of one stretch job: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The path shards naturally: channels are independent Stretchers
(src/main.rs:133-153) and, given the counter-based phase source, hops are independent too,
coupled only through the two-term overlap-add — a shard recomputes the one hop before its
range instead of exchanging it. The ONLY collective is the final concatenation of output
segments (gather / all-gather of contiguous f32 segments); there is no all-reduce anywhere.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional


@dataclass(frozen=True)
class Shard:
    rank: int
    ch_first: int
    ch_count: int
    win_first: int
    win_count: int


def shard_plan(channels: int, total_windows: int, world_size: int) -> List[Shard]:
    """Units are (channel, contiguous window range): whole next_window() windows, so every rank
    emits whole windows. The channel-major sequence of channels*windows windows is cut into
    world_size equal contiguous pieces; a piece becomes at most three shards (partial first
    channel, a block of whole channels, partial last channel). 8 channels on 8 ranks is one whole
    channel per rank; 2 channels on 8 ranks is a quarter of a channel per rank."""
    assert channels >= 1 and world_size >= 1 and total_windows >= 0
    shards: List[Shard] = []
    total = channels * total_windows
    for r in range(world_size):
        lo = r * total // world_size
        hi = (r + 1) * total // world_size
        pieces = []  # (channel, w0, w1)
        while lo < hi:
            c, w0 = divmod(lo, total_windows)
            w1 = min(total_windows, w0 + (hi - lo))
            pieces.append((c, w0, w1))
            lo += w1 - w0
        i = 0
        while i < len(pieces):
            c, w0, w1 = pieces[i]
            j = i
            if w0 == 0 and w1 == total_windows:  # merge consecutive whole channels
                while (j + 1 < len(pieces) and pieces[j + 1][1] == 0
                       and pieces[j + 1][2] == total_windows):
                    j += 1
            shards.append(Shard(r, c, j - i + 1, w0, w1 - w0))
            i = j + 1
    return shards


def shard_view(full, s: Shard, window_out_len: int):
    """The block of the final [channels, windows * window_out_len] layout a shard fills. A shard is
    either part of ONE channel or a block of WHOLE channels (shard_plan), so in a contiguous `full`
    the block is one contiguous span of memory: collectives can land in it directly."""
    return full[s.ch_first:s.ch_first + s.ch_count,
                s.win_first * window_out_len:(s.win_first + s.win_count) * window_out_len]


def stretch_sharded(compute: Callable[..., "object"], channels: int, total_windows: int,
                    window_out_len: int, group=None, dst: Optional[int] = 0, full=None, stage_all: bool = False):
    """Run this rank's shards with `compute(shard, out=None) -> tensor[ch_count, win_count*window_out_len]`
    and concatenate: returns the full [channels, total_windows*window_out_len] tensor on rank `dst`
    (every rank when dst is None), else None.

    The concat is the path's only collective (north_star: RCCL for the multi-channel concat only) and
    moves every output byte exactly once, straight into its final place: a rank that holds `full`
    computes its own shards INTO their views of it; every other shard travels as one message into its
    view (grouped send / recv to the root, or one broadcast per shard when every rank wants the
    result). No padding to the longest segment, no staging list, no block-by-block re-copy.
    `full` may be passed in (reused across calls); it must be contiguous.
    `dst` is a rank OF `group` (0 .. group size - 1), like the `rank` fields of the shard plan - not a global rank;
    with a sub-group, translate a global rank with `dist.get_group_rank(group, global_rank)` first.
    `stage_all` (diagnostic, the twin of rc_multi_set_staging): the root's OWN shards are computed into separate
    buffers and travel as messages too (a grouped send to itself), so a single rank walks the whole send / recv code."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert dst is None or 0 <= dst < world, f"dst={dst} is not a rank of the group (size {world})"
    plan = shard_plan(channels, total_windows, world)
    holds_full = dst is None or rank == dst
    stage_all = stage_all and dst is not None
    width = total_windows * window_out_len
    outs = {}
    device = None
    if holds_full and full is not None:
        device = full.device
    for s in plan:
        if s.rank != rank:
            continue
        if holds_full and full is not None and not stage_all:
            outs[s] = compute(s, out=shard_view(full, s, window_out_len))
        else:
            outs[s] = compute(s)
            device = outs[s].device
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device())
                  if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    if holds_full and full is None:
        full = torch.empty((channels, width), dtype=torch.float32, device=device)
        if not stage_all:
            for s, o in outs.items():  # first call without a caller buffer: one placement copy
                shard_view(full, s, window_out_len).copy_(o)
    if holds_full:
        assert full.is_contiguous() and tuple(full.shape) == (channels, width)
    if dst is None:
        for s in plan:  # every rank receives every foreign shard into its final place
            v = shard_view(full, s, window_out_len)
            assert v.is_contiguous() or s.ch_count == 1 or s.win_count == total_windows
            dist.broadcast(v, src=dist.get_global_rank(group, s.rank) if group is not None else s.rank,
                           group=group)
        return full
    ops = []
    for s in plan:
        if s.rank == dst and not stage_all:
            continue
        peer_dst = dist.get_global_rank(group, dst) if group is not None else dst
        peer_src = dist.get_global_rank(group, s.rank) if group is not None else s.rank
        if rank == dst:
            ops.append(dist.P2POp(dist.irecv, shard_view(full, s, window_out_len), peer_src, group))
        if rank == s.rank:
            o = outs[s]
            ops.append(dist.P2POp(dist.isend, o if o.is_contiguous() else o.contiguous(), peer_dst, group))
    if ops:
        if dist.get_backend(group) == "nccl":
            for w in dist.batch_isend_irecv(ops):  # one grouped RCCL launch
                w.wait()
        else:
            # gloo (CPU tests) has no message to oneself: a stage_all root pairs its own sends and receives, which
            # were appended in the same plan order, as plain copies
            me = dist.get_rank()
            own_recv = [op.tensor for op in ops if op.peer == me and op.op is dist.irecv]
            own_send = [op.tensor for op in ops if op.peer == me and op.op is dist.isend]
            for dst_t, src_t in zip(own_recv, own_send):
                dst_t.copy_(src_t)
            for w in [op.op(op.tensor, op.peer, group=op.group) for op in ops if op.peer != me]:
                w.wait()
    return full if rank == dst else None


def engine_compute(engine, x):
    """compute() for `stretch_sharded` backed by the HIP engine on this rank's GPU.
    x: torch float32 CUDA tensor [channels, L] (replicated input; it is small). With `out` (a view
    of the final layout, any row stride) the engine writes the shard straight into it."""
    import torch

    wout = engine.params.window_out_len

    def compute(s: Shard, out=None):
        if out is None:
            out = torch.empty((s.ch_count, s.win_count * wout), dtype=torch.float32, device=x.device)
        assert out.stride(1) == 1 and tuple(out.shape) == (s.ch_count, s.win_count * wout)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if stream == 0:  # see Engine.stretch_tensor
            torch.cuda.current_stream(x.device).synchronize()
        engine.stretch_device_range_ptr(x.data_ptr(), x.stride(0), x.shape[1], s.ch_first, s.ch_count,
                                        s.win_first, s.win_count, out.data_ptr(), out.stride(0),
                                        out.shape[1], stream)
        if stream == 0:
            engine.synchronize()
        return out

    return compute
