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


def stretch_sharded(compute: Callable[[Shard], "object"], channels: int, total_windows: int,
                    window_out_len: int, group=None, dst: Optional[int] = 0):
    """Run this rank's shards with `compute(shard) -> tensor[ch_count, win_count*window_out_len]`
    and concatenate: returns the full [channels, total_windows*window_out_len] tensor on rank
    `dst` (every rank when dst is None), else None. One all_gather of equally padded segments."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    plan = shard_plan(channels, total_windows, world)
    mine = [s for s in plan if s.rank == rank]
    outs = [compute(s) for s in mine]
    per_rank = {}
    for s in plan:
        per_rank[s.rank] = per_rank.get(s.rank, 0) + s.ch_count * s.win_count * window_out_len
    seg = max(per_rank.values()) if per_rank else 0
    ref = outs[0] if outs else None
    device = ref.device if ref is not None else torch.device("cpu")
    if ref is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    buf = torch.zeros(seg, dtype=torch.float32, device=device)
    off = 0
    for o in outs:
        flat = o.reshape(-1)
        buf[off:off + flat.numel()] = flat
        off += flat.numel()
    if dst is None:
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
    else:
        gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, gathered, dst=dst, group=group)
        if rank != dst:
            return None
    full = torch.empty((channels, total_windows * window_out_len), dtype=torch.float32, device=device)
    offs = {r: 0 for r in range(world)}
    for s in plan:
        n = s.ch_count * s.win_count * window_out_len
        blk = gathered[s.rank][offs[s.rank]:offs[s.rank] + n].reshape(s.ch_count, -1)
        offs[s.rank] += n
        full[s.ch_first:s.ch_first + s.ch_count,
             s.win_first * window_out_len:(s.win_first + s.win_count) * window_out_len] = blk
    return full


def engine_compute(engine, x):
    """compute() for `stretch_sharded` backed by the HIP engine on this rank's GPU.
    x: torch float32 CUDA tensor [channels, L] (replicated input; it is small)."""
    import torch

    wout = engine.params.window_out_len

    def compute(s: Shard):
        out = torch.empty((s.ch_count, s.win_count * wout), dtype=torch.float32, device=x.device)
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
