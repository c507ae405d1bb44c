"""CPU tests of the multi-GPU path: the shard plan and the one collective (segment gather), run
with world_size 2/3 on the gloo backend. The per-shard compute is injected: here it is the CPU
oracle (tests may use it as the checker); on a GPU node it is rocoder_amd.distributed.engine_compute."""
import os
import socket

import numpy as np
import pytest

from rocoder_amd.distributed import Shard, shard_plan


@pytest.mark.parametrize("channels,windows,world", [(2, 12913, 1), (2, 12913, 2), (2, 12913, 4),
                                                    (2, 12913, 8), (8, 2553, 8), (8, 2553, 4),
                                                    (2, 7, 3), (1, 1, 8), (3, 10, 2), (2, 0, 4)])
def test_shard_plan_partitions_every_window_once(channels, windows, world):
    plan = shard_plan(channels, windows, world)
    cover = np.zeros((channels, max(windows, 1)), np.int32)
    for s in plan:
        assert 0 <= s.rank < world and s.ch_count >= 1 and s.win_count >= 1
        cover[s.ch_first:s.ch_first + s.ch_count, s.win_first:s.win_first + s.win_count] += 1
    if windows:
        assert np.all(cover == 1)
    # balanced: no rank holds more than ceil(units/world) + 1 windows-worth of a fair share
    if windows >= world and plan:
        load = np.zeros(world)
        for s in plan:
            load[s.rank] += s.ch_count * s.win_count
        assert load.max() <= np.ceil(channels * windows / world) + channels


def test_baseline_c5_is_one_channel_per_gpu():  # 8 channels -> 1 channel/GPU, zero halo
    plan = shard_plan(8, 2553, 8)
    assert [(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in plan] == \
        [(r, r, 1, 0, 2553) for r in range(8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, x, full_ref, N, f, p, wout, nwin, dst, q):
    import torch
    import torch.distributed as dist

    from rocoder_amd.distributed import stretch_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def compute(s: Shard, out=None):
            # checker-side compute: the shard's slice of the oracle's full output
            blk = torch.from_numpy(np.ascontiguousarray(
                full_ref[s.ch_first:s.ch_first + s.ch_count,
                         s.win_first * wout:(s.win_first + s.win_count) * wout]))
            if out is not None:
                out.copy_(blk)
                return out
            return blk

        # twice: without a caller buffer, then into a reused one (the bench's form)
        out = stretch_sharded(compute, x.shape[0], nwin, wout, dst=dst)
        if dst is None or rank == dst:
            buf = torch.full_like(out, float("nan"))
            out2 = stretch_sharded(compute, x.shape[0], nwin, wout, dst=dst, full=buf)
            assert out2 is buf and torch.equal(out2, out)
        else:
            assert stretch_sharded(compute, x.shape[0], nwin, wout, dst=dst) is None
        if dst is not None:
            # stage_all: the root's own shards travel as messages to itself too (what the one-GPU RCCL test walks)
            buf3 = torch.full((x.shape[0], nwin * wout), float("nan")) if rank == dst else None
            out3 = stretch_sharded(compute, x.shape[0], nwin, wout, dst=dst, full=buf3, stage_all=True)
            assert (out3 is buf3 and torch.equal(out3, out)) if rank == dst else out3 is None
        if dst is None or rank == dst:
            q.put((rank, np.array_equal(out.numpy(), full_ref)))
        else:
            q.put((rank, out is None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dst", [(2, 0), (3, None), (3, 1)])
def test_sharded_gather_gloo(world, dst):
    import torch.multiprocessing as mp

    from oracle import cbind as oc
    from oracle import oracle_np as onp

    N, f, p = 256, 4.0, 1
    x = np.stack([onp.synth_input(c, 6000) for c in range(2)])
    full = oc.stretch_offline(x, N, f, 1.0, p, seed=3)
    wout = N
    nwin = full.shape[1] // wout
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, x, full, N, f, p, wout, nwin, dst, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(ok for _, ok in res), res


def test_multi_device_test_lists_widen_with_the_device_count():
    """tests/test_gpu_multi.py builds its device lists and the RCCL world from rc_device_count()
    (tests/multi_devices.py). With the count faked: on a four-GPU box every multi-device test also runs on lists of
    DISTINCT devices (neighbouring shards on different GPUs: real peer copies) and the RCCL test starts four ranks; on
    this pool's one-GPU boxes the lists and the test count are what they were."""
    from multi_devices import device_lists, rccl_world

    assert device_lists(2, 1) == [[0, 0]] and device_lists(8, 1) == [[0] * 8]
    assert device_lists(2, 4) == [[0, 0], [0, 1]]
    assert device_lists(3, 4) == [[0, 0, 0], [0, 1, 2]]
    assert device_lists(8, 4) == [[0] * 8, [0, 1, 2, 3, 0, 1, 2, 3]]
    assert device_lists(8, 8) == [[0] * 8, list(range(8))]
    for n_have in (2, 4, 8):
        for n_dev in (2, 3, 4, 8):
            spread = device_lists(n_dev, n_have)[-1]
            assert len(spread) == n_dev and max(spread) == min(n_dev, n_have) - 1
            assert all(a != b for a, b in zip(spread, spread[1:]))  # neighbours never share a device
    assert [rccl_world(n) for n in (1, 2, 4, 8, 16)] == [1, 2, 4, 8, 8]
    # the shard plan the lists are indexed by: shard i of an n-way cut runs on ids[i] - with the spread list on 4 GPUs
    # a stereo job cut 4 ways puts its four half-channels on four devices
    from rocoder_amd.distributed import shard_plan

    ids = device_lists(4, 4)[-1]
    assert sorted({ids[s.rank] for s in shard_plan(2, 1000, 4)}) == [0, 1, 2, 3]
