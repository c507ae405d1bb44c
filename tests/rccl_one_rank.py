"""Helper of tests/test_gpu_multi.py::test_rccl_one_rank_walks_the_sharded_concat (run as a child process with a
time limit: a communicator that cannot be created must fail the test, not hang the suite).

One rank, backend "nccl" (RCCL on ROCm): communicator creation, the barrier / all-reduce bench.py's timing uses, the
broadcast of every shard into its view of the final tensor (dst=None), the root-only form (dst=0) and, with
stage_all, the grouped send / recv launch with the root's own shards as messages to itself - each result held against
the single-engine tensor bit for bit. Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rocoder_amd  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402  (input synthesis only)
from rocoder_amd.distributed import engine_compute, stretch_sharded  # noqa: E402

# WORLD_SIZE / RANK / LOCAL_RANK from the environment (the test starts one rank per GPU of the box: one on this pool)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", str(rank)))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
t = torch.tensor([3.5 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
res["all_reduce"] = float(t.item())
x = torch.from_numpy(np.stack([onp.synth_input(c, 400000) for c in range(2)])).to(dev)
with rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=3, device=local) as e:
    full = e.stretch_tensor(x).clone()  # every rank computes the whole job once: the reference the concat is held to
    wout = e.params.window_out_len
    nwin = full.shape[1] // wout
    comp = engine_compute(e, x)
    for name, kw in (("broadcast_all", dict(dst=None)), ("root_only", dict(dst=0)),
                     ("grouped_send_recv_to_self", dict(dst=0, stage_all=True))):
        has_result = kw["dst"] is None or rank == kw["dst"]  # root-only forms: the other ranks get no tensor
        got = stretch_sharded(comp, 2, nwin, wout, **kw)
        torch.cuda.synchronize()
        res[name] = bool(torch.equal(got, full)) if has_result else True
        buf = torch.full_like(full, float("nan")) if has_result else None
        got2 = stretch_sharded(comp, 2, nwin, wout, full=buf, **kw)
        torch.cuda.synchronize()
        res[name + "_into_caller_buffer"] = bool(got2 is buf and torch.equal(buf, full)) if has_result else True
        ok = torch.tensor([1.0 if (res[name] and res[name + "_into_caller_buffer"]) else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # rank 0's line speaks for every rank
        res[name] = res[name] and bool(ok.item() == 1.0)
    e.synchronize()
dist.barrier()
dist.destroy_process_group()
print(json.dumps(res))
