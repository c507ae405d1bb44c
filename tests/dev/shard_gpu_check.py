"""RCCL sanity of the sharded stretch (run under torchrun on the GPU box): every rank computes its shards,
one all_gather concatenates; the result must equal the single-engine output bit for bit."""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
import rocoder_amd
from rocoder_amd.distributed import stretch_sharded, engine_compute
from oracle import oracle_np as onp
x = np.stack([onp.synth_input(c, 400000) for c in range(2)])
xt = torch.from_numpy(x).cuda()
with rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=3) as e:
    full = e.stretch_tensor(xt).clone()
    wout = e.params.window_out_len
    nwin = full.shape[1] // wout
    got = stretch_sharded(engine_compute(e, xt), 2, nwin, wout, dst=None)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "world", dist.get_world_size(), "equal:", bool(torch.equal(got, full)), tuple(got.shape))
dist.destroy_process_group()
