"""One rank's share of the fixed BASELINE C2 job on 1 / 2 / 4 / 8 GPUs, timed on THIS one GPU (rank 0's shard of
rocoder_amd.distributed.shard_plan: what bench.py --gpus N launches per rank and step). Median of the engine's
per-launch event times after a pre-heat. A/B of run plans: ROCODER_HIP_LIB=rocoder_amd/lib_x.so python tools/bench_shards.py"""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd
from rocoder_amd.distributed import engine_compute, shard_plan
import bench

dev = torch.device("cuda", 0)
x = bench.synth_on_device(torch, dev, 2, bench.L_IN)
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=bench.SEED)
wout = e.params.window_out_len
nwin = e.output_len(bench.L_IN) // wout
comp = engine_compute(e, x)
stream = torch.cuda.Stream(dev)
res = {"lib": os.environ.get("ROCODER_HIP_LIB", "product")}
with torch.cuda.stream(stream):
    for n in (1, 2, 4, 8):
        mine = [s for s in shard_plan(2, nwin, n) if s.rank == 0]
        bufs = {s: torch.empty((s.ch_count, s.win_count * wout), dtype=torch.float32, device=dev) for s in mine}
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(16):
                for s in mine:
                    comp(s, out=bufs[s])
            stream.synchronize()
        for _ in range(40):
            for s in mine:
                comp(s, out=bufs[s])
        stream.synchronize()
        ms = statistics.median(e.kernel_times(32))
        hops = sum(s.ch_count * s.win_count for s in mine) * e.params.hops_per_window
        res[f"1_of_{n}"] = {"hops": hops, "kernel_ms": round(ms, 4), "ns_per_hop": round(ms * 1e6 / hops, 2)}
        del bufs
print(json.dumps(res))
