"""The multi-device entry (rc_multi_*) on ONE GPU listed 1, 2, 4 and 8 times against the one-engine call: what the
sharding itself costs (span copies, host threads, peer copies that are plain device copies here). BASELINE C2 job,
input and output resident on the device. Not a scaling measurement: every 'device' is the same GPU."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd
dev = torch.device("cuda", 0)
x = (torch.rand((2, 26_460_000), device=dev) - 0.5)
res = {}
kw = dict(window_len=16384, factor=8.0, channels=2, seed=1)
with rocoder_amd.Engine(**kw) as e:
    out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
    for _ in range(5):
        e.stretch_tensor(x, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); e.stretch_tensor(x, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res["one_engine_ms"] = round(1e3 * statistics.median(ts), 3)
# default: shares on the root's own device compute in place -> what is left is the host side of the sharding (persistent
# worker threads, one launch per share, one synchronisation per thread). staged: every share takes a remote device's
# path (span copy in, compute, shard copy out), which on one GPU adds a device-to-device copy of 7/8 of the output.
for staged in (False, True):
    for n in (1, 2, 4, 8):
        with rocoder_amd.MultiEngine([0] * n, **kw) as m:
            m.set_staging(staged)
            for _ in range(3):
                m.stretch_tensor(x, out=out)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); m.stretch_tensor(x, out=out); ts.append(time.perf_counter() - t0)
            res[f"multi_{n}x_same_gpu{'_staged' if staged else ''}_ms"] = round(1e3 * statistics.median(ts), 3)
print(json.dumps(res))
