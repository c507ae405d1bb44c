"""A/B of two engine libraries over window lengths inside one call: python tools/ab_windows.py libA.so libB.so
(each library in its own subprocess, alternating; stereo, factor 8, default window, median per-launch time)"""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
child = r'''
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(%r))
import rocoder_amd
dev = torch.device("cuda", 0)
x = (torch.rand((2, 13_230_000), device=dev) - 0.5)
res = {}
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    for N in (1024, 2048, 4096, 8192, 32768, 65536):
        e = rocoder_amd.Engine(window_len=N, factor=8.0, channels=2, seed=1)
        out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            for _ in range(4):
                e.stretch_tensor(x, out=out)
            stream.synchronize()
        for _ in range(10):
            e.stretch_tensor(x, out=out)
        stream.synchronize()
        res[N] = round(statistics.median(e.kernel_times(10)), 4)
        e.close()
        del out
print(json.dumps(res))
''' % here
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, ROCODER_HIP_LIB=os.path.join(os.path.dirname(here), "rocoder_amd", lib))
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
