"""dev: phase stamps of hop4_kernel (RC_STAMP build, ROCODER_HIP_LIB=rocoder_amd/lib_stamp.so), C2 shape."""
import os, sys
import torch
sys.path.insert(0, '.')
import rocoder_amd
dev = torch.device("cuda", 0)
x = (torch.rand((2, 26_460_000), device=dev) - 0.5)
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1)
out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
for _ in range(3):
    e.stretch_tensor(x, out=out)
torch.cuda.synchronize()
print(open(os.environ["ROCODER_STAMPS"]).read())
