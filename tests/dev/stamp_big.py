"""dev: phase stamps of big4_kernel (RC_STAMP build, ROCODER_HIP_LIB=rocoder_amd/lib_stamp.so): one C5-shaped job."""
import os, sys
import torch
sys.path.insert(0, '.')
import rocoder_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
x8 = (torch.rand((8, 5_292_000), device=dev) - 0.5)
e = rocoder_amd.Engine(window_len=N, factor=32.0, channels=8, seed=1)
out = torch.empty((8, e.output_len(x8.shape[1])), device=dev)
for _ in range(3):
    e.stretch_tensor(x8, out=out)
torch.cuda.synchronize()
print(open(os.environ["ROCODER_STAMPS"]).read())
