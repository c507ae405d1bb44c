"""dev: wall time of the C++ CLI twin on a 600 s stereo file (BASELINE C2 as `rocoder -i in.wav -o out.wav -w 16384 -f 8`)."""
import os, struct, subprocess, sys, time
import numpy as np
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
sr, ch = 44100, 2
n = sr * secs
x = (np.random.default_rng(0).uniform(-0.5, 0.5, (n, ch)) * 32767).astype("<i2")
data = x.tobytes()
hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data))
inp, outp = f"{d}/rc_in.wav", f"{d}/rc_out.wav"
open(inp, "wb").write(hdr + data)
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "rocoder_amd", "bin", "rocoder")
for i, extra in enumerate(([], [], ["--devices", "0"], ["--devices", "0"])):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-i", inp, "-o", outp, "-w", "16384", "-f", "8", *extra], capture_output=True, text=True, env=dict(os.environ, ROCODER_CLI_TIMING="1"))
    dt = time.perf_counter() - t0
    print(f"run {i} {extra}: rc={r.returncode} {dt:.2f} s, out {os.path.getsize(outp)/1e9:.2f} GB", r.stderr[-500:].replace("\n", " | "))
os.remove(inp); os.remove(outp)
