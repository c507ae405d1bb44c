#!/usr/bin/env python3
"""Benchmark of the stretch hot path on MI355X.

A "step" is one whole pass of the hot path over one synthetic job already resident in HBM:
BASELINE.json configs[1] — stereo 44.1 kHz, window 16384, factor 8, pitch 1, L = 26 460 000
samples per channel (600 s), i.e. 51 652 hops -> 423 133 184 output samples per step.
`value` = whole-job output Msamples/s over all ranks (weak scaling: every rank stretches its own
stereo job; channels/hop ranges are independent, there is no data-path collective).

The same JSON line carries
  roofline     — the dominant kernel (rc::hop3_kernel, the N=16384 fused hop kernel) priced on SURVEY §8(d4)'s
                 algorithmic READ bytes 4*N per hop against the 8 TB/s HBM peak, its launch duration
                 measured live with events on the stream it is launched on;
  cpu_baseline — the CPU restatement of the reference algorithm (oracle/rocoder_oracle.c, "port":
                 the Rust reference cannot be built here) on a bounded sample of the same workload,
                 1 DSP thread as in src/stretcher_processor.rs:55-71 — rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WINDOW = 16384
FACTOR = 8.0
PITCH = 1
CHANNELS = 2
SAMPLE_RATE = 44100
L_IN = 26_460_000
SEED = 0x5EED
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_on_device(torch, device, channels, length):
    """BASELINE.md §3 signal shape: 0.5 sin(2 pi 220 (c+1) t) + 0.05 u_c[t] (device-generated
    uniform noise; the parity tests use the documented splitmix64 stream, throughput does not
    depend on the noise bits)."""
    g = torch.Generator(device=device)
    g.manual_seed(0xC0DEC0DE)
    t = torch.arange(length, device=device, dtype=torch.float64) / SAMPLE_RATE
    rows = []
    for c in range(channels):
        s = 0.5 * torch.sin(2 * torch.pi * 220.0 * (c + 1) * t)
        u = torch.rand(length, device=device, generator=g, dtype=torch.float64) * 2 - 1
        rows.append((s + 0.05 * u).to(torch.float32))
    return torch.stack(rows).contiguous()


def pmc_traffic():
    """HBM-side bytes per hop-kernel launch from the committed rocprofv3 PMC passes of this same
    command (profiles/, separate --pmc runs: FETCH_SIZE, WRITE_SIZE in KiB). Per
    MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 tallies 128-B read requests at 64 B, so the read
    side is doubled; WRITE_SIZE is exact. Returns (bytes, source) or (None, None)."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    if not files:
        return None, None
    txt = open(files[-1]).read()
    f = re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt)
    w = re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt)
    if not (f and w):
        return None, None
    return (2.0 * float(f.group(1)) + float(w.group(1))) * 1024.0, os.path.relpath(files[-1], ROOT)


def cpu_baseline():
    """Time the oracle (a port of the reference algorithm, 1 thread for all channels) on a bounded
    sample: stereo, L = 15 000 000 per channel, same window/factor -> ~240 M output samples
    (about 15 s of CPU work)."""
    import numpy as np

    from oracle import cbind as oc
    from oracle import oracle_np as onp

    length = 15_000_000
    x = np.stack([onp.synth_input(c, length) for c in range(CHANNELS)])
    t0 = time.perf_counter()
    y = oc.stretch_offline(x, WINDOW, FACTOR, 1.0, PITCH, seed=SEED, sample_rate=SAMPLE_RATE)
    dt = time.perf_counter() - t0
    return {
        "value": round(y.size / dt / 1e6, 3),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"stereo L={length}/ch window={WINDOW} factor={FACTOR:g} -> {y.size} output samples "
                  f"in {dt:.1f}s; C restatement of rocoder's algorithm (oracle/), scalar libm, one DSP "
                  f"thread for all channels like src/stretcher_processor.rs:55-71; host has "
                  f"{os.cpu_count()} logical cores",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # Exactly ONE line on stdout (the JSON): RCCL / the HIP runtime print banners to fd 1, so park
    # the real stdout and point fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: RCCL barrier/all-reduce
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import rocoder_amd

    eng = rocoder_amd.Engine(window_len=WINDOW, factor=FACTOR, pitch_multiple=PITCH,
                             sample_rate=SAMPLE_RATE, channels=CHANNELS, seed=SEED + rank,
                             device=local_rank)
    x = synth_on_device(torch, device, CHANNELS, L_IN)
    n_out = eng.output_len(L_IN)
    out = torch.empty((CHANNELS, n_out), dtype=torch.float32, device=device)
    hops_per_step = (n_out * PITCH // (WINDOW // 2)) * CHANNELS

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    # a real (non-default) stream: the engine launches its kernel on exactly this stream, and the
    # events below are recorded on it, so they bracket the hop-kernel launches
    stream = torch.cuda.Stream(device)
    barrier()
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            eng.stretch_tensor(x, out=out)
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            eng.stretch_tensor(x, out=out)
        ev1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # per launch (one hop-kernel launch per step)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_samples = float(n_out) * CHANNELS * args.steps * world
        value = total_samples / dt / 1e6
        algo_bytes = hops_per_step * 4.0 * WINDOW  # 4N read bytes per hop
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic()
        res = {
            "metric": "output Msamples/s, 16384-win f=8 stereo (x CPU-realtime in config)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: stereo 44.1 kHz, window=16384, factor=8, pitch=1, "
                            "L=26460000/ch, inputs resident in HBM",
                "hops_per_step": hops_per_step,
                "output_samples_per_step": n_out * CHANNELS,
                "x_realtime": round(value * 1e6 / CHANNELS / SAMPLE_RATE / world, 1),
                "parallelism": f"{world} rank(s), one stereo job per GPU, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "rc::hop3_kernel<pitch1> (N=16384 fused hop kernel, 3 workgroups/CU)",
                "kernel_ms": round(kernel_ms, 4),
                "hops_per_s": round(hops_per_step / (kernel_ms * 1e-3), 1),
                "algorithmic_bytes_per_launch": algo_bytes,
                "note": "algorithmic READ bytes 4*N per hop (SURVEY §8 d4); the kernel is VALU-pipe bound "
                        "(FFT butterflies + per-bin hash/sincos: VALU busy 77 %) with LDS-exchange "
                        "synchronisation on top, see DESIGN.md §5",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            res["cpu_baseline"] = cb
            res["config"]["x_cpu"] = round(value / cb["value"], 1)
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
