#!/usr/bin/env python3
"""Benchmark of the stretch hot path on MI355X.

A "step" is one whole pass of the hot path over one synthetic job already resident in HBM. At N = 1 the
job is BASELINE.json configs[1] — stereo 44.1 kHz, window 16384, factor 8, pitch 1, L = 26 460 000
samples per channel (600 s): 51 652 hops -> 423 133 184 output samples per step. At N > 1 it is ONE
stereo job of N x L samples per channel cut by rocoder_amd.distributed.shard_plan into (channel, hop
range) shards, one rank per GPU (N = 2: a channel per GPU; N = 4, 8: half / quarter channels; every
rank recomputes the single hop before its range): per-GPU work is that of N = 1 (weak scaling) and the
data path has no collective. `value` = output samples of all ranks / max-rank time of the timed steps,
outputs left sharded in HBM; the cost of the one optional collective, the RCCL concat of the shards on
rank 0, is measured in a second timed region and reported beside it (config.concat); at N > 1 BASELINE
configs[4] (C5: 8 channels, window 65536) cut over the same ranks is timed as well (config.c5_sharded). Both
extras run after the main line is complete, under a watchdog, so they can never cost the measurement.
The metric names a FIXED workload ("16384-win f=8 stereo @1/2/4/8 GPU"), so a second timed region cuts the N = 1
job itself (L = 26 460 000 per channel, whatever N is) over the N ranks with the same shard plan: config.strong =
{ms_per_step, value_Msamples_s, ms_per_step_1gpu (the same job on this rank's GPU alone, same run),
efficiency_vs_1gpu}. `scaling` stays "weak" for `value`; config.strong.scaling says "strong".

The same JSON line carries
  roofline     — the dominant kernel (the N = 16384 fused hop kernel) priced on SURVEY §8(d4)'s
                 algorithmic READ bytes 4*N per hop against the 8 TB/s HBM peak; its duration is the
                 MEDIAN of the per-launch HIP-event times the engine records around the kernel itself
                 (rc_engine_kernel_times), after >= 2 s of back-to-back launches (steady clocks);
                 also against the copy bandwidth measured on this very device, and the total-traffic,
                 compulsory and LDS-traffic figures of SURVEY §8 d3/d4;
  cpu_baseline — the CPU path of the reference algorithm written for speed (oracle/
                 rocoder_cpu_baseline.c, "port": the Rust reference cannot be built here), one DSP thread
                 as in src/stretcher_processor.rs:55-71, on a bounded sample of the same workload;
  cpu_baseline_all_cores — the same code on every core this process may use (rank 0, N = 1 only).
"""
import argparse
import json
import os
import statistics
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
CONCAT_TIMEOUT_S = float(os.environ.get("ROCODER_BENCH_CONCAT_TIMEOUT", "120"))  # watchdog of the post-measurement concat region (N > 1)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
LDS_READ_PEAK_GBS = 150e3    # MI355X_MICROARCH.md §LDS: ds_read_b64/b128, every CU streaming
LDS_WRITE_PEAK_GBS = 45e3    # same: 38-51 TB/s for ds_write_b32..b128
# rc_calib_valu on the reference box of profiles/README.md's normalised table (ns per packed-FMA wave instruction per
# SIMD, eight waves per SIMD): kernel_ms_normalised = kernel_ms * REF / this box's value
BOX_CALIB_REF_NS = 1.74
# reference shader clock for kernel_ms_at_ref_clock (MHz): what a typical box of the pool holds under the hop kernel
SCLK_REF_MHZ = 2100.0


def synth_on_device(torch, device, channels, length):
    """BASELINE.md §3 signal shape: 0.5 sin(2 pi 220 (c+1) t) + 0.05 u_c[t] (device-generated
    uniform noise; the parity tests use the documented splitmix64 stream, throughput does not
    depend on the noise bits)."""
    g = torch.Generator(device=device)
    g.manual_seed(0xC0DEC0DE)
    rows = []
    for c in range(channels):
        t = torch.arange(length, device=device, dtype=torch.float64) / SAMPLE_RATE
        s = 0.5 * torch.sin(2 * torch.pi * 220.0 * (c + 1) * t)
        del t
        u = torch.rand(length, device=device, generator=g, dtype=torch.float64) * 2 - 1
        rows.append((s + 0.05 * u).to(torch.float32))
        del s, u
    return torch.stack(rows).contiguous()


def pmc_traffic(kernel_id):
    """HBM-side bytes per hop-kernel launch from the committed rocprofv3 PMC passes of this same
    command (profiles/, separate --pmc runs: FETCH_SIZE, WRITE_SIZE in KiB). Per
    MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 tallies 128-B read requests at 64 B, so the read
    side is doubled; WRITE_SIZE is exact. Only a summary that names the kernels this library holds
    (`kernel_id:` line) is quoted: counters of an older kernel say nothing about this one.
    Returns (bytes, source) or (None, reason)."""
    import glob
    import re

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")), reverse=True):
        txt = open(path).read()
        kid = re.search(r"^#?\s*kernel_id:\s*(\S+)", txt, re.M)
        if not kid or kid.group(1) != kernel_id:
            continue
        f = re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt)
        w = re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt)
        if f and w:
            return (2.0 * float(f.group(1)) + float(w.group(1))) * 1024.0, os.path.relpath(path, ROOT)
    return None, f"no profiles/r*_pmc_summary.txt for kernel_id {kernel_id}"


def pmc_valu(kernel_id):
    """VALU occupancy and instruction count of the hop kernel from the same committed PMC summary (or None):
    SQ_ACTIVE_INST_VALU counts quad-cycles summed over all waves; 1024 SIMDs x the launch's busy cycles is what the
    chip offers. GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    import glob
    import re

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")), reverse=True):
        txt = open(path).read()
        kid = re.search(r"^#?\s*kernel_id:\s*(\S+)", txt, re.M)
        if not kid or kid.group(1) != kernel_id:
            continue
        g = {k: float(v) for k, v in re.findall(r"^(\w+)\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt, re.M)}
        if {"SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"} <= set(g):
            simd_cycles = g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            return {"valu_busy": round(4.0 * g["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
                    "valu_insts_per_launch": g["SQ_INSTS_VALU"], "source": os.path.relpath(path, ROOT)}
    return None


def cpu_baselines(all_cores=True):
    """Time oracle/rocoder_cpu_baseline.c (the reference's per-hop work with an optimised FFT) on
    bounded samples of the workload: one thread for all channels, then every core available."""
    import numpy as np

    from oracle import cbind as oc
    from oracle import oracle_np as onp

    def run(length, threads):
        x = np.stack([onp.synth_input(c, length) for c in range(CHANNELS)])
        t0 = time.perf_counter()
        y = oc.cpu_baseline_stretch(x, WINDOW, FACTOR, 1.0, PITCH, seed=SEED, threads=threads)
        dt = time.perf_counter() - t0
        return y.size, dt

    what = ("CPU path of rocoder's algorithm written for speed (oracle/rocoder_cpu_baseline.c: full N-point "
            "complex FFTs as src/fft.rs:59,69 with a radix-4 Stockham FFT, scalar libm hypotf/sincosf per bin "
            "as src/fft.rs:65-68), not the rocoder binary (no Rust toolchain)")
    n1, t1 = run(8_000_000, 1)
    one = {
        "value": round(n1 / t1 / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"stereo L=8000000/ch window={WINDOW} factor={FACTOR:g} -> {n1} output samples in {t1:.1f}s; "
                  f"one DSP thread for all channels like src/stretcher_processor.rs:55-71; {what}",
    }
    many = None
    if all_cores:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(cores, 64))
        length = 8_000_000 if cores < 8 else 24_000_000
        nm, tm = run(length, cores)
        many = {
            "value": round(nm / tm / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"stereo L={length}/ch -> {nm} output samples in {tm:.1f}s; OpenMP, one task per (channel, "
                      f"hop range), the hop before a range recomputed; host reports {os.cpu_count()} logical cores",
        }
    return one, many


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat-s", type=float, default=2.0,
                    help="seconds of back-to-back launches before the timed steps (steady DVFS state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the copy microbench, the PCIe-inclusive run and the concat region")
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
    # ROCODER_BENCH_REHEARSAL=1 (dev only, never the driver's command): every rank on cuda:0 over gloo, to walk the
    # N > 1 shard plan / timing code on a one-GPU box. RCCL refuses two ranks on one device, so the concat region
    # reports an error there; the JSON line is marked "rehearsal" and is not a measurement.
    rehearsal = os.environ.get("ROCODER_BENCH_REHEARSAL") in ("1", "hang")  # "hang": stall the concat region (watchdog test)
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: RCCL barrier/all-reduce
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def note(msg):  # progress on stderr in rehearsals only
        if rehearsal:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def max_over_ranks(v):
        if dist is None:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    import rocoder_amd
    from rocoder_amd import _lib
    from rocoder_amd.distributed import engine_compute, shard_plan, shard_view, stretch_sharded

    kernel_id = _lib.lib().rc_kernel_id().decode()
    # ONE job for all ranks: same seed, same (replicated, device-generated) input of world x L samples
    length = L_IN * world
    eng = rocoder_amd.Engine(window_len=WINDOW, factor=FACTOR, pitch_multiple=PITCH,
                             sample_rate=SAMPLE_RATE, channels=CHANNELS, seed=SEED, device=dev_index)
    x = synth_on_device(torch, device, CHANNELS, length)
    wout = eng.params.window_out_len
    n_out = eng.output_len(length)          # per channel, whole job
    nwin = n_out // wout
    plan = shard_plan(CHANNELS, nwin, world)
    mine = [s for s in plan if s.rank == rank]
    compute = engine_compute(eng, x)
    # this rank's shards stay in its own HBM (no collective in the timed steps)
    bufs = {s: torch.empty((s.ch_count, s.win_count * wout), dtype=torch.float32, device=device) for s in mine}
    my_samples = sum(b.numel() for b in bufs.values())
    hops_mine = my_samples * PITCH // (WINDOW // 2)

    def step():
        for s in mine:
            compute(s, out=bufs[s])

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    # a real (non-default) stream: the engine launches its kernels on exactly this stream
    stream = torch.cuda.Stream(device)
    note(f"plan {[(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in plan]}")
    barrier()
    note("first barrier passed")
    # shader clock under THIS load, sampled by a host thread while the pre-heat and the timed steps run (boxes of the
    # pool hold 2.0 - 2.15 GHz under the hop kernel; the calibration kernel below alone does not show it)
    import threading

    clk_samples, clk_stop = [], threading.Event()

    def clk_sampler():
        while not clk_stop.is_set():
            try:
                clk_samples.append((time.perf_counter(), float(torch.cuda.clock_rate(dev_index))))
            except Exception:  # noqa: BLE001  (no SMI library on the box: the field stays null)
                return
            time.sleep(0.05)

    clk_thread = threading.Thread(target=clk_sampler, daemon=True)
    clk_thread.start()
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        # pre-heat: >= preheat_s of back-to-back launches so the timed steps run at the clock the chip
        # holds under this load (MI355X_MICROARCH.md, DVFS), not at the boost clock of an idle chip
        t_heat = time.perf_counter()
        n_heat = 0
        while time.perf_counter() - t_heat < args.preheat_s:
            for _ in range(32):
                step()
            n_heat += 32
            stream.synchronize()
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        dt = time.perf_counter() - t0
        # a kernel that gave up inside the timed region (seam wait expired: output incomplete) reports through the
        # engine's device error word; synchronize() raises it. A line measured on such a run must not be printed.
        eng.synchronize()
        step_event_ms = ev0.elapsed_time(ev1) / args.steps
        # per-launch kernel durations of exactly those steps (the engine brackets each kernel launch)
        per_call = eng.kernel_times(min(64, args.steps * max(1, len(mine))))
        kernel_ms_median = statistics.median(per_call)
        kernel_ms_mean = sum(per_call) / len(per_call)
    note(f"timed region done: {dt:.4f} s")
    clk_stop.set()
    clk_thread.join(timeout=2.0)
    t_end = t0 + dt
    clk_under_load = [c for (tt, c) in clk_samples if t_end - 1.0 <= tt <= t_end]
    sclk_mhz = statistics.median(clk_under_load) if clk_under_load else None
    dt = max_over_ranks(dt)

    # ---- box calibration: a fixed pure-VALU kernel on the same stream right after the timed
    # steps, ~50 ms, so that a slow box shows in the JSON line itself
    calib = None
    try:
        import ctypes as C

        ms_l, ns_i = C.c_float(0), C.c_float(0)
        rc = _lib.lib().rc_calib_valu(dev_index, C.c_void_p(stream.cuda_stream), 48, C.byref(ms_l), C.byref(ns_i))
        if rc == 0:
            calib = {"ms_per_launch": float(ms_l.value), "ns_per_inst": float(ns_i.value)}
    except Exception as ex:  # noqa: BLE001
        calib = {"error": f"{type(ex).__name__}: {ex}"[:200]}

    # ---- strong scaling of the metric's fixed job: the N = 1 job cut N ways
    strong = None
    if world > 1:
        try:
            with torch.cuda.stream(stream):
                xf = x[:, :L_IN]
                n_out_f = eng.output_len(L_IN)
                nwin_f = n_out_f // wout
                plan_f = shard_plan(CHANNELS, nwin_f, world)
                mine_f = [s for s in plan_f if s.rank == rank]
                comp_f = engine_compute(eng, xf)
                bufs_f = {s: torch.empty((s.ch_count, s.win_count * wout), dtype=torch.float32, device=device)
                          for s in mine_f}
                full_f = torch.empty((CHANNELS, n_out_f), dtype=torch.float32, device=device)
                one_f = [s for s in shard_plan(CHANNELS, nwin_f, 1)]

                def timed(fn, k):
                    for _ in range(3):
                        fn()
                    barrier()
                    t = time.perf_counter()
                    for _ in range(k):
                        fn()
                    barrier()
                    return max_over_ranks(time.perf_counter() - t) / k

                ks = max(args.steps, 20)
                t_n = timed(lambda: [comp_f(s, out=bufs_f[s]) for s in mine_f], ks)
                # the same job on ONE GPU, in this run: every rank does the whole job on its own GPU (max over ranks)
                t_1 = timed(lambda: [comp_f(s, out=full_f[s.ch_first:s.ch_first + s.ch_count]) for s in one_f], ks)
                strong = {
                    "scaling": "strong",
                    "workload": f"BASELINE configs[1] at its own size (L={L_IN}/ch), cut over {world} rank(s) by shard_plan",
                    "steps": ks,
                    "ms_per_step": round(t_n * 1e3, 4),
                    "value_Msamples_s": round(float(n_out_f) * CHANNELS / t_n / 1e6, 1),
                    "ms_per_step_1gpu": round(t_1 * 1e3, 4),
                    "efficiency_vs_1gpu": round(t_1 / (world * t_n), 4),
                    "hops_per_rank": sum(s.ch_count * s.win_count for s in mine_f) * eng.params.hops_per_window,
                    "plan": [(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in plan_f],
                    "note": "outputs left sharded in HBM, no collective; wall time between barriers, max over ranks, "
                            "one kernel launch per shard per step (the per-step host cost is inside)",
                }
                del bufs_f, full_f
        except Exception as ex:  # noqa: BLE001
            strong = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        note(f"strong leg: {strong}")

    extras = {}
    concat = None
    if not args.no_extras:
        with torch.cuda.stream(stream):
            # device copy microbench on this very GPU: the measured bandwidth the roofline is also priced on
            n_copy = 1 << 28  # 1 GiB of f32
            a = torch.empty(n_copy, dtype=torch.float32, device=device).normal_()
            b = torch.empty_like(a)
            for _ in range(3):
                b.copy_(a)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(stream)
            for _ in range(10):
                b.copy_(a)
            c1.record(stream)
            stream.synchronize()
            extras["copy_GBs"] = 2.0 * 4.0 * n_copy * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del a, b
        if world == 1:
            # PCIe-inclusive (host buffers in and out): never `value`, reported for SURVEY §8 d1
            xh = x.cpu().numpy()
            t_e = time.perf_counter()
            yh = eng.stretch_host(xh)
            extras["e2e_pcie_Msamples_s"] = yh.size / (time.perf_counter() - t_e) / 1e6
            del xh, yh
            # the other single-GPU BASELINE configs, timed the same way (pre-heated, median of the engine's per-launch
            # event times) so that the driver's own run carries them: C3 (pitch 3) and C5 (8 ch, window 65536, f = 32)
            other = {}
            try:
                with torch.cuda.stream(stream):
                    def timed_config(xx, **kw):
                        e2 = rocoder_amd.Engine(sample_rate=SAMPLE_RATE, channels=xx.shape[0], seed=SEED, device=dev_index, **kw)
                        o2 = torch.empty((xx.shape[0], e2.output_len(xx.shape[1])), dtype=torch.float32, device=device)
                        e2.stretch_tensor(xx, out=o2)
                        th = time.perf_counter()
                        while time.perf_counter() - th < 0.7:
                            for _ in range(4):
                                e2.stretch_tensor(xx, out=o2)
                            stream.synchronize()
                        clk2, stop2 = [], threading.Event()

                        def samp():
                            while not stop2.is_set():
                                try:
                                    clk2.append(float(torch.cuda.clock_rate(dev_index)))
                                except Exception:  # noqa: BLE001
                                    return
                                time.sleep(0.05)

                        t2 = threading.Thread(target=samp, daemon=True)
                        t2.start()
                        for _ in range(40):
                            e2.stretch_tensor(xx, out=o2)
                        stream.synchronize()
                        stop2.set()
                        t2.join(timeout=2.0)
                        ms2 = statistics.median(e2.kernel_times(10))
                        _, hops2, _ = e2.last_kernel_stats()
                        nwin = kw["window_len"]
                        r2 = {"kernel_ms": round(ms2, 4), "hops": int(hops2),
                              "out_Msamples_s": round(o2.numel() / ms2 / 1e3, 1),
                              "frac_hbm_read": round(hops2 * 4.0 * nwin / ms2 / 1e6 / HBM_PEAK_GBS, 4),
                              "sclk_mhz_under_load": round(statistics.median(clk2), 1) if clk2 else None}
                        e2.close()
                        del o2
                        return r2

                    other["C3_pitch3"] = timed_config(x, window_len=WINDOW, factor=FACTOR, pitch_multiple=3)
                    x5 = synth_on_device(torch, device, 8, 5_292_000)
                    other["C5_8ch_window65536_f32"] = timed_config(x5, window_len=65536, factor=32.0)
                    del x5
            except Exception as ex:  # noqa: BLE001
                other["error"] = f"{type(ex).__name__}: {ex}"[:300]
            extras["other_configs"] = other

    res = None
    if rank == 0:
        total_samples = float(n_out) * CHANNELS * args.steps
        value = total_samples / dt / 1e6
        H = WINDOW // 2
        step_len = eng.params.sample_step_len
        read_b, write_b = 4.0 * WINDOW, 4.0 * H / PITCH
        algo_bytes = hops_mine * read_b  # 4N read bytes per hop, one launch of this rank
        sec = kernel_ms_median * 1e-3
        achieved = algo_bytes / sec / 1e9
        traffic, traffic_src = pmc_traffic(kernel_id)
        # LDS bytes per hop: four exchanges of N/2 complex points (8 B each), written once and read once
        lds_w = lds_r = 4 * (WINDOW // 2) * 8.0
        roof = {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": f"N=16384 fused hop kernel ({kernel_id})",
            "kernel_id": kernel_id,
            "kernel_ms": round(kernel_ms_median, 4),
            "kernel_ms_mean": round(kernel_ms_mean, 4),
            "kernel_ms_min": round(min(per_call), 4),
            "kernel_ms_max": round(max(per_call), 4),
            "kernel_launches_timed": len(per_call),
            "step_event_ms": round(step_event_ms, 4),
            "preheat_launches": n_heat,
            "hops_per_s": round(hops_mine / sec, 1),
            "algorithmic_bytes_per_launch": algo_bytes,
            "frac_total_traffic": round(hops_mine * (read_b + write_b) / sec / 1e9 / HBM_PEAK_GBS, 4),
            "frac_compulsory": round(hops_mine * (4.0 * step_len + write_b) / sec / 1e9 / HBM_PEAK_GBS, 4),
            "lds_GBs": round(hops_mine * (lds_w + lds_r) / sec / 1e9, 1),
            "lds_time_frac": round(hops_mine * (lds_w / (LDS_WRITE_PEAK_GBS * 1e9) + lds_r / (LDS_READ_PEAK_GBS * 1e9)) / sec, 4),
            "note": "achieved = algorithmic READ bytes 4*N per hop (SURVEY §8 d4) / median per-launch kernel time; "
                    "frac_total_traffic adds the 4*H/p write bytes, frac_compulsory counts each input sample once "
                    "(4*step + 4*H/p); lds_time_frac = LDS exchange bytes priced at the measured LDS store/load rates "
                    "of MI355X_MICROARCH.md. The kernel is VALU-pipe bound (FFT butterflies + per-bin hash/sincos), "
                    "see DESIGN.md §5",
        }
        pv = pmc_valu(kernel_id)
        if pv:
            # the counter-backed ceiling of this kernel: its VALU instruction stream at full
            # occupancy of the VALU pipe - what the kernel would run at if nothing but VALU issue ever stalled
            roof["valu_busy"] = pv["valu_busy"]
            roof["valu_insts_per_hop"] = round(pv["valu_insts_per_launch"] / max(1, hops_mine), 1)
            roof["frac_at_full_valu_occupancy"] = round(achieved / HBM_PEAK_GBS / max(pv["valu_busy"], 1e-9), 4)
            roof["note"] += (f". CEILING: PMC ({pv['source']}) has the VALU pipes {pv['valu_busy']:.0%} busy for "
                             f"{roof['valu_insts_per_hop']:.0f} VALU instructions per hop; the same instruction stream "
                             f"with the pipes 100 % busy would reach frac {roof['frac_at_full_valu_occupancy']} - the "
                             "0.40 target needs fewer VALU cycles per hop (the frozen phase spec's hash + sincos, the "
                             "butterflies at 1.5 packed FMAs per point and stage), not more overlap")
        roof["counters_box"] = "builder"  # traffic / valu_busy come from the committed PMC summary, not from this box
        if calib and "ns_per_inst" in calib:
            # kernel_ms scaled to the reference box speed: boxes of the pool differ by +-3..5 % on the same binary
            roof["box_calib_ns"] = round(calib["ns_per_inst"], 4)
            roof["box_calib_ms_per_launch"] = round(calib["ms_per_launch"], 4)
            roof["box_calib_ref_ns"] = BOX_CALIB_REF_NS
            roof["kernel_ms_normalised"] = round(kernel_ms_median * BOX_CALIB_REF_NS / calib["ns_per_inst"], 4)
            roof["frac_normalised"] = round(achieved / HBM_PEAK_GBS * calib["ns_per_inst"] / BOX_CALIB_REF_NS, 4)
        elif calib:
            roof["box_calib_error"] = calib.get("error")
        # the shader clock this box held under the load of the hop kernel (median of the SMI samples of the last second
        # of pre-heat + timed steps) and the kernel time scaled to the reference clock
        roof["sclk_mhz_under_load"] = sclk_mhz
        roof["sclk_samples"] = len(clk_under_load)
        if sclk_mhz:
            roof["sclk_ref_mhz"] = SCLK_REF_MHZ
            roof["kernel_ms_at_ref_clock"] = round(kernel_ms_median * sclk_mhz / SCLK_REF_MHZ, 4)
            roof["frac_at_ref_clock"] = round(achieved / HBM_PEAK_GBS * SCLK_REF_MHZ / sclk_mhz, 4)
        if "copy_GBs" in extras:
            roof["measured_copy_GBs"] = round(extras["copy_GBs"], 1)
            roof["frac_measured_peak"] = round(achieved / extras["copy_GBs"], 4)
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
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo, not a measurement)",
            "config": {
                "workload": f"BASELINE configs[1]: stereo 44.1 kHz, window=16384, factor=8, pitch=1, "
                            f"L={length}/ch ({world} x 26460000), inputs resident in HBM; signal = BASELINE.md §3's "
                            "0.5 sin(2 pi 220 (c+1) t) + 0.05 u_c[t] with u_c from torch.rand on the device "
                            "(seed 0xC0DEC0DE), NOT SURVEY d2's splitmix64 stream (the parity tests use that one; "
                            "throughput does not depend on the noise bits)",
                "hops_per_step": hops_mine * world,
                "hops_per_rank": hops_mine,
                "output_samples_per_step": n_out * CHANNELS,
                "x_realtime": round(value * 1e6 / CHANNELS / SAMPLE_RATE, 1),
                "parallelism": (f"{world} rank(s): one job cut into (channel, hop range) shards by shard_plan "
                                f"{[(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in plan]}, "
                                "outputs left sharded in HBM, no data-path collective"),
            },
            "roofline": roof,
        }
        if strong is None and world == 1:  # at N = 1 the main line IS the fixed job
            strong = {"scaling": "strong", "workload": f"BASELINE configs[1] at its own size (L={L_IN}/ch): the main line",
                      "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 4),
                      "value_Msamples_s": round(value, 1), "ms_per_step_1gpu": round(dt / args.steps * 1e3, 4),
                      "efficiency_vs_1gpu": 1.0}
        if strong:
            res["config"]["strong"] = strong
        if "e2e_pcie_Msamples_s" in extras:
            res["config"]["e2e_pcie_Msamples_s"] = round(extras["e2e_pcie_Msamples_s"], 1)
        if "other_configs" in extras:  # not the metric's config: BASELINE configs[2] and configs[4] on this one GPU
            res["config"]["other_configs"] = extras["other_configs"]
        if world == 1 and not args.no_cpu_baseline:
            one, many = cpu_baselines()
            res["cpu_baseline"] = one
            res["config"]["x_cpu"] = round(value / one["value"], 1)
            if many:
                res["cpu_baseline_all_cores"] = many
                res["config"]["x_cpu_all_cores"] = round(value / many["value"], 1)
    # ---- the one collective of the path, AFTER the main line is complete and under a watchdog: a concat that hangs
    # (it is the only code here that a one-GPU box cannot rehearse) must not cost the measurement

    lock = threading.Lock()
    emitted = [False]

    def emit():
        with lock:
            if rank == 0 and not emitted[0]:
                if concat:
                    res["config"]["concat"] = concat
                if c5:
                    res["config"]["c5_sharded"] = c5
                os.write(real_stdout, (json.dumps(res) + "\n").encode())
            emitted[0] = True

    def watchdog():
        # The main line is printed (it is complete), then the process exits NON-ZERO: a collective that hung on
        # GPU-initialised processes must be visible to the launcher, not look like a clean run.
        nonlocal concat
        with lock:
            if rank == 0 and not emitted[0]:
                res["config"]["concat"] = concat or {
                    "error": f"the post-measurement extras (C5 shards, concat) did not finish within "
                             f"{CONCAT_TIMEOUT_S} s; the main line is unaffected"}
                if c5:
                    res["config"]["c5_sharded"] = c5
                os.write(real_stdout, (json.dumps(res) + "\n").encode())
            emitted[0] = True
        os._exit(3)

    timer = threading.Timer(CONCAT_TIMEOUT_S, watchdog)
    timer.daemon = True
    if world > 1:
        timer.start()
    c5 = None
    if not args.no_extras and dist is not None and world > 1:
        # BASELINE configs[4] (C5: 8 channels, window 65536, factor 32, L = 5 292 000 per channel) cut over the
        # ranks by the same shard_plan (8 ranks: one channel per GPU, no halo). Not the metric's config: it rides in
        # config.c5_sharded, measured after the main line is complete.
        try:
            with torch.cuda.stream(stream):
                C5 = dict(window=65536, factor=32.0, channels=8, length=5_292_000)
                eng5 = rocoder_amd.Engine(window_len=C5["window"], factor=C5["factor"], pitch_multiple=1,
                                          sample_rate=SAMPLE_RATE, channels=C5["channels"], seed=SEED, device=dev_index)
                x5 = synth_on_device(torch, device, C5["channels"], C5["length"])
                wout5 = eng5.params.window_out_len
                n_out5 = eng5.output_len(C5["length"])
                nwin5 = n_out5 // wout5
                plan5 = shard_plan(C5["channels"], nwin5, world)
                mine5 = [s for s in plan5 if s.rank == rank]
                comp5 = engine_compute(eng5, x5)
                bufs5 = {s: torch.empty((s.ch_count, s.win_count * wout5), dtype=torch.float32, device=device)
                         for s in mine5}
                for _ in range(3):
                    for s in mine5:
                        comp5(s, out=bufs5[s])
                barrier()
                k5 = max(3, min(10, args.steps))
                t5 = time.perf_counter()
                for _ in range(k5):
                    for s in mine5:
                        comp5(s, out=bufs5[s])
                barrier()
                dt5 = max_over_ranks(time.perf_counter() - t5)
                c5 = {
                    "workload": "BASELINE configs[4]: 8 ch, window=65536, factor=32, L=5292000/ch, one job cut by shard_plan",
                    "steps": k5,
                    "ms_per_step": round(dt5 / k5 * 1e3, 4),
                    "value_Msamples_s": round(float(n_out5) * C5["channels"] * k5 / dt5 / 1e6, 1),
                    "plan": [(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in plan5],
                }
                del bufs5, x5
                eng5.close()
        except Exception as ex:  # noqa: BLE001
            c5 = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        note(f"c5 extra: {c5}")
    if not args.no_extras:
        with torch.cuda.stream(stream):
            if dist is not None and world > 1 and rehearsal:
                concat = {"skipped": "rehearsal: gloo has no device-to-device send/recv"}
                if os.environ.get("ROCODER_BENCH_REHEARSAL") == "hang":
                    concat = None
                    time.sleep(1e6)
            elif dist is not None and world > 1:
                # the one collective of the path: concat of the shards on rank 0, straight into the final
                # layout (stretch_sharded). Timed as compute + concat per step. A failure here must not cost
                # the main line: it is reported inside config.concat instead.
                try:
                    full = torch.empty((CHANNELS, n_out), dtype=torch.float32, device=device) if rank == 0 else None
                    k2 = max(2, min(5, args.steps))
                    stretch_sharded(compute, CHANNELS, nwin, wout, dst=0, full=full)
                    barrier()
                    tc = time.perf_counter()
                    for _ in range(k2):
                        stretch_sharded(compute, CHANNELS, nwin, wout, dst=0, full=full)
                    barrier()
                    dtc = time.perf_counter() - tc
                    dtc = max_over_ranks(dtc)
                    concat = {
                        "steps": k2,
                        "ms_per_step_with_concat": round(dtc / k2 * 1e3, 4),
                        "value_with_concat": round(float(n_out) * CHANNELS * k2 / dtc / 1e6, 1),
                        "bytes_moved_to_rank0": int((n_out * CHANNELS - (my_samples if rank == 0 else 0)) * 4),
                        "how": "grouped RCCL send/recv of each shard into its view of the final [channels, n_out] "
                               "tensor on rank 0 (root-inbound-bound); no pad, no staging copy",
                    }
                    del full
                except Exception as ex:  # noqa: BLE001
                    concat = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    timer.cancel()


if __name__ == "__main__":
    main()
