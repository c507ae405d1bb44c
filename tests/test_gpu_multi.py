"""GPU tests (-m gpu) of the one-process multi-device layer (include/rocoder_hip.h, rc_multi). A one-GPU box lists
device 0 several times: every engine, host thread, input-span copy, range computation and placement copy of the real
path runs (the peer copy degenerates to device-to-device) - what it cannot show is xGMI bandwidth."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from multi_devices import device_lists, rccl_world
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _ra():
    import rocoder_amd
    from rocoder_amd import _lib

    assert _lib.lib().rc_device_count() > 0
    return rocoder_amd


def _n_have():
    from rocoder_amd import _lib

    return int(_lib.lib().rc_device_count())


@pytest.mark.parametrize("n_dev", [2, 3, 8])
@pytest.mark.parametrize("N,f,p,ch,L", [(16384, 8.0, 1, 2, 700_000), (16384, 8.0, 3, 2, 300_000),
                                        (16384, 8.0, 1, 3, 400_000), (1024, 2.0, 2, 1, 50_000),
                                        (65536, 32.0, 1, 8, 150_000), (4096, 0.3, 1, 2, 90_000),
                                        (65536, 32.0, 1, 2, 1_500_000)])  # (long enough for big5's run seams, cut differently per plan)
def test_multi_device_equals_one_engine_bit_for_bit(n_dev, N, f, p, ch, L):
    """Host form and device form, every shard plan: identical bits to the same job on one engine (hops are a pure
    function of (seed, channel, hop, bin); each shard recomputes the hop before its range)."""
    import torch

    ra = _ra()
    x = np.stack([onp.synth_input(c, L) for c in range(ch)])
    with ra.Engine(window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=99) as e:
        one = e.stretch_host(x)
    # [0] * n_dev always; on a box with several GPUs also the list spread over DISTINCT devices (real peer copies)
    for ids in device_lists(n_dev, _n_have()):
        with ra.MultiEngine(ids, window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=99) as m:
            got_h = m.stretch_host(x)
            # the tensors live on the root of the list: both ends of it are tried
            x_last, x_first = (torch.from_numpy(x).to(f"cuda:{ids[r]}") for r in (n_dev - 1, 0))
            # a share on the root's own device computes in place...
            got_ip = m.stretch_tensor(x_last, root=n_dev - 1).cpu().numpy()
            # ... and with forced staging it takes a remote device's path: span copy in, compute, shard copy out
            m.set_staging(True)
            got_d = m.stretch_tensor(x_last, root=n_dev - 1).cpu().numpy()
            got_d2 = m.stretch_tensor(x_first, root=0).cpu().numpy()  # (buffers of the first call are reused)
        assert got_h.shape == one.shape, ids
        assert np.array_equal(got_h, one), ids
        assert np.array_equal(got_ip, one), ids
        assert np.array_equal(got_d, one), ids
        assert np.array_equal(got_d2, one), ids


@pytest.mark.parametrize("L,n_dev", [(0, 2), (700, 3), (1024, 8), (1500, 4)])
def test_multi_device_edge_lengths(L, n_dev):
    """Empty input, input shorter than a window, more devices than windows: the same bits as one engine, host and
    device form, in place and staged."""
    import torch

    ra = _ra()
    x = np.stack([onp.synth_input(c, L) for c in range(2)]) if L else np.zeros((2, 0), np.float32)
    with ra.Engine(window_len=1024, factor=2.0, channels=2, seed=5) as e:
        one = e.stretch_host(x)
    for ids in device_lists(n_dev, _n_have()):
        with ra.MultiEngine(ids, window_len=1024, factor=2.0, channels=2, seed=5) as m:
            assert np.array_equal(m.stretch_host(x), one), ids
            x_last, x_first = ((torch.from_numpy(x) if L else torch.zeros((2, 0))).to(f"cuda:{ids[r]}") for r in (n_dev - 1, 0))
            assert np.array_equal(m.stretch_tensor(x_last, root=n_dev - 1).cpu().numpy(), one), ids
            m.set_staging(True)
            assert np.array_equal(m.stretch_tensor(x_first, root=0).cpu().numpy(), one), ids


def test_multi_device_refuses_a_host_kernel_and_a_bad_root():
    ra = _ra()
    from rocoder_amd import _lib

    with pytest.raises(_lib.RocoderError) as ei:
        ra.MultiEngine([0, 0], window_len=1024, factor=2.0, channels=1, kernel=lambda t, s: s)
    assert ei.value.code == _lib.RC_EUNSUPPORTED
    import torch

    with ra.MultiEngine([0, 0], window_len=1024, factor=2.0, channels=1) as m:
        with pytest.raises(_lib.RocoderError) as ei:
            m.stretch_tensor(torch.zeros((1, 5000), device="cuda"), root=2)
        assert ei.value.code == _lib.RC_EINVAL


def test_multi_device_checks_where_the_tensors_live_and_keeps_the_callers_device():
    """pointers that are not device memory of the list's root are RC_EINVAL (not garbage or a fault), and
    the calling thread's current HIP device is what it was."""
    import ctypes as C

    import torch

    ra = _ra()
    from rocoder_amd import _lib

    L = _lib.lib()
    with ra.MultiEngine([0, 0], window_len=1024, factor=2.0, channels=2) as m:
        x = torch.zeros((2, 5000), device="cuda")
        n_out = m.output_len(5000)
        out = torch.empty((2, n_out), device="cuda")
        host = np.zeros((2, 5000), np.float32)
        got = C.c_size_t(0)
        rc = L.rc_multi_stretch_device(m._h, 0, C.c_void_p(host.ctypes.data), 5000, 5000, C.c_void_p(out.data_ptr()),
                                       n_out, n_out, C.byref(got), None)
        assert rc == _lib.RC_EINVAL and b"d_in" in L.rc_last_error()
        rc = L.rc_multi_stretch_device(m._h, 0, C.c_void_p(x.data_ptr()), 5000, 5000, C.c_void_p(host.ctypes.data),
                                       n_out, n_out, C.byref(got), None)
        assert rc == _lib.RC_EINVAL and b"d_out" in L.rc_last_error()
        rc = L.rc_multi_stretch_device(m._h, 0, C.c_void_p(x.data_ptr()), 100, 5000, C.c_void_p(out.data_ptr()),
                                       n_out, n_out, C.byref(got), None)  # rows overlap
        assert rc == _lib.RC_EINVAL
        with pytest.raises(AssertionError):
            m.stretch_tensor(torch.zeros((3, 5000), device="cuda"))  # channel count
        with pytest.raises(AssertionError):
            m.stretch_tensor(x, out=torch.empty((2, n_out - 1), device="cuda"))
        before = torch.cuda.current_device()
        m.stretch_tensor(x, out=out)
        assert torch.cuda.current_device() == before


def test_multi_device_from_plain_c(tmp_path):
    """The C program a host binding would look like (tests/c/multi_driver.c), compiled here against the header and
    the library: a stereo job on a list of two and of four devices equals the one-engine job bit for bit."""
    exe = tmp_path / "multi_driver"
    lib_dir = os.path.join(ROOT, "rocoder_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "c", "multi_driver.c"), "-L", lib_dir, "-lrocoder_hip", "-lm",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    for n_dev, N, f, p, L in ((2, 16384, 8.0, 1, 500000), (4, 16384, 8.0, 3, 200000), (3, 65536, 32.0, 1, 200000)):
        r = subprocess.run([str(exe), str(n_dev), str(N), str(f), str(p), str(L)], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        assert r.stdout.startswith("OK"), r.stdout


@pytest.mark.gpu
def test_rccl_one_rank_walks_the_sharded_concat():
    """the RCCL branch of rocoder_amd.distributed (communicator creation, broadcast
    into shard views, the grouped send / recv launch) executes on hardware, bit-exact against the single-engine tensor.
    One rank PER GPU of the box (at most 8), each started as a fresh child process on its own device (RCCL refuses two
    ranks on one device): with one GPU it is the one-rank walk of round 5, on a node it is a real concat over xGMI."""
    import json
    import socket
    import sys

    world = rccl_world(_n_have())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py")],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:  # exactly the children started above
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, se[-3000:])
    res = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert res["backend"] == "nccl" and res["world"] == world and res["all_reduce"] == 3.5 + (world - 1)
    for k in ("broadcast_all", "root_only", "grouped_send_recv_to_self"):
        assert res[k] is True and res[k + "_into_caller_buffer"] is True, res


@pytest.mark.gpu
def test_bench_rehearsal_survives_a_leg_that_fails_on_one_rank():
    """bench.py --gpus 2 without a launcher (it starts its own ranks), both ranks on this one GPU over gloo
    (ROCODER_BENCH_REHEARSAL=1), with the set-up of the `weak` leg failing on rank 1 only: the ranks agree to skip that
    leg's timed collectives (a rank that skipped its barriers alone hung the others), the main line is
    printed with the error inside config.weak, the later legs still run, exit code 0."""
    import json
    import sys

    env = dict(os.environ, ROCODER_BENCH_REHEARSAL="1", ROCODER_BENCH_REHEARSAL_FAIL="weak:1",
               ROCODER_BENCH_CONCAT_TIMEOUT="150")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--preheat-s", "0.3"], env=env, capture_output=True, text=True, timeout=420)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    cfg = res["config"]
    assert "error" in cfg["weak"], cfg["weak"]
    assert "ms_per_step_1gpu" in cfg["ref_1gpu"] and "ms_per_step" in cfg["c5_sharded"], cfg
    assert "legs_error" not in cfg
