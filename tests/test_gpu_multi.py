"""GPU tests (-m gpu) of the one-process multi-device layer (include/rocoder_hip.h, rc_multi). A one-GPU box lists
device 0 several times: every engine, host thread, input-span copy, range computation and placement copy of the real
path runs (the peer copy degenerates to device-to-device) - what it cannot show is xGMI bandwidth."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def _ra():
    import rocoder_amd
    from rocoder_amd import _lib

    assert _lib.lib().rc_device_count() > 0
    return rocoder_amd


@pytest.mark.parametrize("n_dev", [2, 3, 8])
@pytest.mark.parametrize("N,f,p,ch,L", [(16384, 8.0, 1, 2, 700_000), (16384, 8.0, 3, 2, 300_000),
                                        (16384, 8.0, 1, 3, 400_000), (1024, 2.0, 2, 1, 50_000),
                                        (65536, 32.0, 1, 8, 150_000), (4096, 0.3, 1, 2, 90_000)])
def test_multi_device_equals_one_engine_bit_for_bit(n_dev, N, f, p, ch, L):
    """Host form and device form, every shard plan: identical bits to the same job on one engine (hops are a pure
    function of (seed, channel, hop, bin); each shard recomputes the hop before its range)."""
    import torch

    ra = _ra()
    x = np.stack([onp.synth_input(c, L) for c in range(ch)])
    with ra.Engine(window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=99) as e:
        one = e.stretch_host(x)
    with ra.MultiEngine([0] * n_dev, window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=99) as m:
        got_h = m.stretch_host(x)
        xt = torch.from_numpy(x).cuda()
        got_d = m.stretch_tensor(xt, root=n_dev - 1).cpu().numpy()
        got_d2 = m.stretch_tensor(xt, root=0).cpu().numpy()  # (buffers of the first call are reused)
    assert got_h.shape == one.shape
    assert np.array_equal(got_h, one)
    assert np.array_equal(got_d, one)
    assert np.array_equal(got_d2, one)


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
