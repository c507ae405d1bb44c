"""GPU tests of the C++ CLI twin (`rocoder -i in.wav -o out.wav ...`): WAV in -> engine -> f32 WAV
out, compared with the oracle run on the samples the reference's reader would decode."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rms
from oracle import cbind as oc
from oracle import oracle_np as onp
from wavutil import read_wav_f32, write_wav

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "rocoder_amd", "bin", "rocoder")
TOL = 1e-4
REG_TOL = 2e-6  # regression gate (tests/test_gpu_parity.py): the kernels deliver < 1e-6 of the signal's RMS


def run(*args, **kw):
    r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300, **kw)
    assert r.returncode == 0, r.stderr
    return r


def check(got, ref):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    for c in range(ref.shape[0]):
        e, r = rms(got[c].astype(np.float64) - ref[c]), rms(ref[c])
        assert e <= TOL and e <= TOL * r + 1e-9, (c, e, r)
        assert e <= REG_TOL * r + 1e-7, ("regression", c, e, r)  # (+1e-7: int16 / int24 sources round on the way in)


@pytest.mark.parametrize("fmt,extra,okw", [
    ("i16", ["-w", "1024", "-f", "4"], dict(window_len=1024, factor=4.0)),
    ("f32", ["-w", "4096", "-f", "8", "-p", "2", "-a", "0.5"],
     dict(window_len=4096, factor=8.0, pitch_multiple=2, amplitude=0.5)),
    ("i24", ["--window", "2048", "--factor", "2", "--pitch_multiple", "-2"],
     dict(window_len=2048, factor=2.0, pitch_multiple=-2)),
])
def test_cli_matches_oracle(tmp_path, fmt, extra, okw):
    x = np.stack([onp.synth_input(c, 30000) for c in range(2)])
    wav, out = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    dec = write_wav(wav, x, 44100, fmt)
    run("-i", wav, "-o", out, "--seed", "77", *extra)
    rate, got = read_wav_f32(out)
    assert rate == 44100
    check(got, oc.stretch_offline(dec, seed=77, **okw))


@pytest.mark.parametrize("extra", [["-w", "16384", "-f", "8"], ["-w", "4096", "-f", "4", "-p", "3"],
                                   ["-w", "3000", "-f", "2"], ["-w", "2048", "-f", "2", "-p", "-2"]])
def test_cli_devices_list_writes_the_same_file(tmp_path, extra):
    """--devices a,b,... (rc_multi_*: the job sharded over the listed GPUs; here the one GPU listed twice / three
    times) writes byte for byte what the Stretcher / StretcherProcessor loop on one engine writes."""
    x = np.stack([onp.synth_input(c, 150000) for c in range(3)])
    wav = str(tmp_path / "in.wav")
    write_wav(wav, x, 44100, "f32")
    outs = []
    for name, dev in (("one", []), ("two", ["--devices", "0,0"]), ("three", ["--devices", "0,0,0"])):
        out = str(tmp_path / f"{name}.wav")
        run("-i", wav, "-o", out, "--seed", "5", *extra, *dev)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]
    assert len(outs[0]) > 44 + 3 * 4 * 150000


def test_cli_streams_the_output_file_with_bounded_memory(tmp_path):
    """README.md:74 / SURVEY f1: the output is written as the windows arrive (header first, sizes patched at the end),
    byte for byte the file of the collect-then-write order (src/main.rs:197-203, ROCODER_CLI_COLLECT=1), and the
    process no longer holds the output: 300 s of mono at factor 8 is 423 MB of samples."""
    x = onp.synth_input(0, 44100 * 300)[None]
    wav = str(tmp_path / "in.wav")
    write_wav(wav, x, 44100, "f32")
    files, peak = {}, {}
    for name, env in (("stream", {}), ("collect", {"ROCODER_CLI_COLLECT": "1"})):
        out = str(tmp_path / f"{name}.wav")
        # the CLI reports the high-water mark of its own image (VmHWM): ru_maxrss of wait4 would also count this test
        # process's pages from between fork and exec
        r = subprocess.run([CLI, "-i", wav, "-o", out, "-w", "16384", "-f", "8", "--seed", "3"],
                           env=dict(os.environ, ROCODER_CLI_TIMING="1", **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        hwm = [ln for ln in r.stderr.splitlines() if "peak_rss_kib" in ln]
        assert hwm, r.stderr
        peak[name] = int(hwm[-1].split()[-1]) * 1024
        files[name] = out
    a, b = open(files["stream"], "rb").read(), open(files["collect"], "rb").read()
    assert len(a) > 400e6 and a == b
    del a, b
    assert peak["collect"] - peak["stream"] > 300e6, peak


def test_cli_devices_refuses_host_kernels(tmp_path):
    x = np.stack([onp.synth_input(c, 20000) for c in range(2)])
    wav, out, src = str(tmp_path / "in.wav"), str(tmp_path / "o.wav"), str(tmp_path / "k.c")
    write_wav(wav, x, 44100, "f32")
    open(src, "w").write("#include <stddef.h>\n#include <stdint.h>\nint apply(uint64_t t, const float *in, float *out, size_t n, void *u) {\n"
                         "    for (size_t i = 0; i < 2 * n; ++i) out[i] = in[i];\n    return 0;\n}\n")
    r = subprocess.run([CLI, "-i", wav, "-o", out, "-w", "1024", "--freq-kernel", src, "--devices", "0,0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rocoder_hip" in r.stderr


def test_cli_default_window_start_duration_rotate(tmp_path):
    x = np.stack([onp.synth_input(c, 44100 * 3) for c in range(2)])
    wav, out = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    dec = write_wav(wav, x, 44100, "i16")
    run("-i", wav, "-o", out, "-f", "2", "-s", "0:0:0.5", "-d", "2", "--rotate-channels", "--seed", "3")
    _, got = read_wav_f32(out)
    clip = dec[:, 22050:22050 + 88200][::-1]  # clip (audio.rs:57-63) then rotate_right(1) (:73-75)
    check(got, oc.stretch_offline(clip, 16384, 2.0, 1.0, 1, seed=3))


def test_cli_stdin_input(tmp_path):
    x = onp.synth_input(0, 20000)[None]
    wav, out = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    dec = write_wav(wav, x, 48000, "i32")
    with open(wav, "rb") as f:
        r = subprocess.run([CLI, "-i", "-", "-o", out, "-w", "512", "-f", "3", "--seed", "1"], stdin=f,
                           capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rate, got = read_wav_f32(out)
    assert rate == 48000
    check(got, oc.stretch_offline(dec, 512, 3.0, 1.0, 1, seed=1, sample_rate=48000))


KERNEL_C = r"""
#include <stddef.h>
#include <stdint.h>
/* the README's example kernel (README.md:121-128) in its C-ABI form: multiply every bin by 2 */
int apply(uint64_t time_ms, const float *in, float *out, size_t n, void *user) {
    (void)time_ms; (void)user;
    for (size_t i = 0; i < 2 * n; ++i) out[i] = in[i] * 2.0f;
    return 0;
}
"""


def test_cli_freq_kernel_compiled_from_source(tmp_path):
    x = np.stack([onp.synth_input(c, 20000) for c in range(2)])
    wav, out, ksrc = str(tmp_path / "in.wav"), str(tmp_path / "out.wav"), str(tmp_path / "kernel.c")
    open(ksrc, "w").write(KERNEL_C)
    dec = write_wav(wav, x, 44100, "f32")
    r = run("-i", wav, "-o", out, "-w", "2048", "-f", "4", "--freq-kernel", ksrc, "--seed", "9")
    assert "Got new kernel" in r.stderr  # src/fft.rs:79
    _, got = read_wav_f32(out)
    ref = oc.stretch_offline(dec, 2048, 4.0, 1.0, 1, seed=9, kernel=lambda t, s: s * np.float32(2.0))
    check(got, ref)


def test_cli_broken_kernel_source_falls_back_to_noop(tmp_path):
    x = onp.synth_input(0, 10000)[None]
    wav, out, ksrc = str(tmp_path / "in.wav"), str(tmp_path / "out.wav"), str(tmp_path / "kernel.c")
    open(ksrc, "w").write("this is not C")
    dec = write_wav(wav, x, 44100, "f32")
    r = run("-i", wav, "-o", out, "-w", "1024", "-f", "2", "--freq-kernel", ksrc, "--seed", "9")
    assert "Failed to compile library" in r.stderr  # src/hotswapper.rs:39
    _, got = read_wav_f32(out)
    check(got, oc.stretch_offline(dec, 1024, 2.0, 1.0, 1, seed=9))


SLOW_GAIN_C = r"""
#include <stddef.h>
#include <stdint.h>
#include <unistd.h>
int apply(uint64_t time_ms, const float *in, float *out, size_t n, void *user) {
    (void)time_ms; (void)user;
    usleep(%d);
    for (size_t i = 0; i < 2 * n; ++i) out[i] = in[i] * %sf;
    return 0;
}
"""


def test_cli_kernel_source_is_reloaded_while_running(tmp_path):
    """src/hotswapper.rs:19-30 + src/fft.rs:78-85: the kernel source is polled every 100 ms; an edit made
    while the stretch runs is compiled and takes over for the hops that follow ("Got new kernel" a second
    time, output gain changes from 1 to 3 somewhere inside the file)."""
    import time

    x = onp.synth_input(0, 60000)[None]
    wav, out, ksrc = str(tmp_path / "in.wav"), str(tmp_path / "out.wav"), str(tmp_path / "kernel.c")
    dec = write_wav(wav, x, 44100, "f32")
    open(ksrc, "w").write(SLOW_GAIN_C % (3000, "1.0"))  # ~470 hops x 3 ms: the run lasts > 1 s
    p = subprocess.Popen([CLI, "-i", wav, "-o", out, "-w", "1024", "-f", "4", "--freq-kernel", ksrc, "--seed", "9"],
                         stderr=subprocess.PIPE, text=True)
    time.sleep(0.5)
    open(ksrc, "w").write(SLOW_GAIN_C % (0, "3.0"))
    _, err = p.communicate(timeout=120)
    assert p.returncode == 0, err
    assert err.count("Got new kernel") == 2, err
    _, got = read_wav_f32(out)
    ref = oc.stretch_offline(dec, 1024, 4.0, 1.0, 1, seed=9)  # gain 1
    assert got.shape == ref.shape
    n = ref.shape[1]
    head, tail = slice(0, n // 20), slice(n - n // 10, n)
    g_head = rms(got[0, head]) / rms(ref[0, head])
    g_tail = rms(got[0, tail]) / rms(ref[0, tail])
    assert abs(g_head - 1.0) < 1e-3 and abs(g_tail - 3.0) < 3e-3, (g_head, g_tail)
