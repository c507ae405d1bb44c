"""CPU tests of the C++ CLI twin's host pieces: duration grammar
(src/duration_parser.rs:32-39), WAV decoding with hound's sample formats and the reference's
int->f32 scaling (src/audio.rs:16-29, src/audio_files.rs:159-188), flag surface (src/main.rs:27-122)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from wavutil import write_wav

CLI = os.path.join(ROOT, "rocoder_amd", "bin", "rocoder")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(CLI):
        from rocoder_amd import build

        build.build()
    assert os.path.exists(CLI)


def run(*args, **kw):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120, **kw)


@pytest.mark.parametrize("s,expected_ms", [  # the reference's 8 test_case lines, verbatim inputs
    ("adkjfn", None), ("1", 1000), ("1:1", 61000), ("1:1:1", 3661000), ("1:1:1.234", 3661234),
    ("1:2:3:4", None), ("1:2.9:4", None), ("1.9:2:4", None)])
def test_parse_duration(s, expected_ms):
    out = run("--parse-duration", s).stdout.strip()
    assert out == ("error" if expected_ms is None else str(expected_ms))


@pytest.mark.parametrize("fmt,ch", [("u8", 1), ("i16", 2), ("i24", 2), ("i32", 1), ("f32", 3)])
def test_wav_reader_formats_and_scaling(tmp_path, fmt, ch):
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, size=(ch, 1000))
    x[:, 0] = -1.0  # exercises the asymmetric int ranges (n / i16::MAX etc.)
    x[:, 1] = 1.0
    wav, raw = str(tmp_path / "in.wav"), str(tmp_path / "out.f32")
    expected = write_wav(wav, x, 22050, fmt)
    r = run("--decode-wav", wav, raw)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == [str(ch), "22050", "1000"]
    got = np.fromfile(raw, np.float32).reshape(ch, 1000)
    assert np.array_equal(got, expected)


def test_wav_from_stdin(tmp_path):
    wav, raw = str(tmp_path / "in.wav"), str(tmp_path / "out.f32")
    expected = write_wav(wav, np.linspace(-0.5, 0.5, 64)[None], 8000, "i16")
    with open(wav, "rb") as f:
        r = subprocess.run([CLI, "--decode-wav", "-", raw], stdin=f, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(raw, np.float32)[None], expected)


def test_flag_surface_matches_reference():
    h = run("--help").stderr
    for flag in ["-w, --window", "-b, --buffer", "-f, --factor", "-p, --pitch_multiple", "-a, --amplitude",
                 "-i, --input", "--rotate-channels", "--freq-kernel", "-x, --fade", "-s, --start",
                 "-d, --duration", "-o, --output"]:
        assert flag in h, flag


def test_playback_and_recording_are_refused_not_faked(tmp_path):
    wav = str(tmp_path / "in.wav")
    write_wav(wav, np.zeros((1, 10)), 8000, "i16")
    assert run("-o", "x.wav").returncode != 0           # no -i: the reference records (cpal)
    assert run("-i", wav).returncode != 0               # no -o: the reference plays (cpal)
