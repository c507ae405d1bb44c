"""CPU tests of the C++ CLI twin's host pieces: duration grammar
(src/duration_parser.rs:32-39), WAV decoding with hound's sample formats and the reference's
int->f32 scaling (src/audio.rs:16-29, src/audio_files.rs:159-188), flag surface (src/main.rs:27-122)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from wavutil import write_wav

# ROCODER_CLI: run the same tests on another build of the CLI (tools/run_sanitizers.sh: the ASan + UBSan binary)
CLI = os.environ.get("ROCODER_CLI") or os.path.join(ROOT, "rocoder_amd", "bin", "rocoder")


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


def _wav_corpus():
    """Hostile inputs for the WAV reader (it parses untrusted files): every truncation of a valid header, byte
    flips in the header, absurd chunk lengths, zero / huge channel counts, odd bit depths, chunks in the wrong order."""
    import io
    import struct

    rng = np.random.default_rng(11)

    def wav(fmt_tag=1, ch=2, rate=44100, bits=16, data=b"\x01\x02" * 64, fmt_len=16, data_len=None, extra=b"",
            riff=b"RIFF", wave=b"WAVE", order=("fmt", "data")):
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, rate, (rate * ch * max(bits, 8) // 8) & 0xFFFFFFFF,
                          (ch * max(bits, 8) // 8) & 0xFFFF, bits)
        fmt = (fmt + b"\0" * 64)[:max(fmt_len, 0)] if fmt_len != 16 else fmt
        chunks = {"fmt": b"fmt " + struct.pack("<I", fmt_len & 0xFFFFFFFF) + fmt,
                  "data": b"data" + struct.pack("<I", (len(data) if data_len is None else data_len) & 0xFFFFFFFF) + data}
        body = wave + extra + b"".join(chunks[k] for k in order)
        return riff + struct.pack("<I", len(body) & 0xFFFFFFFF) + body

    good = wav()
    corpus = [good[:n] for n in range(0, len(good), 1)][:80]          # every truncation through the header + some data
    for _ in range(120):                                              # byte flips inside the first 48 bytes
        b = bytearray(good)
        for _k in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, 48))] = int(rng.integers(0, 256))
        corpus.append(bytes(b))
    corpus += [
        wav(fmt_len=0xFFFFFFFF), wav(fmt_len=0x7FFFFFF0), wav(fmt_len=15), wav(fmt_len=17), wav(fmt_len=40, fmt_tag=0xFFFE),
        wav(data_len=0xFFFFFFFE), wav(data_len=0x80000000), wav(data_len=0), wav(data_len=0xFFFFFFFF), wav(data_len=3),
        wav(ch=0), wav(ch=65535), wav(bits=0), wav(bits=7), wav(bits=12), wav(bits=64), wav(bits=24, data=b"\xff" * 7),
        wav(fmt_tag=3, bits=16), wav(fmt_tag=3, bits=32, data=b"\x00\x00\xc0\x7f" * 8), wav(fmt_tag=2), wav(rate=0),
        wav(order=("data", "fmt")), wav(order=("data",)), wav(order=("fmt",)), wav(extra=b"LIST" + struct.pack("<I", 0xFFFFFFFF)),
        wav(extra=b"junk" + struct.pack("<I", 5) + b"12345\0"), wav(riff=b"RIFX"), wav(wave=b"WAVF"), b"", b"RIFF", good * 2,
    ]
    return corpus


def test_wav_reader_survives_truncated_and_garbage_files(tmp_path):
    """No input may crash the reader (a signal) or make it allocate what the file merely claims: it either decodes
    or fails with a message and a non-zero status - in this build and, through tools/run_sanitizers.sh, under
    AddressSanitizer + UndefinedBehaviorSanitizer."""
    corpus = _wav_corpus()
    assert len(corpus) > 200
    bad = []
    for k, blob in enumerate(corpus):
        wav, raw = str(tmp_path / "f.wav"), str(tmp_path / "f.f32")
        open(wav, "wb").write(blob)
        r = run("--decode-wav", wav, raw)
        if r.returncode < 0 or r.returncode > 1 or "Sanitizer" in r.stderr or "runtime error" in r.stderr:
            bad.append((k, r.returncode, r.stderr[-300:]))
        elif r.returncode == 0:  # decoded: the counts it printed are what it wrote
            ch, _rate, frames = (int(v) for v in r.stdout.split())
            assert os.path.getsize(raw) == 4 * ch * frames, (k, r.stdout)
    assert not bad, bad[:5]


def test_stub_engine_defines_every_engine_symbol_the_cli_imports(tmp_path):
    """The TSan target links the CLI against tests/c/stub_engine.c (tools/run_sanitizers.sh): a new engine call in the
    CLI without its stub breaks that build. Compile both here (no sanitizer, seconds) and compare symbol tables."""
    obj, so = str(tmp_path / "cli.o"), str(tmp_path / "stub.so")
    src = os.path.join(ROOT, "rocoder_amd", "csrc", "host", "rocoder_cli.cpp")
    subprocess.run(["g++", "-std=c++17", "-O0", "-c", src, "-o", obj], check=True, timeout=300)
    subprocess.run(["gcc", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "stub_engine.c"), "-o", so],
                   check=True, timeout=120)
    und = {l.split()[-1] for l in subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout.splitlines()
           if l.split()[-1].startswith("rc_")}
    have = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout.splitlines()}
    assert und and und <= have, sorted(und - have)
