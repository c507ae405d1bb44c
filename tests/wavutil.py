"""Tiny WAV writers/readers for the CLI tests (independent of the C++ code under test)."""
import struct

import numpy as np


def write_wav(path, data, rate, fmt):
    """data: float array [channels, frames] in [-1, 1]; fmt in {'u8','i16','i24','i32','f32'}.
    Returns the planar f32 samples the reference's reader yields for this file
    (hound + src/audio.rs:16-29 scaling)."""
    data = np.atleast_2d(np.asarray(data, np.float64))
    ch, n = data.shape
    inter = data.T.reshape(-1)
    if fmt == "f32":
        raw = inter.astype("<f4").tobytes()
        tag, bits = 3, 32
        dec = inter.astype(np.float32)
    elif fmt == "u8":
        q = np.clip(np.round(inter * 127.0), -128, 127).astype(np.int64)
        raw = (q + 128).astype(np.uint8).tobytes()
        tag, bits = 1, 8
        dec = (q.astype(np.float32) / np.float32(127.0)).astype(np.float32)
    elif fmt == "i16":
        q = np.clip(np.round(inter * 32767.0), -32768, 32767).astype(np.int64)
        raw = q.astype("<i2").tobytes()
        tag, bits = 1, 16
        dec = (q.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    elif fmt == "i24":
        q = np.clip(np.round(inter * 8388607.0), -8388608, 8388607).astype(np.int64)
        b = (q & 0xFFFFFF).astype("<u4").tobytes()
        raw = b"".join(b[i:i + 3] for i in range(0, len(b), 4))
        tag, bits = 1, 24
        dec = (q.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)
    elif fmt == "i32":
        q = np.clip(np.round(inter * 2147483647.0), -2147483648, 2147483647).astype(np.int64)
        raw = q.astype("<i4").tobytes()
        tag, bits = 1, 32
        dec = (q.astype(np.float32) / np.float32(2147483647.0)).astype(np.float32)
    else:
        raise ValueError(fmt)
    block = ch * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, tag, ch, rate, rate * block, block, bits) + b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)
    return dec.reshape(n, ch).T.copy()


def read_wav_f32(path):
    """Reads a 32-bit float WAV (plain or WAVE_FORMAT_EXTENSIBLE) -> (rate, [channels, frames])."""
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    ch = rate = None
    while pos + 8 <= len(b):
        cid, ln = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + ln]
        if cid == b"fmt ":
            tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE:
                tag = struct.unpack("<H", body[24:26])[0]
            assert tag == 3 and bits == 32, (tag, bits)
        elif cid == b"data":
            x = np.frombuffer(body, "<f4")
            return rate, x.reshape(-1, ch).T.copy()
        pos += 8 + ln + (ln & 1)
    raise ValueError("no data chunk")
