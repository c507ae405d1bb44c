"""numpy-f64 twin of the oracle (TEST INFRASTRUCTURE ONLY).

An independent restatement of the same reference path in numpy, computing the FFTs and the
overlap-add in float64 (pocketfft) while keeping the reference's f32 *tables* (window,
envelope, amp) and the shared phase-source spec. It pins oracle/rocoder_oracle.c (different
FFT implementation, different language) and generates tests/golden/*.npz.

Cites are file:line under /root/reference.
"""
from __future__ import annotations

import numpy as np

PI_F32 = np.float32(np.pi)
MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------- phase source
def _mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def phase_key(seed: int, channel: int, hop: int) -> int:
    ctr = ((channel & 0xFFFFFFFF) << 40 | (hop & 0xFFFFFFFFFF)) & MASK64
    return _mix64(_mix64(seed & MASK64) ^ ctr)


def phase_hash(key: int, bins) -> np.ndarray:
    k0 = np.uint32(key & 0xFFFFFFFF)
    k1 = np.uint32((key >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = np.asarray(bins, dtype=np.uint32) * (k1 | np.uint32(1)) + k0
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x21F0AAAD)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x735A2D97)
        x ^= x >> np.uint32(15)
    return x


def phase_theta(key: int, bins, n_bins: int) -> np.ndarray:
    """theta in [0, pi) as f32 (fft.rs:13,67; rand 0.8.5 UniformFloat). Bins b < n_bins/2 take the
    top 23 bits of hash(b): (h >> 9) * 2^-23 * PI_f32; bins b >= n_bins/2 take the low 16 bits of
    hash(b - n_bins/2): (h & 0xFFFF) * 2^-16 * PI_f32."""
    bins = np.asarray(bins, dtype=np.uint32)
    half = np.uint32(n_bins // 2)
    upper = bins >= half
    h = phase_hash(key, np.where(upper, bins - half, bins))
    u_lo = (h >> np.uint32(9)).astype(np.float32) * np.float32(1.0 / 8388608.0)
    u_up = (h & np.uint32(0xFFFF)).astype(np.float32) * np.float32(1.0 / 65536.0)
    return (np.where(upper, u_up, u_lo).astype(np.float32) * PI_F32).astype(np.float32)


# --------------------------------------------------------------------------- tables
def hanning(n: int) -> np.ndarray:
    """windows.rs:4-9 in f32 operation order."""
    two_pi = np.float32(PI_F32 * np.float32(2.0))
    i = np.arange(n, dtype=np.float32)
    arg = (i * two_pi) / np.float32(n - 1)
    return (np.float32(0.5) - np.cos(arg, dtype=np.float32) * np.float32(0.5)).astype(np.float32)


def hanning_crossfade_compensation(n: int) -> np.ndarray:
    """crossfade.rs:4-10 in f32 operation order."""
    two_pi = np.float32(PI_F32 * np.float32(2.0))
    h = np.float32((np.float32(1.0) + np.sqrt(np.sqrt(np.float32(0.5)))) * np.float32(0.5))
    i = np.arange(n, dtype=np.float32)
    arg = (i * two_pi) / np.float32(n - 1)
    return (np.float32(0.5) - (np.float32(1.0) - h) * np.cos(arg, dtype=np.float32)).astype(np.float32)


def derive(window_len: int, factor: float, amplitude: float, pitch_multiple: int) -> dict:
    """Stretcher::new parameter derivation (stretcher.rs:40-56), f32 arithmetic."""
    assert pitch_multiple != 0
    f = np.float32(factor)
    ap = np.float32(abs(pitch_multiple))
    psf = np.float32(f / ap) if pitch_multiple < 0 else np.float32(f * ap)
    if pitch_multiple < 0:
        S = int(np.ceil(np.float32(window_len) / ap))
    else:
        S = window_len * abs(pitch_multiple)
    amp = np.float32(max(np.float32(4.0), np.float32(psf / np.float32(4.0))) * np.float32(amplitude))
    H = window_len // 2
    step = int(np.float32(window_len) / np.float32(psf * np.float32(2.0)))
    return dict(psf=psf, S=S, amp=amp, H=H, step=step, N=window_len, p=pitch_multiple)


def resample(x: np.ndarray, factor: int) -> np.ndarray:
    """resampler.rs:3-35."""
    if factor == 1:
        return x.copy()
    if factor > 1:
        return x[::factor].copy()
    if factor < -1:
        f = -factor
        a, b = x[:-1, None], x[1:, None]
        r = (np.arange(f, dtype=np.float32) / np.float32(f))[None, :]
        return (a + (b - a) * r).reshape(-1)
    raise ValueError("invalid resample factor")


# --------------------------------------------------------------------------- one hop
def phase_theta_independent23(key: int, bins, n_bins: int) -> np.ndarray:
    """SURVEY §8 c5's literal form, coupled to the frozen spec: EVERY bin gets rand 0.8.5's 23-bit
    draw, and the draws of bins b and b + n/2 share no hash bit. Bin b < n/2 keeps bits 16..31 of
    hash(b) as the top 16 of its 23 bits, bin b + n/2 takes bits 0..15 of hash(b) as ITS top 16; the
    low 7 bits of both come from a second, independently keyed hash. Used only by the test that prices
    the frozen spec's two shortcuts (16-bit upper phases; bits 9..15 of hash(b) seen by both bins):
    |theta_spec - theta_independent23| < pi 2^-16 for every bin, by construction."""
    bins = np.asarray(bins, dtype=np.uint32)
    half = np.uint32(n_bins // 2)
    upper = bins >= half
    h = phase_hash(key, np.where(upper, bins - half, bins))
    fresh = phase_hash(_mix64(key ^ 0xA5A5A5A5DEADBEEF), bins)  # own counter per bin, other key
    lo7 = fresh & np.uint32(0x7F)
    u_lo = (((h >> np.uint32(16)) << np.uint32(7)) | lo7)
    u_up = (((h & np.uint32(0xFFFF)) << np.uint32(7)) | lo7)
    u = np.where(upper, u_up, u_lo).astype(np.float32) * np.float32(1.0 / 8388608.0)
    return (u * PI_F32).astype(np.float32)


def resynth(samples: np.ndarray, window: np.ndarray, key: int, kernel=None, time_ms: int = 0,
            return_spectrum: bool = False, theta_fn=None):
    """ReFFT::resynth (fft.rs:42-74) in f64. theta_fn (tests only) replaces phase_theta."""
    n = window.size
    a = samples[:n].astype(np.float64) * window.astype(np.float64)
    X = np.fft.fft(a)  # unnormalised, e^{-i...}: fft.rs:59
    if kernel is not None:
        try:
            Y = np.asarray(kernel(time_ms, X.astype(np.complex64)), dtype=np.complex64)
            if Y.size == n:
                X = Y.astype(np.complex128)
        except Exception:
            pass  # panic -> noop (fft.rs:100-106)
    theta = (theta_fn or phase_theta)(key, np.arange(n), n).astype(np.float64)
    Z = np.abs(X) * (np.cos(theta) + 1j * np.sin(theta))  # fft.rs:65-68
    y = np.fft.ifft(Z).real  # ifft = unnormalised inverse / N  (fft.rs:69,72)
    out = y * window.astype(np.float64)
    if return_spectrum:
        return out, X
    return out


# --------------------------------------------------------------------------- literal loop
def stretch_channel_literal(x, window_len, factor, amplitude, pitch_multiple, seed, channel,
                            kernel=None, window=None, max_windows=None) -> np.ndarray:
    """Literal transliteration of Stretcher::next_window + the processor loop for ONE channel
    fed as a single chunk then closed (main.rs:148; stretcher_processor.rs:63-70)."""
    d = derive(window_len, factor, amplitude, pitch_multiple)
    N, H, S, step, amp, p = d["N"], d["H"], d["S"], d["step"], float(d["amp"]), d["p"]
    assert step >= 1
    w = hanning(N) if window is None else np.asarray(window, np.float32)
    env = hanning_crossfade_compensation(H).astype(np.float64)
    inp = np.asarray(x, np.float64).copy()
    out_buf = np.zeros(H, np.float64)
    done = False
    hop = 0
    chunks = []
    while not done:
        if max_windows is not None and len(chunks) >= max_windows:
            break
        pos = 0
        while out_buf.size < S + H:
            if inp.size < N:  # ensure_input_samples_available on a closed channel
                inp = np.concatenate([inp, np.zeros(N - inp.size)])
                done = True
            y = resynth(inp[:N], w, phase_key(seed, channel, hop), kernel)
            hop += 1
            out_buf[pos:pos + H] = (y[:H] + out_buf[pos:pos + H]) * env * amp
            out_buf = np.concatenate([out_buf, y[H:]])
            pos += H
            inp = inp[step:]
        chunks.append(resample(out_buf[:S], p))
        out_buf = out_buf[-H:]
    return np.concatenate(chunks) if chunks else np.zeros(0)


# --------------------------------------------------------------------------- closed form (p >= 1)
def hop_count(length: int, window_len: int, step: int, pitch_multiple: int) -> int:
    """K = 2p * ceil((k_d + 1) / 2p), k_d = first hop whose window runs past the input."""
    kd = (length - window_len) // step + 1 if length >= window_len else 0
    hpw = 2 * pitch_multiple
    return hpw * (kd // hpw + 1)


def stretch_channel_closed(x, window_len, factor, amplitude, pitch_multiple, seed, channel,
                           kernel=None) -> np.ndarray:
    """SURVEY §3.2 closed form: O[kH+i] = (y_k[i] + y_{k-1}[H+i]) env[i] amp ; F[t] = O[t p]."""
    assert pitch_multiple >= 1 and window_len % 2 == 0
    d = derive(window_len, factor, amplitude, pitch_multiple)
    N, H, step, amp, p = d["N"], d["H"], d["step"], float(d["amp"]), d["p"]
    w = hanning(N)
    env = hanning_crossfade_compensation(H).astype(np.float64)
    x = np.asarray(x, np.float64)
    K = hop_count(x.size, N, step, p)
    xp = np.concatenate([x, np.zeros(max(0, (K - 1) * step + N - x.size))])
    O = np.zeros(K * H)
    prev_tail = np.zeros(H)
    for k in range(K):
        y = resynth(xp[k * step:k * step + N], w, phase_key(seed, channel, k), kernel)
        O[k * H:(k + 1) * H] = (y[:H] + prev_tail) * env * amp
        prev_tail = y[H:]
    return O[::p].copy()


def synth_input(channel: int, length: int, sample_rate: int = 44100) -> np.ndarray:
    """BASELINE.md §3 synthetic input: 0.5 sin(2 pi 220 (c+1) t) + 0.05 u_c[t],
    u_c ~ uniform(-1,1) from splitmix64 seeded 0xC0DEC0DE + c."""
    t = np.arange(length, dtype=np.float64) / sample_rate
    s = 0.5 * np.sin(2 * np.pi * 220.0 * (channel + 1) * t)
    # vectorised splitmix64
    idx = np.arange(1, length + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(0xC0DEC0DE + channel) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) * 2.0 - 1.0
    return (s + 0.05 * u).astype(np.float32)
