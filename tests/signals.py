"""Deterministic test signals (same generators the reference's tests use)."""
import numpy as np


def sine(freq, sr, n, amp=0.5, channels=1, phase=0.0):
    t = np.arange(n, dtype=np.float32) / np.float32(sr)
    s = (np.sin(np.float32(2.0 * np.pi) * np.float32(freq) * t + np.float32(phase)) * np.float32(amp)).astype(np.float32)
    if channels == 1:
        return s
    return np.repeat(s[:, None], channels, axis=1).reshape(-1).astype(np.float32)


def lcg_noise(n, seed=12345, amp=0.3):
    """libflo/tests/rust/lossy_transform_tests.rs:89-96 — state*1103515245+12345, (state>>16)/32768-1."""
    out = np.empty(n, dtype=np.float32)
    state = np.uint32(seed)
    a, c = np.uint32(1103515245), np.uint32(12345)
    with np.errstate(over="ignore"):
        for i in range(n):
            state = state * a + c
            out[i] = np.float32(np.float32(state >> np.uint32(16)) / np.float32(32768.0) - np.float32(1.0))
    return (out * np.float32(amp)).astype(np.float32)


def fast_noise(n, seed=1, amp=0.3):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-1, 1, n) * amp).astype(np.float32)


def music_like(sr, n, channels=2, seed=0):
    """A few partials + decaying noise bursts, different per channel; exercises tonal + transient paths."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    out = np.zeros((n, channels), dtype=np.float64)
    for c in range(channels):
        for f in rng.uniform(80, 6000, 5):
            out[:, c] += rng.uniform(0.02, 0.2) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
        env = np.exp(-((t * 4) % 1.0) * 6.0)
        out[:, c] += 0.1 * env * rng.standard_normal(n)
    if channels == 2:
        out[:, 1] = 0.6 * out[:, 1] + 0.4 * out[:, 0]
    return np.clip(out, -1, 1).astype(np.float32).reshape(-1)


def quantize16(x):
    """Snap to the 16-bit grid the lossless codec is exact on (value = i/32767 style inputs)."""
    i = np.trunc(np.clip(x * np.float32(32767.0), -32768, 32767))
    return i.astype(np.int32)
