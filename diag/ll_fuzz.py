"""Randomised differential test of the lossless encoder: device bytes vs oracle bytes on random clips."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, flo_amd
from oracle import oracle as O
ctx = flo_amd.Context(0)
rng = np.random.default_rng(2026)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for it in range(N):
    sr = int(rng.choice([8000, 11025, 22050, 44100, 48000, 96000]))
    ch = int(rng.choice([1, 2, 2, 2, 3]))
    n = int(rng.integers(0, 3 * sr))
    level = int(rng.integers(0, 10))
    kind = int(rng.integers(0, 7))
    t = np.arange(n) / sr
    x = np.zeros((n, ch))
    for c in range(ch):
        if kind == 0:
            x[:, c] = rng.uniform(-1, 1, n) * rng.choice([1e-4, 0.01, 0.3, 1.0])
        elif kind == 1:
            x[:, c] = 0.0
        elif kind == 2:
            for f in rng.uniform(50, sr / 2.2, 4):
                x[:, c] += rng.uniform(0.05, 0.4) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6))
        elif kind == 3:
            x[:, c] = np.clip(2.5 * np.sin(2 * np.pi * 220 * t), -1.2, 1.2)           # clipping, beyond full scale
        elif kind == 4:
            x[:, c] = np.cumsum(rng.normal(0, 0.002, n))                               # random walk
        elif kind == 5:
            x[:, c] = (rng.integers(-3, 4, n)) / 32767.0                               # tiny integers
        else:
            x[:, c] = 0.5 * np.sin(2 * np.pi * 1000 * t) * (t % 0.5 < 0.25) + rng.normal(0, 1e-3, n)
    if ch == 2 and rng.random() < 0.5:
        x[:, 1] = x[:, 0] * 0.97 + rng.normal(0, 1e-4, n)                              # mid/side material
    pcm = x.astype(np.float32).reshape(-1)
    if rng.random() < 0.2 and pcm.size:
        pcm = pcm[: pcm.size - int(rng.integers(0, ch))]                               # trailing partial frame
    g = ctx.encode_lossless(pcm, sr, ch, 16, level)
    o = O.encode_lossless(pcm, sr, ch, 16, level)
    if g != o:
        bad += 1
        print("MISMATCH", it, sr, ch, n, level, kind, len(g), len(o), flush=True)
    gi = ctx.decode_lossless_i32(g)
    oi, _, _ = O.decode_lossless_i32(g)
    if not np.array_equal(gi, oi):
        bad += 1
        print("DECODE MISMATCH", it, sr, ch, n, level, kind, flush=True)
print("lossless fuzz done:", N, "clips, mismatches:", bad)
sys.exit(1 if bad else 0)
