"""One iteration of diag/dec_fuzz.py (same generator): where do the device and the oracle decoder differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, flo_amd, flofile, signals
from oracle import oracle as O
ctx = flo_amd.Context(0)
rng = np.random.default_rng(77)
target = int(sys.argv[1])
pcm = signals.music_like(44100, 60000, 2, seed=1)
goods = [ctx.encode_lossless(pcm, 44100, 2, 16, 5), ctx.encode_lossless(pcm[:30001], 44100, 1, 16, 8),
         ctx.encode_lossy(pcm, 44100, 2, 0.55), ctx.encode_lossy(pcm, 44100, 2, 1.0)]
for it in range(target + 1):
    g = goods[it % len(goods)]
    f = flofile.parse(g)
    d0 = 70 + f.toc_size
    b = bytearray(g)
    dmg = []
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 3))
        at = int(rng.integers(d0, d0 + f.data_size))
        dmg.append((kind, at - d0))
        if kind == 0: b[at] = int(rng.integers(0, 256))
        elif kind == 1: b[at:at + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        else: b[at:at + 40] = b"\xff" * min(40, len(b) - at)
b = bytes(b)
want = O.decode(b)[0]; got = ctx.decode(b)
print("damage (kind, offset in DATA):", dmg, "frames", len(f.frames) if hasattr(f, "frames") else "?")
bad = np.nonzero(~(np.isfinite(want) == np.isfinite(got)) | (np.isnan(want) != np.isnan(got)))[0]
print("positions where finiteness differs:", bad.size, bad[:10], "blocks", np.unique(bad // 2048)[:10])
for i in bad[:6]: print(i, "oracle", want[i], "device", got[i])
