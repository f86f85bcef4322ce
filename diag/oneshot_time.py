import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, flo_amd
from oracle import oracle as O
ctx = flo_amd.Context(0)
pcm = O.synth_clip(441000, 2, clip_id=3)
for name, fn in (("lossy", lambda: ctx.encode_lossy(pcm, 44100, 2, 0.55)), ("lossless", lambda: ctx.encode_lossless(pcm, 44100, 2, 16, 5))):
    fn(); fn()
    t = time.perf_counter()
    for _ in range(20): out = fn()
    dt = (time.perf_counter() - t) / 20
    print(name, "one-shot 10 s stereo:", round(dt * 1e3, 3), "ms ->", round(pcm.size / dt / 1e6, 1), "Msamples/s,", len(out), "bytes")
clips = [O.synth_clip(441000, 2, clip_id=i) for i in range(64)]
ctx.encode_batch(1, clips, 44100, 2, 0.55)
t = time.perf_counter(); outs = ctx.encode_batch(1, clips, 44100, 2, 0.55); dt = time.perf_counter() - t
print("encode_batch 64 x 10 s from host buffers:", round(dt * 1e3, 2), "ms ->", round(64 * pcm.size / dt / 1e6, 1), "Msamples/s")
