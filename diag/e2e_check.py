import sys, time, numpy as np
sys.path.insert(0, '.')
import flo_amd
ctx = flo_amd.Context(0)
sr, ch, n_il = 44100, 2, 441000 * 2
bg = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il] * 64, sr, ch, 0.55)
bg.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
a = [bg.download_pcm(i) for i in range(64)]
b = [x.copy() for x in a]
bg.close()
for name, clips in (("download", a), ("copy", b), ("download", a), ("copy", b)):
    ctx.encode_batch(flo_amd.MODE_LOSSY, clips, sr, ch, 0.55)
    best = 1e9
    for _ in range(4):
        t = time.perf_counter(); ctx.encode_batch(flo_amd.MODE_LOSSY, clips, sr, ch, 0.55); best = min(best, time.perf_counter() - t)
    t = time.perf_counter()
    for i in range(20): ctx.encode_lossy(clips[i], sr, ch, 0.55)
    d6 = (time.perf_counter() - t) / 20
    print(name, "batch64 ms", round(best * 1e3, 2), "one clip ms", round(d6 * 1e3, 3))
