import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, flo_amd, signals
ctx = flo_amd.Context(0)
lens = [0, 1, 1023, 1024, 5000, 44100, 70001, 3 * 1024]
for q in (0.0, 0.55, 1.0):
    clips = [signals.music_like(44100, n, 2, seed=40 + i) for i, n in enumerate(lens)]
    ctx.force_path(3); a = ctx.encode_batch(1, clips, 44100, 2, q)
    ctx.force_path(4); b = ctx.encode_batch(1, clips, 44100, 2, q)
    print("q", q, "identical" if a == b else "DIFFERENT", [len(x) for x in a][:4])
ctx.force_path(0)
for n in (256, 1024, 1250, 2500):
    bt = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [441000 * 2] * n, 44100, 2, 0.55)
    bt.fill_synthetic()
    for path in (3, 4):
        for _ in range(2): bt.encode(path); bt.sync()
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(5): bt.encode(path); bt.sync()
        ctx.profile_enable(False)
        ms, k = ctx.profile_query("lossy_chain3" if path == 3 else "lossy_chain2x")
        print(n, "path", path, round(ms / max(k, 1), 4), "ms")
    bt.close()
