import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, flo_amd, signals
ctx = flo_amd.Context(0)
sr, ch = 44100, 2
rng = np.random.default_rng(5)
bad = 0
for amp in (1.0, 30.0, 3000.0, 1e6, 1e12):
    clips = []
    for i in range(300):
        x = signals.music_like(sr, 20000 + 37 * i, ch, seed=i) * amp
        if i % 3 == 0: x[1::2] *= 1e-3        # very different channel levels
        if i % 5 == 0: x[::2] = 0.0           # one silent channel
        clips.append(x.astype(np.float32))
    outs = {}
    for form in (4, 1, 2, 3):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [c.size for c in clips], sr, ch, 0.55)
        for i, c in enumerate(clips): b.upload(i, c)
        b.encode(form); b.sync()
        outs[form] = [b.fetch(i) for i in range(0, 300, 7)]
        b.close()
    for form in (1, 2, 3):
        same = outs[form] == outs[4]
        if not same: bad += 1
        print("amp", amp, "form", form, "== form 4:", same, flush=True)
print("mismatching forms:", bad)
