import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, flo_amd, signals
from oracle import oracle as O
ctx = flo_amd.Context(0)
for sr in (128000, 192000):
    pcm = signals.music_like(sr, 20000, 2, seed=sr)
    o = O.lossy_analyze(pcm, sr, 2, 0.55)
    g = ctx.lossy_analyze(pcm, sr, 2, 0.55)
    d = np.abs(o["sf_words"].astype(int) - g["sf_words"].astype(int))
    print(sr, "max diff per band:", d.max(axis=(0, 1)))
    band = O.psy_tables(sr)[1]
    print("bins per band", np.bincount(band, minlength=25))
