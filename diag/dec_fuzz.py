"""Randomised differential test of the device decoders on damaged files: whatever the oracle decoder makes of a file
(an error, or PCM), the device decoder must make the same (lossless: same integers; lossy: within 2e-6)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "/root/repo/tests")
import numpy as np, flo_amd, flofile, signals
from oracle import oracle as O
ctx = flo_amd.Context(0)
rng = np.random.default_rng(77)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pcm = signals.music_like(44100, 60000, 2, seed=1)
goods = [ctx.encode_lossless(pcm, 44100, 2, 16, 5), ctx.encode_lossless(pcm[:30001], 44100, 1, 16, 8),
         ctx.encode_lossy(pcm, 44100, 2, 0.55), ctx.encode_lossy(pcm, 44100, 2, 1.0)]
bad = errs = 0
for it in range(N):
    g = goods[it % len(goods)]
    f = flofile.parse(g)
    d0 = 70 + f.toc_size
    b = bytearray(g)
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 3))
        at = int(rng.integers(d0, d0 + f.data_size))
        if kind == 0: b[at] = int(rng.integers(0, 256))
        elif kind == 1: b[at:at + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        else: b[at:at + 40] = b"\xff" * min(40, len(b) - at)
    b = bytes(b)
    try:
        want = O.decode(b)[0]
        oerr = None
    except Exception as e:  # noqa: BLE001
        want, oerr = None, str(e)
    try:
        got = ctx.decode(b)
        gerr = None
    except flo_amd.FloError as e:
        got, gerr = None, str(e)
    if (oerr is None) != (gerr is None):
        bad += 1; print("MISMATCH (error vs result)", it, oerr, gerr)
    elif oerr is None:
        if want.shape != got.shape:
            bad += 1; print("MISMATCH (shape)", it, want.shape, got.shape)
        elif want.size:
            with np.errstate(invalid="ignore", over="ignore"):
                fin = np.isfinite(want) & np.isfinite(got)
                # (a damaged scale word can overflow a frame: which of its samples end as inf and which as NaN is a matter of the
                # FFT's order of operations - the oracle's FFT is not the device's - so only WHERE the result is not finite must agree)
                same_nonfinite = np.array_equal(np.isfinite(want), np.isfinite(got))
                diff = np.abs(want[fin].astype(np.float64) - got[fin].astype(np.float64))
            # lossy: 2e-6 absolute on full-scale audio; a damaged scale word can blow a frame up, so relative to the
            # frame's own magnitude (1024-sample-frame blocks)
            if f.is_lossy:
                scale = np.maximum(1.0, np.abs(want[fin]).max() if fin.any() else 1.0)
                tol = 2e-6 * scale
            else:
                tol = 0.0
            worst = float(diff.max()) if diff.size else 0.0
            if worst > tol or not same_nonfinite:
                bad += 1; print("MISMATCH (values)", it, "worst", worst, "max|want|", float(np.abs(want[fin]).max()) if fin.any() else None, "nonfinite same:", same_nonfinite)
    else:
        errs += 1
print("decode fuzz done:", N, "files,", errs, "rejected by both, mismatches:", bad)
