"""Soak: the lock-step stereo chain (form 5) and the one-wave-per-channel chain (form 1) against the frame-parallel form, many
times, several batch shapes and qualities (q = 0.2 / 0.55 / 0.8 / 1.0: sparse frames, item form with and without the block form behind it, dense frames)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flo_amd
ctx = flo_amd.Context(0)
sr, ch = 44100, 2
def packed(b, form):
    b.encode(form); b.sync()
    buf = torch.zeros(b.data_bytes() + 16 * b.n_clips + 64, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.pack_streams(buf.data_ptr(), buf.numel()); b.sync()
    return buf[: offs[-1]].clone()
bad = 0
for shape in ([sr * 10 * ch] * 1250, [sr * 3 * ch] * 2500, [(1000 + 977 * i) * ch for i in range(700)], [sr * 30 * ch] * 300):
    for q in (0.2, 0.55, 0.8, 1.0):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, shape, sr, ch, q)
        b.fill_synthetic(seed=1234, clip_id0=7)
        ref = packed(b, 2)
        for it in range(10):
            for form in (5, 5, 1):
                got = packed(b, form)
                if got.shape != ref.shape or not torch.equal(got, ref):
                    bad += 1
                    print("MISMATCH", len(shape), q, it, form)
        b.close()
        print("shape", len(shape), "q", q, "ok so far, bad =", bad, flush=True)
print("soak done, mismatches:", bad)
