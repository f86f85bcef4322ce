"""Soak: the lock-step stereo, three-wave and two-wave chain kernels against the frame-parallel form, many times, several batch shapes."""
import sys
sys.path.insert(0, "/root/repo")
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
    for q in (0.55, 1.0):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, shape, sr, ch, q)
        b.fill_synthetic(seed=1234, clip_id0=7)
        ref = packed(b, 2)
        for it in range(12):
            for form in (4, 3, 1):
                got = packed(b, form)
                if got.shape != ref.shape or not torch.equal(got, ref):
                    bad += 1
                    print("MISMATCH", len(shape), q, it, form)
        b.close()
        print("shape", len(shape), "q", q, "ok so far, bad =", bad, flush=True)
print("soak done, mismatches:", bad)
