import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flo_amd
ctx = flo_amd.Context(0)
sr, ch, n = 44100, 2, 1250
b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [sr * 10 * ch] * n, sr, ch, 0.55)
b.fill_synthetic()
b.encode(0); b.sync()
hops = (sr * 10 + 1024 + 1023) // 1024
out = torch.empty(n * (hops - 1) * 1024 * ch, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
ctx.profile_enable(True)
for _ in range(2): b.decode_to(out.data_ptr(), out.numel())
ctx.profile_reset()
t = time.perf_counter()
for _ in range(5): b.decode_to(out.data_ptr(), out.numel())
dt = (time.perf_counter() - t) / 5
ms, cnt = ctx.profile_query("lossy_decode")
print("batch decode", round(dt * 1e3, 3), "ms wall;", "kernel", round(ms / cnt, 3), "ms;", round(out.numel() / (ms / cnt) / 1e3, 1), "Msamples/s")
# single 3-min file through flo_decode (host round trip included)
b1 = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [sr * 180 * ch], sr, ch, 0.55)
b1.fill_synthetic(); b1.encode(0); b1.sync(); f = b1.fetch(0)
ctx.decode(f)
t = time.perf_counter(); ctx.decode(f); print("3-min file flo_decode", round((time.perf_counter() - t) * 1e3, 2), "ms (incl. H2D/D2H)", len(f), "bytes")
pcm = b1  # lossless 10 s file
import numpy as np
x = (np.random.default_rng(0).uniform(-0.3, 0.3, sr * 10 * ch)).astype(np.float32)
fl = ctx.encode_lossless(x, sr, ch, 16, 5)
ctx.decode(fl)
ctx.profile_reset()
t = time.perf_counter(); ctx.decode(fl); dtl = time.perf_counter() - t
print("10 s lossless file flo_decode", round(dtl * 1e3, 2), "ms;", {k: round(ctx.profile_query(k)[0], 3) for k in ("ll_decode", "ll_finish")})
