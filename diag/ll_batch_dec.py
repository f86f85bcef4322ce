# lossless batch decode of the bench shape (128 x 10 s 96 kHz stereo, level 5): wall + per-kernel times come from rocprofv3
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flo_amd
ctx = flo_amd.Context(0)
lsr, lsec, lclips, ch = 96000, 10, int(os.environ.get("CLIPS", "128")), 2
bl = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [lsr * lsec * ch] * lclips, lsr, ch, int(os.environ.get("LEVEL", "5")))
bl.fill_synthetic(seed=0xF10A0D10, clip_id0=20_000_000)
bl.encode(0); bl.sync()
n = lsr * lsec * ch * lclips
dst = torch.empty(n, dtype=torch.float32, device="cuda:0")
pcm = torch.empty(n, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
bl.decode_to(dst.data_ptr(), n)
t = time.perf_counter()
for _ in range(3): bl.decode_to(dst.data_ptr(), n)
d = (time.perf_counter() - t) / 3
print(f"decode {d*1e3:.3f} ms  {n/d/1e9:.1f} Gsamples/s")
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(3): bl.decode_to(dst.data_ptr(), n)
print("event-timed per call:", {k: round(ctx.profile_query(k)[0] / 3, 3) for k in ("ll_decode_parallel", "ll_decode", "ll_finish")})
ctx.profile_enable(False)
# exactness: the decoded floats are the 16-bit quantisation of the input
src = bl.download_pcm(0)
import numpy as np
got = dst[: lsr * lsec * ch].cpu().numpy()
q = np.clip(np.round(src.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0
print("clip 0 max |decoded - quantised input|:", float(np.abs(got - q).max()))
