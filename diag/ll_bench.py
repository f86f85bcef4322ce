import time, sys
sys.path.insert(0, "/root/repo")
import torch, flo_amd
ctx = flo_amd.Context(0)
for (sr, secs, n) in ((96000, 10, 64), (44100, 10, 256), (44100, 10, 1250)):
    ch = 2
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [sr * secs * ch] * n, sr, ch, 5)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
    ctx.profile_enable(True)
    for _ in range(2):
        b.encode(0); b.sync()
    ctx.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        b.encode(0); b.sync()
    dt = (time.perf_counter() - t0) / reps
    ks = {k: ctx.profile_query(k) for k in ("ll_prepare", "ll_analyze", "ll_layout", "ll_pack")}
    print(sr, secs, n, "ms", round(dt * 1e3, 3), "Msamples/s", round(sr * secs * ch * n / dt / 1e6, 1), "bytes", b.data_bytes(),
          {k: round(v[0] / max(v[1], 1), 3) for k, v in ks.items()})
    b.close()
