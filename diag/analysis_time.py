import sys, time, numpy as np
sys.path.insert(0, '.')
import flo_amd
ctx = flo_amd.Context(0)
sr, ch = 44100, 2
for secs in (10, 180):
    n = secs * sr * ch
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n], sr, ch, 0.55)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=1)
    pcm = b.download_pcm(0)
    b.close()
    ctx.analysis_metadata(pcm, sr, ch)
    t = time.perf_counter(); m = ctx.analysis_metadata(pcm, sr, ch); dt = time.perf_counter() - t
    t = time.perf_counter(); f = ctx.encode_lossy(pcm, sr, ch, 0.55); de = time.perf_counter() - t
    print(f"{secs} s clip: flo_analysis_metadata {dt*1e3:.2f} ms ({len(m)} B), flo_encode_lossy {de*1e3:.2f} ms")
