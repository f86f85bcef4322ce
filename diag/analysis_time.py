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
    ctx.analysis_metadata(pcm, sr, ch)
    t = time.perf_counter(); m = ctx.analysis_metadata(pcm, sr, ch); dt = time.perf_counter() - t
    b.analysis_metadata(0)
    t = time.perf_counter(); m2 = b.analysis_metadata(0); db = time.perf_counter() - t
    assert m == m2
    b.close()
    t = time.perf_counter(); f = ctx.encode_lossy(pcm, sr, ch, 0.55); de = time.perf_counter() - t
    flo_amd.encode_lossy(pcm, sr, ch, 16, 2)
    t = time.perf_counter(); g = flo_amd.encode_lossy(pcm, sr, ch, 16, 2); dg = time.perf_counter() - t
    print(f"{secs} s clip: flo_analysis_metadata {dt*1e3:.2f} ms ({len(m)} B), on a batch's device copy {db*1e3:.2f} ms, flo_encode_lossy {de*1e3:.2f} ms, "
          f"flo_amd.encode_lossy (upload + analysis + encode + fetch) {dg*1e3:.2f} ms")
