import sys
sys.path.insert(0, "/root/repo")
import torch, flo_amd
ctx = flo_amd.Context(0)
sr, ch, n = 44100, 2, 1250
b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [sr * 10 * ch] * n, sr, ch, 0.55)
b.fill_synthetic()
ctx.profile_enable(True)
for _ in range(3):
    b.encode(0); b.sync()
ctx.profile_reset()
for _ in range(10):
    b.encode(0); b.sync()
for k in ("lossy_chain3", "finish_files"):
    ms, cnt = ctx.profile_query(k)
    print(k, round(ms / max(cnt, 1), 4), "ms")
