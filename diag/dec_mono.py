# lossy batch decode: stereo against the same number of channel-frames as mono clips (are the interleaved 4-byte stores what bounds it?)
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flo_amd
ctx = flo_amd.Context(0)
sr = 44100
for ch, n in ((2, 1250), (1, 2500)):
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [sr * 10 * ch] * n, sr, ch, 0.55)
    b.fill_synthetic()
    b.encode(0); b.sync()
    hops = (sr * 10 + 1024 + 1023) // 1024
    out = torch.empty(n * (hops - 1) * 1024 * ch, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    for _ in range(2): b.decode_to(out.data_ptr(), out.numel())
    ctx.profile_reset()
    for _ in range(5): b.decode_to(out.data_ptr(), out.numel())
    ms, cnt = ctx.profile_query("lossy_decode")
    print(f"{n} clips x {ch} ch: kernel {ms / cnt:.3f} ms; {out.numel() / (ms / cnt) / 1e6:.1f} Gsamples/s; compressed {b.data_bytes() / 1e6:.1f} MB")
    b.close(); del out
