"""flo_encode_batch on 64 host buffers: wall time per call (diagnostic; run under rocprofv3 --kernel-trace
--memory-copy-trace for the device timeline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flo_amd
ctx = flo_amd.Context(0)
n = 441000 * 2
b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n] * 64, 44100, 2, 0.55)
b.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
clips = [np.ascontiguousarray(b.download_pcm(i)) for i in range(64)]
b.close()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t = time.perf_counter()
    outs = ctx.encode_batch(1, clips, 44100, 2, 0.55)
    dt = time.perf_counter() - t
    print(f"call {it}: {dt * 1e3:.2f} ms -> {64 * n / dt / 1e6:.0f} Msamples/s", flush=True)
