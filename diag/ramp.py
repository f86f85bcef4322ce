"""Per-launch time of the 10 000-clip encode over a long run: does the clock ramp up or throttle down?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flo_amd
ctx = flo_amd.Context(0)
n = 441000 * 2
b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n] * 10000, 44100, 2, 0.55)
b.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    t = time.perf_counter(); b.encode(0); b.sync(); ts.append((time.perf_counter() - t) * 1e3)
for i in range(0, len(ts), 10):
    print(f"steps {i:3d}-{i+9:3d}: {sum(ts[i:i+10])/len(ts[i:i+10]):.3f} ms per step", flush=True)
