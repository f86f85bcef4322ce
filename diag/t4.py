import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flo_amd
ctx = flo_amd.Context(0)
paths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,4").split(",")]
for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256,1024,1250,10000").split(",")]:
    bt = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [441000 * 2] * n, 44100, 2, 0.55)
    bt.fill_synthetic()
    for path in paths:
        for _ in range(2): bt.encode(path); bt.sync()
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(5): bt.encode(path); bt.sync()
        ctx.profile_enable(False)
        ms, k = ctx.profile_query({1: "lossy_chain", 3: "lossy_chain3", 4: "lossy_chain2x"}[path])
        print(os.environ.get("FLO_HIP_LIB", "product").split("/")[-1], "clips", n, "path", path, round(ms / max(k, 1), 4), "ms", flush=True)
    bt.close()
