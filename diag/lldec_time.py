# lossless decode: parallel form against the serial kernel (same bytes out), per-kernel times
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, flo_amd
ctx = flo_amd.Context(0)
sr, ch = 44100, 2
rng = np.random.default_rng(0)
t = np.arange(sr * 10) / sr
music = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1234.5 * t) + 0.02 * rng.standard_normal(t.size))
cases = {
    "noise 10 s": rng.uniform(-0.3, 0.3, sr * 10 * ch).astype(np.float32),
    "tones 10 s": np.stack([music, 0.7 * music + 0.01 * rng.standard_normal(t.size)], 1).reshape(-1).astype(np.float32),
    "quiet 10 s": (rng.standard_normal(sr * 10 * ch) * 1e-4).astype(np.float32),
    "tones 180 s": np.tile(np.stack([music, 0.5 * music], 1).reshape(-1), 18).astype(np.float32),
}
names = ("ll_decode_parallel", "ll_decode", "ll_finish")
for name, x in cases.items():
    fl = ctx.encode_lossless(x, sr, ch, 16, 5)
    ctx.profile_enable(True)
    res = {}
    for mode in ("parallel", "serial"):
        if mode == "serial": os.environ["FLO_LL_DECODE_SERIAL"] = "1"
        else: os.environ.pop("FLO_LL_DECODE_SERIAL", None)
        ctx.decode(fl)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(3): out = ctx.decode_lossless_i32(fl)
        dt = (time.perf_counter() - t0) / 3
        res[mode] = out[0] if isinstance(out, tuple) else out
        import ctypes as C
        outp, nn, srr, chh = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint8()
        t1 = time.perf_counter()
        ctx._L.flo_decode(ctx._h, fl, len(fl), C.byref(outp), C.byref(nn), C.byref(srr), C.byref(chh))
        dtc = time.perf_counter() - t1
        ctx._L.flo_free(outp)
        print(f"      flo_decode C call {dtc * 1e3:.2f} ms", end="")
        prof = {k: round(ctx.profile_query(k)[0] / max(1, ctx.profile_query(k)[1]), 3) for k in names}
        print(f"{name:12s} {mode:8s} wall {dt * 1e3:7.2f} ms  {prof}  bytes {len(fl)}", flush=True)
    print("   same integers:", bool(np.array_equal(np.asarray(res["parallel"]), np.asarray(res["serial"]))), flush=True)
