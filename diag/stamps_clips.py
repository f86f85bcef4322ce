#!/usr/bin/env python3
"""Per-clip records of a FLO_STAMPS run (FLO_STAMPS_DUMP=file): transform / packer busy ticks per frame by SIMD, XCC, slot."""
import sys
import numpy as np
st = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 16)
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 432
t = st[:, 0, :].astype(np.float64); p = st[:, 1, :].astype(np.float64)
tb = (t[:, :9].sum(1) - t[:, 7]) / hops
pb = p[:, 1:5].sum(1) / hops
def ids(x):
    x = x.astype(np.uint64)
    hw = x & np.uint64(0xFFFFFFFF)
    return dict(wave=(hw & np.uint64(15)).astype(int), simd=((hw >> np.uint64(4)) & np.uint64(3)).astype(int),
                cu=((hw >> np.uint64(8)) & np.uint64(15)).astype(int), se=((hw >> np.uint64(13)) & np.uint64(7)).astype(int),
                xcc=((x >> np.uint64(32)) & np.uint64(15)).astype(int), slot=((x >> np.uint64(40)) & np.uint64(255)).astype(int))
ti, pi = ids(st[:, 0, 13]), ids(st[:, 1, 13])
print("clips", len(tb), "T busy mean", tb.mean(), "P busy mean", pb.mean())
for key in ("simd", "slot", "xcc", "se", "wave"):
    print("T by", key, {int(k): (round(float(tb[ti[key] == k].mean())), int((ti[key] == k).sum())) for k in np.unique(ti[key])})
for key in ("simd", "slot", "xcc"):
    print("P by", key, {int(k): (round(float(pb[pi[key] == k].mean())), int((pi[key] == k).sum())) for k in np.unique(pi[key])})
slow = tb > 11000
print("slow T clips:", slow.mean(), "their P busy", pb[slow].mean(), "others' P busy", pb[~slow].mean())
print("slow T by (T simd, P simd):")
for a in range(4):
    print("  ", [f"{(slow & (ti['simd'] == a) & (pi['simd'] == b)).sum()}/{((ti['simd'] == a) & (pi['simd'] == b)).sum()}" for b in range(4)])
print("corr(T busy, P busy) =", np.corrcoef(tb, pb)[0, 1])
for i, nm in enumerate(["fold", "prefetch", "fft", "postrot", "bandstats", "mask", "quant", "wait-consumed", "handover"]):
    print(f"  T {nm:14s} fast clips {t[~slow, i].mean() / hops:8.0f}   slow clips {t[slow, i].mean() / hops:8.0f}")
