#!/usr/bin/env python3
"""When do the CUs run dry? From the per-clip records of a FLO_STAMPS run (FLO_STAMPS_DUMP): the end time of every clip
(packer wave, 100 MHz) grouped by CU: clips per CU, and how long before the launch's end each CU took its last clip / finished."""
import sys, collections
import numpy as np
st = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2, 16)
x = st[:, 1, 13]; hw = x & np.uint64(0xFFFFFFFF)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int)
xcc = ((x >> np.uint64(32)) & np.uint64(15)).astype(int); slot = ((x >> np.uint64(40)) & np.uint64(255)).astype(int)
key = xcc * 10000 + se * 100 + sh * 50 + cu
end = st[:, 1, 14].astype(np.float64) / 100.0   # us
t_end = end.max(); t0 = end.min()
print("clips", len(end), "first clip done at", 0.0, "last at", round(t_end - t0, 1), "us after it")
per = collections.defaultdict(list)
for k, e in zip(key.tolist(), end.tolist()): per[k].append(e)
fin = np.array([max(v) for v in per.values()]); cnt = np.array([len(v) for v in per.values()])
print("CUs", len(per), "clips per CU", dict(collections.Counter(cnt.tolist())))
lag = t_end - fin
print("CU finish before the launch's end (us): min %.0f  median %.0f  p90 %.0f  max %.0f" % (lag.min(), np.median(lag), np.percentile(lag, 90), lag.max()))
for c in sorted(set(cnt.tolist())):
    print("  CUs with %d clips finish %.0f us before the end on average" % (c, lag[cnt == c].mean()))
# how many (T, P) pairs are still at work t us before the end
pair_last = collections.defaultdict(float)
for k, sl, e in zip(key.tolist(), slot.tolist(), end.tolist()): pair_last[(k, sl)] = max(pair_last[(k, sl)], e)
pl = np.array(list(pair_last.values()))
for t in (2000, 1500, 1000, 750, 500, 250, 100):
    print("  %5d us before the end: %4d of %d pairs still have a clip to finish" % (t, int((pl > t_end - t).sum()), len(pl)))
