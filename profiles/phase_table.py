#!/usr/bin/env python3
"""Static per-phase instruction counts of the transform wave of lossy_chain2q_kernel from a -DFLO_MARKS assembly listing.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -Iinclude -DFLO_MARKS \
        -S --cuda-device-only -o /tmp/lk_marks.s flo_amd/csrc/lossy_kernels.hip
  python3 profiles/phase_table.py /tmp/lk_marks.s [mangled-name-substring]

The markers pin the schedule (they are memory clobbers), so the counts are those of the marked build, a few per cent
above the shipped one. Counted per stereo frame (the loop is unrolled by two: both bodies are averaged).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "lossy_chain2q_kernelILb0ELj48574ELb0EE"
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if ".size" in lines[i] and want in lines[i])
    phases = collections.OrderedDict()
    cur = "pre (clip set-up, packer wave)"
    order = []
    for l in lines[start:end]:
        s = l.strip()
        m = re.match(r"; MARK (\w+)", s)
        if m:
            cur = m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        k = classify(op)
        if k is None:
            continue
        # the phase a marker NAMES ends at the marker: instructions belong to the NEXT marker seen; collect by previous
        phases.setdefault(cur, collections.Counter())[k] += 1
        if cur not in order:
            order.append(cur)
    # instructions after marker X and before marker Y belong to phase Y
    # (round 4: the frame loop is software-pipelined - the next frame's fold runs at the end of a frame, the masking pass of the
    # previous frame inside the FFT, behind the first exchange)
    # what FOLLOWS each marker (round 4: the frame loop is software-pipelined - the next frame's fold runs at the end of a
    # frame, the masking pass of the previous frame inside the FFT, behind the first exchange)
    names = {"frame_begin": "prefetch (16 loads), masking rows", "prefetch_done": "fft512 x2 + masking pass (prev. frame)",
             "fft_done": "post-rotation + transpose", "postrot_done": "band_stats_2",
             "bandstats_done": "fold of the next frame (a)", "frame_end": "fold of the next frame (b), loop"}
    print(f"{'phase':32s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'wait':>5s}   (per stereo frame, mean of the two unrolled bodies)")
    tot = collections.Counter()
    for ph in order:
        c = phases[ph]
        nm = names.get(ph, ph)
        div = 1 if ph.startswith("pre (") else 2
        row = {k: c[k] / div for k in ("valu", "salu", "lds", "vmem", "wait")}
        if not ph.startswith("pre ("):
            for k, v in row.items():
                tot[k] += v
        print(f"{nm:32s} {row['valu']:6.0f} {row['salu']:6.0f} {row['lds']:5.0f} {row['vmem']:5.0f} {row['wait']:5.0f}")
    print(f"{'transform wave, per frame':32s} {tot['valu']:6.0f} {tot['salu']:6.0f} {tot['lds']:5.0f} {tot['vmem']:5.0f} {tot['wait']:5.0f}")


if __name__ == "__main__":
    main()
