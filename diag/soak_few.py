"""Soak of the FEW-clips frame-parallel path (coefficients and band maxima handed from pass 1 to pass 2, the temporal chain
inside pass 2, fused offsets + packing, CRC slices + TOC in one launch): its bytes against the chain forms', the header's CRC
against zlib's over the DATA chunk, the TOC against the frame sizes - several shapes and qualities, repeated."""
import sys, os, struct, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flo_amd
ctx = flo_amd.Context(0)
sr, ch = 44100, 2
def files(b, form):
    b.encode(form); b.sync()
    return [b.fetch(i) for i in range(b.n_clips)]
def check_file(f):
    assert f[:4] == b"FLO!", f[:4]
    crc = struct.unpack_from("<I", f, 26)[0]
    toc_size = struct.unpack_from("<Q", f, 38)[0]; data_size = struct.unpack_from("<Q", f, 46)[0]
    nf = struct.unpack_from("<I", f, 70)[0]
    assert toc_size == 4 + 20 * nf
    data = f[70 + toc_size: 70 + toc_size + data_size]
    assert len(data) == data_size and zlib.crc32(data) == crc, ("crc", hex(crc), hex(zlib.crc32(data)))
    off = 0
    for i in range(nf):
        idx, o, sz, ts = struct.unpack_from("<IQII", f, 74 + 20 * i)
        assert idx == i and o == off, (i, idx, o, off)
        off += sz
    assert off == data_size
bad = 0
shapes = ([sr * 180 * ch], [sr * 60 * ch] * 3, [(50000 + 33331 * i) * ch for i in range(5)], [sr * 10 * ch], [1, 2047 * ch, 2048 * ch, 2049 * ch],
          [sr * 10 * ch] * 16, [sr * 3 * ch] * 17)
for shape in shapes:
    for q in (0.2, 0.55, 1.0):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, shape, sr, ch, q)
        b.fill_synthetic(seed=4321, clip_id0=11)
        ref = files(b, 5)
        for f in ref: check_file(f)
        for it in range(4):
            for form in (2, 2, 1):
                got = files(b, form)
                if got != ref:
                    bad += 1; print("MISMATCH", len(shape), q, it, form)
        b.close()
        print("shape", len(shape), "clips, q", q, "ok so far, bad =", bad, flush=True)
print("few-clips soak done, mismatches:", bad)
