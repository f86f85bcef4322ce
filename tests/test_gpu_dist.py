"""The multi-GPU exchange step behind the C ABI (flo_dist_*: RCCL directly). A gpurun box has one GPU: the single-rank
case runs the whole code path (all-gather of sizes, root copy, double buffering, deferred posting); the two-rank case
puts two processes on the same GPU, which RCCL refuses with "invalid usage" on this pool - ONLY that refusal skips the
test, every other error or a timeout fails it. The ordering logic itself runs with several ranks on the CPU
(tests/native/dist_engine_test.cpp through test_dist_cpu.py); N > 1 on hardware is the driver's multi-GPU run."""
import ctypes
import multiprocessing as mp
import os

import numpy as np
import pytest

import flofile
import signals
from gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def _d2h(ptr, n):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    buf = (ctypes.c_uint8 * max(n, 1))()
    assert hip.hipMemcpy(buf, ptr, n, 2) == 0
    return bytes(buf[:n])


def _split_files(blob):
    """finished .flo files packed back to back at 16-byte aligned offsets: lengths come from their headers"""
    out, pos = [], 0
    while pos + 70 <= len(blob):
        assert blob[pos:pos + 4] == b"FLO!", pos
        toc, data = int.from_bytes(blob[pos + 38:pos + 46], "little"), int.from_bytes(blob[pos + 46:pos + 54], "little")
        n = 70 + toc + data
        out.append(blob[pos:pos + n])
        pos += (n + 15) & ~15
    return out


def test_single_rank_gather_returns_the_batch_files(ctx):
    import flo_amd
    from flo_amd.dist import NativeGather, unique_id
    lens = [0, 1000, 44100, 70001, 3 * 1024, 22050]
    clips = [signals.music_like(44100, n, 2, seed=60 + i) for i, n in enumerate(lens)]
    g = NativeGather(ctx, unique_id(), 0, 1, 0)
    for mode, qol in ((flo_amd.MODE_LOSSY, 0.55), (flo_amd.MODE_LOSSLESS, 5)):
        b = flo_amd.Batch(ctx, mode, [c.size for c in clips], 44100, 2, qol)
        for i, c in enumerate(clips):
            b.upload(i, c)
        for step in range(4):          # several steps: both buffer slots, deferred posting, buffer reuse
            b.encode(0)
            b.sync()
            g.submit(b)
        g.flush()
        base, offs, sizes = g.result()
        assert offs == [0] and sizes[0] > 0
        files = _split_files(_d2h(base, sizes[0]))
        assert len(files) == len(clips)
        for i in range(len(clips)):
            assert files[i] == b.fetch(i), (mode, i)
            assert flofile.parse(files[i]).crc_valid
        b.close()
    g.close()


def test_single_rank_table_names_every_file(ctx):
    # the second exchange mode: files stay where they are, one all-gather of (size, offset, CRC32) per clip. With one rank
    # the whole code path runs (table kernel, ncclAllGather, both slots); the rows must describe the batch's files.
    import zlib
    import flo_amd
    from flo_amd.dist import NativeGather, unique_id
    lens = [0, 1000, 44100, 70001, 3 * 1024]
    clips = [signals.music_like(44100, n, 2, seed=80 + i) for i, n in enumerate(lens)]
    assert ctx.reserved_cus() == 0
    g = NativeGather(ctx, unique_id(), 0, 1, 0)
    assert ctx.reserved_cus() == 0          # a one-rank communicator reserves nothing
    for mode, qol in ((flo_amd.MODE_LOSSY, 0.55), (flo_amd.MODE_LOSSLESS, 5)):
        b = flo_amd.Batch(ctx, mode, [c.size for c in clips], 44100, 2, qol)
        for i, c in enumerate(clips):
            b.upload(i, c)
        for step in range(3):
            b.encode(0)
            b.sync()
            g.table_submit(b, max_clips=len(clips) + 3)
        g.table_flush()
        (sizes, offs, crcs), = g.table_result()
        assert len(sizes) == len(clips)
        for i in range(len(clips)):
            f = b.fetch(i)
            assert sizes[i] == len(f), (mode, i)
            assert crcs[i] == zlib.crc32(flofile.parse(f).data), (mode, i)
        assert sorted(offs) == offs and len(set(offs)) == len(offs)
        with pytest.raises(flo_amd.FloError, match="max_clips"):
            g.table_submit(b, max_clips=2)
        b.close()
    g.close()


def _rank_main(rank, world, idq, outq):
    try:
        import flo_amd
        from flo_amd.dist import NativeGather, unique_id
        c = flo_amd.Context(0)
        if rank == 0:
            tok = unique_id()
            for _ in range(world - 1):
                idq.put(tok)
        else:
            tok = idq.get(timeout=120)
        g = NativeGather(c, tok, rank, world, 0)
        # with more than one rank the persistent encode kernels leave compute units to RCCL's kernels by default: the
        # first ncclSend / ncclRecv under the reserved-CU launch is then a tested configuration
        assert c.reserved_cus() == 8, c.reserved_cus()
        lens = [5000 + 777 * rank, 30000, 1024 * (rank + 1)]
        clips = [signals.music_like(44100, n, 2, seed=100 * rank + i) for i, n in enumerate(lens)]
        b = flo_amd.Batch(c, flo_amd.MODE_LOSSY, [x.size for x in clips], 44100, 2, 0.55)
        for i, x in enumerate(clips):
            b.upload(i, x)
        for _ in range(3):
            b.encode(0)
            b.sync()
            g.submit(b)
        g.flush()
        mine = [b.fetch(i) for i in range(len(clips))]
        g.table_submit(b, max_clips=4)
        g.table_flush()
        table = g.table_result()
        assert [len(t[0]) for t in table] == [3] * world and table[rank][0] == [len(f) for f in mine]
        if rank == 0:
            base, offs, sizes = g.result()
            got = [_split_files(_d2h(base + offs[r], sizes[r])) for r in range(world)]
            outq.put(("root", got, mine, table))
        else:
            outq.put(("peer", rank, mine))
        g.close()
        assert c.reserved_cus() == 0     # the default reservation ends with the communicator
        c.close()
    except Exception as e:   # noqa: BLE001
        outq.put(("error", rank, repr(e)))


def test_two_ranks_on_one_gpu_if_rccl_allows_it():
    mpc = mp.get_context("spawn")
    idq, outq = mpc.Queue(), mpc.Queue()
    world = 2
    procs = [mpc.Process(target=_rank_main, args=(r, world, idq, outq)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(outq.get(timeout=240))
    except Exception:   # noqa: BLE001
        for p in procs:
            p.kill()
        # a rank that died or hung is a failure: the one condition that excuses this test is RCCL's documented refusal
        pytest.fail(f"a rank did not report within 240 s (got {[r[:2] for r in res]})")
    for p in procs:
        p.join(timeout=60)
    errs = [r for r in res if r[0] == "error"]
    if errs:
        msgs = " | ".join(e[2][:300] for e in errs)
        refused = ("invalid usage" in msgs.lower() or "duplicate gpu" in msgs.lower()) and "ncclCommInitRank" in msgs
        if refused and all("ncclCommInitRank" in e[2] for e in errs):
            pytest.skip(f"RCCL refuses two ranks on one GPU (ncclCommInitRank: invalid usage): {msgs[:200]}")
        pytest.fail(f"the two-rank exchange failed: {msgs}")
    root = [r for r in res if r[0] == "root"][0]
    peer = [r for r in res if r[0] == "peer"][0]
    got = root[1]
    assert got[0] == root[2]          # the root's own files
    assert got[1] == peer[2]          # the peer's files arrived byte for byte
    assert root[3][1][0] == [len(f) for f in peer[2]]    # and the table mode names the peer's files by size


def test_bench_exchange_legs_with_one_rank():
    # bench.py's multi-GPU legs (the gather inside `value`, the strong-scaling shard of configs[3], the table mode) run
    # here with a one-rank communicator: the keys the driver's N > 1 runs will carry exist and hold sane numbers
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-exchange", "--clips-per-gpu", "96", "--clip-seconds", "1",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-single-clip", "--no-lossless", "--no-shard", "--no-e2e"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    ex = d["exchange_detail"]
    assert d["value"] > 0 and ex["valid"] is True and "byte for byte" in ex["self_check"] and ex["reserved_cus"] == 0
    s, t = ex["strong_96"], ex["table_exchange"]
    assert s["scaling"] == "strong" and s["value"] > 0 and s["encode_only"]["value"] >= s["value"] * 0.5
    assert len(s["bytes_per_rank_per_step"]) == 1 and s["bytes_per_rank_per_step"][0] > 96 * 1000
    assert t["files_named"] == 96 and t["bytes_named"] > 96 * 1000 and t["value"] > 0
    assert ex["encode_only"]["value"] > 0 and ex["bytes_per_rank_per_step"][0] > 0
