"""N > 1 host logic on CPU: clip sharding; the exchange step's C++ ordering logic (flo_amd/csrc/dist_engine.hpp, the template
flo_dist_* instantiates with RCCL) run with 2, 3 and 5 ranks over sockets; and the protocol's Python reference
(tests/dist_ref.py), world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dist_ref import PipelinedGather, gather_payloads
from flo_amd.dist import contiguous_shard, shard_clips


def test_shard_clips_partition_and_balance():
    rng = np.random.default_rng(0)
    lens = rng.integers(1000, 500000, 101).tolist()
    for world in (1, 2, 4, 8):
        shards = shard_clips(lens, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lens)))
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)
    assert shard_clips([], 4) == [[], [], [], []]
    assert shard_clips(lens, 2) == shard_clips(lens, 2)


def test_contiguous_shard_covers_everything():
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 3, 8):
            spans = [contiguous_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert contiguous_shard(10000, 3, 8) == (3750, 5000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        for trial, n in enumerate([(1000 + 37 * rank), 0 if rank == 1 else 5, 4096]):
            payload = torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8))
            got, sizes = gather_payloads(dist, payload, rank, world, 0)
            assert sizes[rank] == n
            if rank == 0:
                assert len(got) == world
                for r in range(world):
                    exp = np.random.default_rng(100 + r)
                    for t2, n2 in enumerate([(1000 + 37 * r), 0 if r == 1 else 5, 4096]):
                        e = exp.integers(0, 256, n2, dtype=np.uint8)
                        if t2 == trial:
                            assert got[r].numpy().tobytes() == e.tobytes(), (trial, r)
            else:
                assert got is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_variable_size_gather_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _payload(rank, step):
    rng = np.random.default_rng(1000 * rank + step)
    n = [3000, 0, 70000, 17, 4096, 1][step % 6] + 11 * rank
    return torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8))


def _pipe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = PipelinedGather(dist, rank, world, 0, 2)
        steps = 7
        seen = []
        for k in range(steps):
            slot = k % 2
            out = pipe.retire(slot)          # the transfer of step k - 2
            if out is not None and rank == 0:
                seen.append((k - 2, [t.clone() for t in out]))
            pipe.submit(slot, _payload(rank, k))
        # the two transfers still in flight, oldest first
        first = steps % 2
        for j, slot in enumerate((first, 1 - first)):
            out = pipe.retire(slot)
            if rank == 0:
                seen.append((steps - 2 + j, [t.clone() for t in out]))
        assert pipe.flush() == [None, None]
        if rank == 0:
            assert [k for k, _ in seen] == list(range(steps))
            for k, out in seen:
                for r in range(world):
                    assert out[r].numpy().tobytes() == _payload(r, k).numpy().tobytes(), (k, r)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_dist_engine_state_machine_with_several_ranks(tmp_path):
    # the product's own slot / ordering logic (DistEngine: deferred posting, slot parity, growing buffers, empty payloads,
    # a flush in the middle), bound to sockets instead of RCCL: one process per rank
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "dist_engine_test")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "dist_engine_test.cpp")], check=True)
    for world, steps in ((1, 7), (2, 7), (3, 8), (5, 6), (2, 1), (3, 2)):
        r = subprocess.run([exe, str(world), str(steps)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (world, steps, r.stdout, r.stderr)
        assert "runtime error" not in r.stderr, r.stderr
