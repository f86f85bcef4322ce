"""Data-parallel sharding of a clip corpus across the GPUs of one node, and the one exchange step of the path:
a variable-size gather of every rank's finished files to rank 0 (RCCL over xGMI).

The exchange step is `NativeGather` below: a thin caller of flo_dist_* in the C ABI (RCCL directly, own stream,
double-buffered; its ordering logic is the C++ template flo_amd/csrc/dist_engine.hpp, which
tests/native/dist_engine_test.cpp also runs with several ranks over sockets on the CPU).

Clips are independent, so ranks never talk during the encode; the only communication is
  1. all_gather of one int64 per rank (payload bytes), and
  2. rank r > 0 sends its packed payload straight to rank 0 (point-to-point: on the fully connected xGMI node each
     peer has its own link to the root, so the seven transfers run in parallel; a ring collective would be slower).
The payload is the compressed stream (about a tenth of the PCM bytes), so this step is small next to the encode.
"""
import ctypes as C
from typing import List, Sequence, Tuple

ID_BYTES = 128


def shard_clips(n_samples: Sequence[int], world: int) -> List[List[int]]:
    """Static partition of clip indices over `world` ranks, balanced by sample count (longest-first greedy).
    Deterministic, and every clip lands on exactly one rank."""
    order = sorted(range(len(n_samples)), key=lambda i: (-int(n_samples[i]), i))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(n_samples[i])
    for s in shards:
        s.sort()
    return shards


def contiguous_shard(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of equal-length clips for one rank (used by the benchmark's synthetic corpus)."""
    per, rem = divmod(n_clips, world)
    start = rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)


def unique_id() -> bytes:
    """flo_dist_unique_id: the RCCL rendezvous token rank 0 makes and shares with the other ranks by any side channel."""
    from . import _native
    L = _native.lib()
    buf = C.create_string_buffer(ID_BYTES)
    if L.flo_dist_unique_id(buf) != 0:
        raise _native.FloError(L.flo_last_create_error().decode())
    return buf.raw


class NativeGather:
    """The multi-GPU exchange step behind the C ABI (flo_dist_* in include/flo_hip.h): RCCL directly, on the library's
    own communication stream, double-buffered, no host synchronisation per step. This object only forwards.

        g = NativeGather(ctx, id_bytes, rank, world)      # all ranks; id_bytes = unique_id() made on rank 0
        per step:  batch.encode(); batch.sync(); g.submit(batch)
        at the end: g.flush();  root: g.result() -> (device pointer, offsets by rank, sizes by rank)
    """

    def __init__(self, ctx, id_bytes: bytes, rank: int, world: int, root: int = 0):
        assert len(id_bytes) == ID_BYTES
        self.ctx, self._L, self.rank, self.world, self.root = ctx, ctx._L, rank, world, root
        h = C.c_void_p()
        ctx._chk(self._L.flo_dist_create(ctx._h, id_bytes, rank, world, root, C.byref(h)))
        self._h = h
        if not hasattr(ctx, "_dists"):
            ctx._dists = set()
        ctx._dists.add(self)

    def submit(self, batch):
        self.ctx._chk(self._L.flo_dist_gather_submit(self._h, batch._h))

    def flush(self):
        self.ctx._chk(self._L.flo_dist_gather_flush(self._h))

    def result(self):
        base = C.c_void_p()
        offs, sizes = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        self.ctx._chk(self._L.flo_dist_gather_result(self._h, C.byref(base), C.byref(offs), C.byref(sizes)))
        return base.value, [offs[i] for i in range(self.world)], [sizes[i] for i in range(self.world)]

    # second exchange mode: the files stay on their ranks, every rank learns (offset, size, CRC32) of every file
    def table_submit(self, batch, max_clips: int):
        self.ctx._chk(self._L.flo_dist_table_submit(self._h, batch._h, max_clips))

    def table_flush(self):
        self.ctx._chk(self._L.flo_dist_table_flush(self._h))

    def table_result(self):
        """per rank: (sizes, offsets, crc32s) of its clips, as the last submitted step left them"""
        rows, words, mx = C.POINTER(C.c_uint64)(), C.c_size_t(), C.c_size_t()
        self.ctx._chk(self._L.flo_dist_table_result(self._h, C.byref(rows), C.byref(words), C.byref(mx)))
        out = []
        for r in range(self.world):
            base = r * words.value
            n = int(rows[base])
            out.append(([int(rows[base + 1 + i]) for i in range(n)],
                        [int(rows[base + 1 + mx.value + i]) for i in range(n)],
                        [int(rows[base + 1 + 2 * mx.value + i]) for i in range(n)]))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.flo_dist_destroy(self._h)
            self._h = None
            getattr(self.ctx, "_dists", set()).discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
