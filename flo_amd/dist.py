"""Data-parallel sharding of a clip corpus across the GPUs of one node, and the one exchange step of the path:
a variable-size gather of the packed bitstreams to rank 0 (RCCL over xGMI when the process group is "nccl").

Clips are independent, so ranks never talk during the encode; the only communication is
  1. all_gather of one int64 per rank (payload bytes), and
  2. rank r > 0 sends its packed payload straight to rank 0 (point-to-point: on the fully connected xGMI node each
     peer has its own link to the root, so the seven transfers run in parallel; a ring collective would be slower).
The payload is the compressed stream (about a tenth of the PCM bytes), so this step is small next to the encode.
"""
from typing import List, Sequence, Tuple


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


def gather_payloads(dist, payload, rank: int, world: int, dst: int = 0):
    """Variable-size gather of one uint8 tensor per rank to rank `dst`.
    Returns (list of tensors by rank on dst, None elsewhere), and the int64 size vector on every rank."""
    import torch
    sizes = torch.zeros(world, dtype=torch.int64, device=payload.device)
    mine = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes_h = [int(x) for x in sizes.tolist()]
    if rank == dst:
        out = [None] * world
        out[dst] = payload
        reqs = []
        for r in range(world):
            if r == dst:
                continue
            out[r] = torch.empty(sizes_h[r], dtype=torch.uint8, device=payload.device)
            if sizes_h[r]:
                reqs.append(dist.irecv(out[r], src=r))
        for q in reqs:
            q.wait()
        return out, sizes_h
    if payload.numel():
        dist.isend(payload, dst=dst).wait()
    return None, sizes_h


class BitstreamGather:
    """Per-step gather used by bench.py: pack this rank's finished .flo files (header, TOC and CRC are made on the
    device) into one device tensor, then gather to rank 0."""

    def __init__(self, ctx, batch, dist, rank, world, local_rank):
        import torch
        self.ctx, self.batch, self.dist, self.rank, self.world = ctx, batch, dist, rank, world
        self.device = torch.device("cuda", local_rank)
        self.buf = None
        self.last_total = 0
        # header + TOC of every file: 74 + 20 bytes per frame; a frame is at most 1 s (lossless) or 1024 samples (lossy)
        self.head_bytes = sum(74 + 20 * (n // (1024 * 1) + 2) for n in batch.n_interleaved)

    def run(self):
        import torch
        need = self.batch.data_bytes() + self.head_bytes + 16 * self.batch.n_clips + 64
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        offs = self.batch.pack_files(self.buf.data_ptr(), self.buf.numel())
        self.batch.sync()
        payload = self.buf[: offs[-1]]
        got, sizes = gather_payloads(self.dist, payload, self.rank, self.world, 0)
        self.last_total = sum(sizes)
        return got, offs
