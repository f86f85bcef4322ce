"""Reference implementation of the exchange step's protocol over a torch.distributed process group (test infrastructure:
the product's exchange step is flo_dist_* behind the C ABI). The world-size-2 `gloo` tests of test_dist_cpu.py and the
one-rank "nccl" test of test_gpu_lossy.py drive it; it documents the protocol in twenty lines of Python:
  1. all_gather of one int64 per rank (payload bytes), and
  2. rank r > 0 sends its packed payload straight to rank 0 (point-to-point: on the fully connected xGMI node each peer
     has its own link to the root, so the seven transfers run in parallel; a ring collective would be slower).
"""


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


class PipelinedGather:
    """The same variable-size gather, split into submit / retire so that the transfer of step k runs while step
    k + 1 is being encoded (the payload is about 240 MB per rank and step: over one xGMI link that takes about as long
    as the encode itself, so a blocking gather would halve the throughput of every N > 1).

    submit(slot, payload): exchange the sizes (one int64 per rank; blocking but tiny), then post the point-to-point
    transfers without waiting. The payload tensor of a slot must stay untouched until retire(slot) returns.
    retire(slot): wait for the slot's transfers; on the root returns the list of received tensors by rank."""

    def __init__(self, dist, rank: int, world: int, dst: int = 0, slots: int = 2):
        self.dist, self.rank, self.world, self.dst = dist, rank, world, dst
        self.pending = [None] * slots       # per slot: (requests, out list or None, payload)
        self.recv = [dict() for _ in range(slots)]   # root: reusable receive buffers per slot and rank
        self.last_sizes = None

    def submit(self, slot: int, payload):
        import torch
        assert self.pending[slot] is None, "retire the slot before reusing it"
        dist, world, rank, dst = self.dist, self.world, self.rank, self.dst
        sizes = torch.zeros(world, dtype=torch.int64, device=payload.device)
        mine = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
        dist.all_gather_into_tensor(sizes, mine)
        sizes_h = [int(x) for x in sizes.tolist()]
        self.last_sizes = sizes_h
        reqs, out = [], None
        if rank == dst:
            out = [None] * world
            out[dst] = payload
            ops = []
            for r in range(world):
                if r == dst:
                    continue
                buf = self.recv[slot].get(r)
                if buf is None or buf.numel() < sizes_h[r]:
                    buf = torch.empty(int(sizes_h[r] * 1.25) + 64, dtype=torch.uint8, device=payload.device)
                    self.recv[slot][r] = buf
                out[r] = buf[: sizes_h[r]]
                if sizes_h[r]:
                    ops.append(dist.P2POp(dist.irecv, out[r], r))
            if ops:
                reqs = dist.batch_isend_irecv(ops)
        elif payload.numel():
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, payload, dst)])
        self.pending[slot] = (reqs, out, payload)

    def retire(self, slot: int):
        import torch
        p = self.pending[slot]
        if p is None:
            return None
        reqs, out, payload = p
        for q in reqs:
            q.wait()
        if payload.is_cuda:
            # a NCCL wait() only orders the current stream behind the transfer; the buffers are reused by kernels on
            # another stream, so the host waits too
            torch.cuda.current_stream(payload.device).synchronize()
        self.pending[slot] = None
        return out

    def flush(self):
        return [self.retire(s) for s in range(len(self.pending))]


class BitstreamGather:
    """Per-step gather used by bench.py: pack this rank's finished .flo files (header, TOC and CRC are made on the
    device) into one of two device buffers and hand it to the pipelined gather; the transfer overlaps the next step's
    encode. flush() at the end of the job waits for the last transfers."""

    def __init__(self, ctx, batch, dist, rank, world, local_rank):
        import torch
        self.ctx, self.batch, self.dist, self.rank, self.world = ctx, batch, dist, rank, world
        self.device = torch.device("cuda", local_rank)
        self.bufs = [None, None]
        self.step = 0
        self.pipe = PipelinedGather(dist, rank, world, 0, 2)
        # header + TOC of every file: 74 + 20 bytes per frame; a frame is at most 1 s (lossless) or 1024 samples (lossy)
        self.head_bytes = sum(74 + 20 * (n // 1024 + 2) for n in batch.n_interleaved)

    def run(self):
        import torch
        slot = self.step % 2
        self.step += 1
        self.pipe.retire(slot)               # the transfer that used this buffer two steps ago
        need = self.batch.data_bytes() + self.head_bytes + 16 * self.batch.n_clips + 64
        if self.bufs[slot] is None or self.bufs[slot].numel() < need:
            self.bufs[slot] = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
        offs = self.batch.pack_files(self.bufs[slot].data_ptr(), self.bufs[slot].numel())
        self.batch.sync()
        self.pipe.submit(slot, self.bufs[slot][: offs[-1]])
        return offs

    def flush(self):
        return self.pipe.flush()

    @property
    def last_total(self):
        return sum(self.pipe.last_sizes) if self.pipe.last_sizes else 0


