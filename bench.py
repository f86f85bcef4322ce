#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of lossy quality=high (0.55) encode of synthetic 44.1 kHz stereo.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the encode hot path over the rank's batch of clips with the PCM already resident in HBM
and the bitstreams left in HBM; for N > 1 the step ends with the RCCL gather of the packed bitstreams to rank 0.
Workload (config.workload): BASELINE.json configs[3], the 10 000-clip corpus of synthetic 10-second stereo clips
(35 GB of f32 PCM, resident in HBM), whole on every GPU — the configuration the metric is quoted on; it fits one
MI355X. The same per-GPU work at every N (weak scaling: N GPUs encode N corpora). At N = 1 the run also times the
1250-clip per-GPU shard of the 8-way split ("shard_1250": one launch that fills the chip exactly once) and the single
3-minute clip of configs[1] ("single_clip_180s": 63 MB of PCM, lives in the Infinity Cache and lasts a fraction of a
millisecond, so it cannot carry an HBM-roofline claim).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SAMPLE = 4.0 + 2.0 + 50.0 / 1024.0   # f32 in + i16 out + 25 u16 scale words per 1024
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box shows every
    CPU of the host but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                     # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = fh.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = int(fq.read()), int(fp.read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


PROFILE_SUMMARY = "profiles/r04_profile_summary.json"


def kernel_source_sha():
    """Hash of the sources the lossy kernels are compiled from. profiles/summarize.py stores it in the summary it
    writes; a summary measured on other kernel sources is stale and its counters are not reported."""
    import hashlib
    h = hashlib.sha256()
    for name in ("lossy_kernels.hip", "lossy_device.hpp", "lossy_kernels.hpp"):
        with open(os.path.join(ROOT, "flo_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def hbm_traffic(kname, clips, seconds):
    """HBM bytes per launch of the dominant kernel from the PMC passes of profiles/collect.sh (rocprofv3 cannot run
    inside this process). Reported only when the committed summary was measured on THESE kernel sources and on this
    workload; otherwise None (the counters would describe some other kernel)."""
    path = os.path.join(ROOT, PROFILE_SUMMARY)
    try:
        with open(path) as fh:
            doc = json.load(fh)
        if doc.get("kernel_source_sha") != kernel_source_sha():
            return None, f"{PROFILE_SUMMARY} was measured on other kernel sources (stale): not reported"
        if doc.get("clips_per_gpu") != clips or doc.get("clip_seconds") != seconds:
            return None, f"{PROFILE_SUMMARY} holds another workload: not reported"
        for name, e in doc["kernels"].items():
            if ("::" + kname + "_kernel") in name and "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
                return (int(e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]),
                        f"{PROFILE_SUMMARY} (kernel sources {doc['kernel_source_sha']}): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) "
                        f"and --pmc WRITE_SIZE, separate passes of this command")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips-per-gpu", type=int, default=10000)
    ap.add_argument("--clip-seconds", type=float, default=10.0)
    ap.add_argument("--quality", type=float, default=0.55)
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 chain kernel with one wave per channel, 2 frame-parallel kernels, 5 lock-step stereo chain kernel")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the exchange legs (gather, strong-scaling shard, table mode) with a one-rank communicator when N = 1 (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-clip", action="store_true")
    ap.add_argument("--no-lossless", action="store_true")
    ap.add_argument("--no-shard", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=200)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import flo_amd

    sr, ch = 44100, 2
    n_sf = int(round(args.clip_seconds * sr))
    n_il = n_sf * ch
    ctx = flo_amd.Context(local_rank)
    batch = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il] * args.clips_per_gpu, sr, ch, args.quality)
    batch.fill_synthetic(seed=0xF10A0D10, clip_id0=rank * args.clips_per_gpu)
    samples_per_step_rank = n_il * args.clips_per_gpu

    def all_ok(flag):
        """True only if `flag` holds on every rank (every rank must take the same branch)"""
        if dist is None:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def all_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    gather, gather_error = None, None
    exchanging = world > 1 or args.force_exchange
    if exchanging:
        # the exchange step lives behind the C ABI (flo_dist_*: RCCL directly, own stream, double-buffered); torch's
        # process group only carries the 128-byte rendezvous token from rank 0 to the others
        from flo_amd.dist import ID_BYTES, NativeGather, unique_id
        if dist is not None:
            tok = torch.zeros(ID_BYTES, dtype=torch.uint8, device=f"cuda:{local_rank}")
            if rank == 0:
                tok.copy_(torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8))
            dist.broadcast(tok, src=0)
            tok_bytes = bytes(tok.cpu().numpy().tobytes())
        else:
            tok_bytes = unique_id()
        try:
            gather = NativeGather(ctx, tok_bytes, rank, world, 0)
        except Exception as e:   # noqa: BLE001 - reported in the JSON line, never silent
            gather_error = f"rank {rank}: {e}"
        # if the communicator failed anywhere, nobody gathers (and the line says so)
        if not all_ok(gather is not None):
            gather = None
            gather_error = gather_error or "the RCCL communicator of flo_dist_create failed on another rank"
            print(f"[bench] exchange step disabled: {gather_error}", file=sys.stderr)

    exchange_check = None
    if gather is not None:
        # Before anything is timed: one small job (8 clips per rank, 3 steps) through the same communicator, checked end to
        # end - every rank's files as the root received them against a CRC the rank computed from its own fetch(). A
        # watchdog turns a hung exchange (a protocol bug would block in RCCL for ever) into a failed run with a message.
        import threading
        import zlib

        def _hung():
            print("[bench] exchange self-check hung for 180 s: aborting", file=sys.stderr, flush=True)
            os._exit(3)
        wd = threading.Timer(180.0, _hung)
        wd.daemon = True
        wd.start()
        try:
            small = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il // 10] * 8, sr, ch, args.quality)
            small.fill_synthetic(seed=0xF10A0D10, clip_id0=50_000_000 + rank * 8)
            for _ in range(3):
                small.encode(args.path)
                small.sync()
                gather.submit(small)
            gather.flush()
            mine = [small.fetch(i) for i in range(8)]
            crc = torch.tensor([zlib.crc32(b"".join(mine)), sum(len(f) for f in mine)], dtype=torch.int64, device=f"cuda:{local_rank}")
            crcs = torch.zeros(2 * world, dtype=torch.int64, device=f"cuda:{local_rank}")
            if dist is not None:
                dist.all_gather_into_tensor(crcs, crc)
            else:
                crcs.copy_(crc)
            check_ok = True
            if rank == 0:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                base, offs, sizes = gather.result()
                torch.cuda.synchronize()
                ok = True
                for r in range(world):
                    n_r = int(sizes[r])
                    raw = (ctypes.c_uint8 * max(n_r, 1))()
                    ok = ok and hip.hipMemcpy(raw, base + int(offs[r]), n_r, 2) == 0
                    blob, pos, files = bytes(raw[:n_r]), 0, []
                    while pos + 70 <= len(blob) and blob[pos:pos + 4] == b"FLO!":
                        n = 70 + int.from_bytes(blob[pos + 38:pos + 46], "little") + int.from_bytes(blob[pos + 46:pos + 54], "little")
                        files.append(blob[pos:pos + n])
                        pos += (n + 15) & ~15
                    ok = ok and len(files) == 8 and zlib.crc32(b"".join(files)) == int(crcs[2 * r]) and sum(map(len, files)) == int(crcs[2 * r + 1])
                exchange_check = ("every rank's files arrived byte for byte (8 clips per rank, 3 steps, CRC32 against the rank's own fetch)"
                                  if ok else "MISMATCH: gathered files differ from what the ranks encoded")
                check_ok = ok
                if not ok:
                    print("[bench] exchange self-check FAILED: gathered files differ from what the ranks encoded", file=sys.stderr)
            small.close()
        except Exception as e:   # noqa: BLE001 - reported, and fatal below
            check_ok = False
            exchange_check = f"self-check raised {type(e).__name__}: {e}"
            print(f"[bench] {exchange_check}", file=sys.stderr)
        wd.cancel()
        # A gather that does not deliver the ranks' bytes must not produce a scaling number: every rank learns of the
        # failure, rank 0 prints a line WITHOUT a value, and every rank exits non-zero.
        if not all_ok(check_ok):
            if rank == 0:
                print(json.dumps({"metric": "Msamples/s encoded (44.1k stereo, q=high)", "value": None, "unit": "Msamples/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                                  "error": "exchange self-check failed: " + (exchange_check or "a rank raised; see stderr"),
                                  "exchange_detail": {"self_check": exchange_check or "failed on another rank", "valid": False}}))
            sys.stdout.flush()
            os._exit(4)

    def step(with_gather=True):
        batch.encode(args.path)
        batch.sync()
        if gather is not None and with_gather:
            gather.submit(batch)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if gather is not None:
        gather.flush()
    encode_only_dt = None
    if gather is not None:
        # the same K steps without the exchange: what the GPUs do when nothing is gathered (never `value`)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        barrier()
        encode_only_dt = all_max(time.perf_counter() - t0)
    ctx.profile_reset()
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gather is not None:
        gather.flush()     # the last transfers (the gather of step k overlaps the encode of step k + 1)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    dt = all_max(dt)

    # the dominant kernel is whichever chain form ran (auto picks the three-wave pipeline for stereo batches)
    kname, k_ms, k_n = "lossy_frames", 0.0, 0
    for cand in ("lossy_chain2q", "lossy_chain2x", "lossy_chain3", "lossy_chain", "lossy_frames"):
        ms, n = ctx.profile_query(cand)
        if n:
            kname, k_ms, k_n = cand, ms, n
            break
    data_bytes = batch.data_bytes()
    gathered = None
    if gather is not None and rank == 0:
        _, g_offs, g_sizes = gather.result()
        gathered = [int(x) for x in g_sizes]

    strong, table_leg = None, None
    if gather is not None:
        from flo_amd.dist import contiguous_shard
        # --- BASELINE configs[3] read literally: ONE corpus of clips_per_gpu clips sharded across the ranks (strong
        # scaling), every step = encode of the rank's shard + the ONE gather of the finished files to rank 0
        c0, c1 = contiguous_shard(args.clips_per_gpu, rank, world)
        sb = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il] * (c1 - c0), sr, ch, args.quality)
        sb.fill_synthetic(seed=0xF10A0D10, clip_id0=c0)

        def sstep(with_gather):
            sb.encode(args.path)
            sb.sync()
            if with_gather:
                gather.submit(sb)
        for _ in range(max(1, args.warmup)):
            sstep(True)
        gather.flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sstep(False)
        barrier()
        s_enc = all_max(time.perf_counter() - t0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sstep(True)
        gather.flush()
        barrier()
        s_all = all_max(time.perf_counter() - t0)
        s_total = n_il * args.clips_per_gpu * args.steps
        strong = {"workload": f"ONE corpus of {args.clips_per_gpu} x {args.clip_seconds:g} s clips sharded over {world} rank(s) "
                              f"({c1 - c0} on rank 0), encode + one gather of the finished files per step",
                  "scaling": "strong", "value": round(s_total / s_all / 1e6, 1), "unit": "Msamples/s",
                  "ms_per_step": round(s_all / args.steps * 1e3, 4),
                  "encode_only": {"value": round(s_total / s_enc / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(s_enc / args.steps * 1e3, 4)}}
        if rank == 0:
            strong["bytes_per_rank_per_step"] = [int(x) for x in gather.result()[2]]
        sb.close()
        # --- second exchange mode, beside the gather and never instead of it: the files stay on their ranks and only the
        # (size, offset, CRC32) table is all-gathered - what the encode scales to when the root's links are not in the way
        def tstep():
            batch.encode(args.path)
            batch.sync()
            gather.table_submit(batch, args.clips_per_gpu)
        tstep()
        gather.table_flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tstep()
        gather.table_flush()
        barrier()
        t_all = all_max(time.perf_counter() - t0)
        tab = gather.table_result()
        table_leg = {"workload": "the weak-scaling workload (clips_per_gpu clips on every rank); files stay sharded, one ncclAllGather of "
                                 "24 bytes per clip names every file (owner, offset, size, CRC32)",
                     "scaling": "weak", "value": round(samples_per_step_rank * world * args.steps / t_all / 1e6, 1), "unit": "Msamples/s",
                     "ms_per_step": round(t_all / args.steps * 1e3, 4),
                     "table_bytes_per_rank_per_step": 8 * (1 + 3 * args.clips_per_gpu),
                     "files_named": sum(len(t[0]) for t in tab),
                     "bytes_named": sum(sum(t[0]) for t in tab)}

    if rank != 0:
        if gather is not None:
            gather.close()
        batch.close()
        ctx.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = samples_per_step_rank * world * args.steps
    value = total_samples / dt / 1e6
    out = {
        "metric": "Msamples/s encoded (44.1k stereo, q=high)",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "realtime_factor": round(value * 1e6 / (sr * ch), 1),
        "exchange": ("none (one rank)" if not exchanging else
                     ("every rank's finished files gathered to rank 0 each step: flo_dist_* (ncclAllGather of sizes + "
                      "grouped ncclSend/ncclRecv on its own stream, overlapping the next encode), inside the timed region"
                      if gather is not None else f"DISABLED, files stayed on their ranks: {gather_error}")),
        "config": {
            "workload": f"{args.clips_per_gpu} x {args.clip_seconds:g} s 44.1 kHz stereo clips per GPU, lossy quality=high "
                        f"(0.55)" + (": BASELINE configs[3], the 10 000-clip corpus, whole on each GPU (35 GB of PCM resident in "
                                     "HBM); its 1250-clip shard is timed under shard_1250, configs[1] (one 3-min clip) under "
                                     "single_clip_180s" if args.clips_per_gpu == 10000 and args.clip_seconds == 10.0 else ""),
            "clips_per_gpu": args.clips_per_gpu, "clip_seconds": args.clip_seconds, "quality": args.quality,
            "kernel_form": {"lossy_chain2q": "chain, one lock-step stereo transform wave + one quantiser-and-packer wave per clip", "lossy_chain2x": "chain, one lock-step stereo transform wave (with the quantiser) + one packer wave per clip", "lossy_chain3": "chain, three waves per stereo clip", "lossy_chain": "chain, one wave per channel", "lossy_frames": "frame-parallel"}[kname],
            "compressed_bytes_per_gpu": data_bytes,
        },
    }
    if exchanging:
        ex = {"reserved_cus": ctx.reserved_cus(), "self_check": exchange_check, "valid": gather is not None}
        if strong is not None:
            ex["strong_" + str(args.clips_per_gpu)] = strong
        if table_leg is not None:
            ex["table_exchange"] = table_leg
        if gather is not None and encode_only_dt is not None:
            enc_ms = encode_only_dt / args.steps * 1e3
            into_root = sum(gathered[r] for r in range(world) if r != 0)
            link_gbs = 77.0     # one xGMI link per direction, nominal (each peer has ONE link to the root)
            t_link_ms = max([gathered[r] for r in range(world) if r != 0] or [0]) / (link_gbs * 1e9) * 1e3
            ex.update({
                "encode_only": {"value": round(total_samples / args.steps * args.steps / encode_only_dt / 1e6, 1), "unit": "Msamples/s",
                                "ms_per_step": round(enc_ms, 4), "note": "the same K steps without flo_dist_gather_submit"},
                "with_gather_ms_per_step": round(dt / args.steps * 1e3, 4),
                "gather_cost_ms_per_step": round((dt - encode_only_dt) / args.steps * 1e3, 4),
                "bytes_per_rank_per_step": gathered,
                "bytes_into_root_per_step": into_root,
                "achieved_GBs_into_root": round(into_root / (dt / args.steps) / 1e9, 2),
                "bound": {"per_link_GBs_nominal": link_gbs, "slowest_peer_transfer_ms_at_link_peak": round(t_link_ms, 3),
                          "overlapped_efficiency_bound": round(min(1.0, enc_ms / max(enc_ms, t_link_ms)), 4),
                          "note": "weak-scaling efficiency cannot exceed encode / max(encode, transfer): every peer's files cross ONE "
                                  "xGMI link into the root; the value above is measured, this is the ceiling at nominal link rate"},
            })
        out["exchange_detail"] = ex
    if k_n:
        per_launch_s = k_ms / k_n / 1e3
        traffic, traffic_src = hbm_traffic(kname, args.clips_per_gpu, args.clip_seconds)
        achieved = ALG_BYTES_PER_SAMPLE * samples_per_step_rank / per_launch_s / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch",
            "traffic_source": traffic_src,
            "kernel_ms": round(k_ms / k_n, 4), "algorithmic_bytes_per_sample": round(ALG_BYTES_PER_SAMPLE, 4),
        }

    if not args.no_shard and world == 1 and args.clips_per_gpu != 1250:
        # the per-GPU share of the 8-way split of configs[3]: 1250 clips = 250 workgroups of five, one launch that
        # fills the 256 CUs exactly once (what round 1 reported as the headline)
        bs = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il] * 1250, sr, ch, args.quality)
        bs.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
        for _ in range(3):
            bs.encode(args.path)
            bs.sync()
        ctx.profile_reset()
        ctx.profile_enable(True)
        reps = 20
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(reps):
            bs.encode(args.path)
            bs.sync()
        d5 = (time.perf_counter() - t5) / reps
        ctx.profile_enable(False)
        sk_ms, sk_n = ctx.profile_query(kname)
        out["shard_1250"] = {"workload": "1250 x 10 s stereo clips, q=high: per-GPU shard of configs[3] split 8 ways",
                             "value": round(1250 * n_il / d5 / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(d5 * 1e3, 4)}
        if sk_n:
            out["shard_1250"]["kernel_ms"] = round(sk_ms / sk_n, 4)
            out["shard_1250"]["roofline_frac"] = round(ALG_BYTES_PER_SAMPLE * 1250 * n_il / (sk_ms / sk_n / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
        # decode of the same batch from its bitstreams in HBM into device memory (flo_batch_decode: lossy_decode_kernel).
        # Algorithmic bytes: the compressed files in + 4 B of f32 PCM out per sample.
        hops = (n_sf + 1024 + 1023) // 1024
        dec_n = 1250 * (hops - 1) * 1024 * ch
        dst = torch.empty(dec_n, dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        bs.decode_to(dst.data_ptr(), dst.numel())
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(3):
            bs.decode_to(dst.data_ptr(), dst.numel())
        ctx.profile_enable(False)
        dk_ms, dk_n = ctx.profile_query("lossy_decode")
        if dk_n:
            dms = dk_ms / dk_n
            dbytes = 4.0 * dec_n + bs.data_bytes()
            out["shard_1250"]["decode"] = {"kernel_ms": round(dms, 3), "value": round(1250 * n_il / dms / 1e3, 1), "unit": "Msamples/s",
                                           "hbm_algorithmic_GBs": round(dbytes / (dms / 1e3) / 1e9, 1),
                                           "frac_of_hbm_peak": round(dbytes / (dms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
        del dst
        bs.close()

    if not args.no_single_clip and world == 1:
        n180 = 180 * sr * ch
        b1 = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n180], sr, ch, args.quality)
        b1.fill_synthetic(seed=0xF10A0D10, clip_id0=10_000_000)
        for _ in range(3):
            b1.encode(2)
            b1.sync()
        reps = 20
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            b1.encode(2)
        b1.sync()
        d1 = (time.perf_counter() - t1) / reps
        out["single_clip_180s"] = {"workload": "BASELINE configs[1]: one 3-min 44.1 kHz stereo clip, q=high, frame-parallel kernels",
                                   "value": round(n180 / d1 / 1e6, 1), "unit": "Msamples/s", "ms": round(d1 * 1e3, 4),
                                   "realtime_factor": round(n180 / d1 / (sr * ch), 1)}
        b1.close()

    if not args.no_e2e and world == 1:
        # the drop-in calls on HOST buffers (what encode_to_flo / Encoder::encode bind to): pageable PCM in, malloc'ed
        # .flo files out, PCIe both ways included. Never `value`: the resident-batch rate above is the kernel's.
        import numpy as np
        # the 64 input clips: the product's own device generator (include/flo_synth.h) + one copy back to the host
        bg = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_il] * 64, sr, ch, args.quality)
        bg.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
        clips = [bg.download_pcm(i) for i in range(64)]
        bg.close()
        for _ in range(3):
            ctx.encode_lossy(clips[0], sr, ch, args.quality)
        t6 = time.perf_counter()
        for i in range(20):
            ctx.encode_lossy(clips[i], sr, ch, args.quality)
        d6 = (time.perf_counter() - t6) / 20
        for _ in range(2):   # (the first calls make the pinned buffers and touch the result pages: steady state is what is timed)
            ctx.encode_batch(flo_amd.MODE_LOSSY, clips, sr, ch, args.quality)
        best = None
        for _ in range(5):
            t7 = time.perf_counter()
            ctx.encode_batch(flo_amd.MODE_LOSSY, clips, sr, ch, args.quality)
            d7 = time.perf_counter() - t7
            best = d7 if best is None else min(best, d7)
        up_name, up_direct, up_ring = ctx.upload_path()
        out["e2e"] = {"note": "host buffers in, .flo files out (H2D + encode + D2H through the C ABI); not the headline",
                      "upload_path": {"large_uploads": up_name, "probe_pageable_direct_GBs": round(up_direct, 1), "probe_pinned_ring_GBs": round(up_ring, 1),
                                      "note": "chosen once per context by timing both on 24 MB; one clip always goes direct"},
                      "flo_encode_lossy_10s_clip_ms": round(d6 * 1e3, 4),
                      "flo_encode_lossy_10s_clip_Msamples_s": round(n_il / d6 / 1e6, 1),
                      "flo_encode_batch_64x10s_ms": round(best * 1e3, 3),
                      "flo_encode_batch_64x10s_Msamples_s": round(64 * n_il / best / 1e6, 1)}

    if not args.no_lossless and world == 1:
        # BASELINE configs[4] shape (96 kHz stereo, level 5) as a batch; bit-exactness is the tests' business, this is
        # the throughput of the ALPC + Rice kernels (no roofline claim: bound by 64-bit multiply-adds)
        lsr, lsec, lclips = 96000, 10, 128
        bl = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [lsr * lsec * ch] * lclips, lsr, ch, 5)
        bl.fill_synthetic(seed=0xF10A0D10, clip_id0=20_000_000)
        for _ in range(2):
            bl.encode(0)
            bl.sync()
        reps = 5
        t4 = time.perf_counter()
        for _ in range(reps):
            bl.encode(0)
            bl.sync()
        d4 = (time.perf_counter() - t4) / reps
        lsamples = lsr * lsec * ch * lclips
        lbytes = bl.data_bytes()
        # Rooflines of the lossless path. It is neither HBM- nor multiplier-bound: the level-5 search costs 35 i64
        # multiply-adds per sample (autocorrelation lags 0..8 + LPC orders 5..8 once each; lpc.rs:213-221,279-298), the
        # chip sustains 11.6 T of them per second (diag/mad64_rate.hip, 4 waves per SIMD), and the algorithmic HBM
        # traffic is 4 B in + the emitted bytes; both fractions are reported, the kernels are instruction-issue bound
        # (5.4 vector + 2 scalar wave-instructions per sample, profiles/).
        mads_per_sample, mad_peak = 35.0, 11.6e12
        out["lossless_96k"] = {"workload": f"{lclips} x {lsec} s 96 kHz stereo clips, lossless level 5 (BASELINE configs[4] shape)",
                               "value": round(lsamples / d4 / 1e6, 1), "unit": "Msamples/s", "ms": round(d4 * 1e3, 3),
                               "compressed_bytes": lbytes,
                               "roofline": {"bound": "instruction issue (int64 multiply-add search)",
                                            "i64_mads_per_sample": mads_per_sample,
                                            "achieved_Tmad_s": round(mads_per_sample * lsamples / d4 / 1e12, 3),
                                            "peak_Tmad_s_measured": mad_peak / 1e12,
                                            "frac_of_i64_mad_peak": round(mads_per_sample * lsamples / d4 / mad_peak, 4),
                                            "hbm_algorithmic_GBs": round((4.0 * lsamples + lbytes) / d4 / 1e9, 1),
                                            "frac_of_hbm_peak": round((4.0 * lsamples + lbytes) / d4 / 1e9 / HBM_PEAK_GBS, 4)}}
        # decode of the same batch from its files in HBM into device memory (flo_batch_decode): the parallel Rice
        # stages + the transposed f64 LPC recurrence (lldec_kernels.hip); the wrapper descriptions come from the
        # encoder's own records (one small read-back), nothing is parsed
        import torch
        dst = torch.empty(lsamples, dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        bl.decode_to(dst.data_ptr(), dst.numel())
        ctx.profile_enable(True)
        ctx.profile_reset()
        t5 = time.perf_counter()
        for _ in range(3):
            bl.decode_to(dst.data_ptr(), dst.numel())
        d5 = (time.perf_counter() - t5) / 3
        kms = sum(ctx.profile_query(k)[0] for k in ("ll_decode_parallel", "ll_decode", "ll_finish")) / 3
        ctx.profile_enable(False)
        out["lossless_96k"]["decode"] = {"value": round(lsamples / d5 / 1e6, 1), "unit": "Msamples/s", "ms": round(d5 * 1e3, 3),
                                         "kernels_ms": round(kms, 3), "kernels_Msamples_s": round(lsamples / kms / 1e3, 1),
                                         "hbm_algorithmic_GBs": round((8.0 * lsamples + lbytes) / (kms / 1e3) / 1e9, 1)}
        del dst
        bl.close()
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as O
        clips = [O.synth_clip(n_sf, ch, 0xF10A0D10, i) for i in range(args.cpu_clips)]
        t2 = time.perf_counter()
        nbytes = 0
        for c in clips:
            nbytes += len(O.encode_lossy(c, sr, ch, args.quality))
        d2 = time.perf_counter() - t2
        out["cpu_baseline"] = {
            "value": round(args.cpu_clips * n_il / d2 / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{args.cpu_clips} of the same synthetic 10 s stereo clips, q=0.55, oracle (C restatement of libflo, "
                      f"single thread like the reference; the Rust toolchain is not available here), {d2:.1f} s",
        }
        # SURVEY 8d also asks for "one clip per thread on all host cores": the same clips again, one oracle call per
        # pool thread (ctypes releases the GIL during the call)
        from concurrent.futures import ThreadPoolExecutor
        cores = host_cores()
        reps = max(1, min(4, (2 * cores + args.cpu_clips - 1) // args.cpu_clips))
        t3 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as pool:
            list(pool.map(lambda c: len(O.encode_lossy(c, sr, ch, args.quality)), clips * reps))
        d3 = time.perf_counter() - t3
        out["cpu_baseline"]["all_cores"] = {"value": round(reps * args.cpu_clips * n_il / d3 / 1e6, 3), "unit": "Msamples/s",
                                            "cores": cores, "sample": f"{reps * args.cpu_clips} clips, one per pool thread, {d3:.1f} s"}
    print(json.dumps(out))
    if gather is not None:
        gather.close()      # (before the context it was made on)
    batch.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
