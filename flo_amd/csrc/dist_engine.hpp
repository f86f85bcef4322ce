// dist_engine.hpp — the ordering logic of the multi-GPU exchange step (flo_dist_* in include/flo_hip.h), separated from
// what moves the bytes. One process per rank; per step every rank packs its finished .flo files into one of two send
// buffers, the ranks all-gather the packed sizes, and every peer sends its buffer to the root, which receives them side
// by side (256-byte aligned) into one of two receive buffers. The logic is the pipelining:
//   submit(step k): pack into slot k & 1 | exchange sizes of step k (asynchronous) | post the transfers of step k - 1,
//                   whose sizes have arrived while step k was being encoded
//   flush():        post whatever is still pending and drain
// so that the transfer of step k overlaps the encode of step k + 1 and the host never waits for the device per step.
//
// The Backend supplies the mechanisms (queues, buffers, collectives). The product instantiates the engine with HIP
// streams + RCCL (flo_api.cpp: RcclBackend); tests/native/dist_engine_test.cpp instantiates the SAME template with host
// memory and sockets and runs it with several ranks on the CPU. Backend concept (all int returns: 0 = ok):
//   struct Buffer;                                         a growable byte buffer the transfers can address
//   int  reserve(Buffer &b, size_t need);                  grow b to >= need bytes (may wait for work still using the old one)
//   int  pack(void *batch, Buffer &dst, uint64_t *bytes);  pack the batch's files into dst on the compute queue (async)
//   int  payload_bytes(void *batch, uint64_t *need);       upper bound of what pack() writes
//   int  wait_moved_before_pack(int slot);                 compute queue waits for slot's last transfers (device-side)
//   int  mark_packed(int slot);                            compute queue: "slot is packed"
//   int  sizes_exchange(int slot, uint64_t mine);          comm queue: all-gather of one u64 per rank (async) ...
//   int  sizes_wait(int slot, const uint64_t **sizes);     ... and its result on the host (waits if it has not arrived)
//   int  comm_waits_for_pack(int slot);                    comm queue waits for mark_packed(slot)
//   int  copy_own(Buffer &dst, size_t off, Buffer &src, size_t n);   root: its own files, on the comm queue
//   int  group_begin(); int recv(Buffer &dst, size_t off, size_t n, int peer); int send(Buffer &src, size_t n, int peer);
//   int  group_end();                                      the grouped point-to-point transfers, on the comm queue
//   int  mark_moved(int slot);                             comm queue: "slot's transfers are done"
//   int  drain();                                          host waits for the comm queue
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace flo {

template <class Backend>
struct DistEngine {
    typedef typename Backend::Buffer Buffer;
    Backend *be = nullptr;
    int rank = 0, world = 1, root = 0;
    Buffer send[2], recv[2];                  // per slot (step parity); recv on the root only
    uint64_t send_bytes[2] = {0, 0};
    bool posted[2] = {true, true};            // the slot's transfers have been posted (nothing pending)
    bool used[2] = {false, false};
    uint64_t submits = 0;
    // result of the most recently posted step (root)
    std::vector<uint64_t> res_off, res_size;
    int res_slot = -1;

    void init(Backend *b, int rank_, int world_, int root_) {
        be = b;
        rank = rank_;
        world = world_;
        root = root_;
    }

    // post the point-to-point transfers of slot s (its sizes have arrived on the host, or are awaited here)
    int post(int s) {
        if (posted[s]) return 0;
        const uint64_t *sz = nullptr;
        int rc = be->sizes_wait(s, &sz);   // exchanged a whole step ago: no stall in steady state
        if (rc) return rc;
        if (rank == root) {
            uint64_t total = 0;
            res_off.assign(world, 0);
            res_size.assign(sz, sz + world);
            for (int r = 0; r < world; r++) {
                res_off[r] = total;
                total += (sz[r] + 255) & ~(uint64_t)255;
            }
            if ((rc = be->reserve(recv[s], total ? total : 256))) return rc;
            if (sz[root] && (rc = be->copy_own(recv[s], res_off[root], send[s], sz[root]))) return rc;
            if ((rc = be->group_begin())) return rc;
            for (int r = 0; r < world; r++)
                if (r != root && sz[r] && (rc = be->recv(recv[s], res_off[r], sz[r], r))) {
                    be->group_end();   // never leave the group open behind an error
                    return rc;
                }
            if ((rc = be->group_end())) return rc;
            res_slot = s;
        } else if (sz[rank]) {
            if ((rc = be->group_begin())) return rc;
            if ((rc = be->send(send[s], sz[rank], root))) {
                be->group_end();
                return rc;
            }
            if ((rc = be->group_end())) return rc;
        }
        if ((rc = be->mark_moved(s))) return rc;   // the slot's send buffer may be packed into again behind this
        posted[s] = true;
        return 0;
    }

    int submit(void *batch) {
        const int s = (int)(submits & 1), prev = s ^ 1;
        // 1. this slot's previous transfers (two submits ago) were posted one submit ago; its send buffer is free once
        //    they have run: the pack waits for that on the device, the host does not
        int rc = post(s);   // (only pending when submits were skipped; normally a no-op)
        if (rc) return rc;
        // 2. pack this batch's finished files into the slot's send buffer (compute queue)
        uint64_t need = 0;
        if ((rc = be->payload_bytes(batch, &need))) return rc;
        if ((rc = be->reserve(send[s], need ? need : 16))) return rc;
        if (used[s] && (rc = be->wait_moved_before_pack(s))) return rc;
        if ((rc = be->pack(batch, send[s], &send_bytes[s]))) return rc;
        if ((rc = be->mark_packed(s))) return rc;
        // 3. sizes: all-gather on the comm queue, result to the host asynchronously
        if ((rc = be->sizes_exchange(s, send_bytes[s]))) return rc;
        if ((rc = be->comm_waits_for_pack(s))) return rc;   // the payload transfers (posted next submit) read the packed buffer
        posted[s] = false;
        used[s] = true;
        submits++;
        // 4. the transfers of the previous submit: their sizes arrived during the step that has just been encoded
        return post(prev);
    }

    int flush() {
        const int last = (int)((submits + 1) & 1);   // slot of the most recent submit
        int rc = post(last ^ 1);
        if (rc == 0) rc = post(last);
        if (rc) return rc;
        return be->drain();
    }
};

}  // namespace flo
