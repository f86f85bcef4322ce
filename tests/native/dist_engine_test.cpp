// dist_engine_test.cpp — the multi-GPU exchange step's ordering logic (flo_amd/csrc/dist_engine.hpp: slot parity,
// deferred posting of the previous step's transfers, receive buffers that grow while peers are ready to send, zero-size
// payloads, a flush in the middle of a job) run with SEVERAL RANKS ON THE CPU: the same DistEngine template the product
// binds to HIP streams + RCCL is bound here to host memory and sockets. One process per rank (fork), a star of
// socketpairs around the root; messages are tagged so that a size of step k may overtake the payload of step k - 1 on
// the wire, as it does with RCCL's separate all-gather and send / receive operations.
//   usage: dist_engine_test WORLD STEPS   (exit code 0 = every step's gathered bytes were right on the root)
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <map>
#include <utility>
#include <vector>

#include "../../flo_amd/csrc/dist_engine.hpp"

namespace {

struct HostBatch {
    std::vector<std::vector<uint8_t>> files;
};

// what rank r hands over in step s: 1..3 "files" whose sizes grow with the step (so the buffers have to grow while the
// job runs) - except that rank 1 has nothing at all in step 2
HostBatch make_batch(int r, int s) {
    HostBatch b;
    if (r == 1 && s == 2) return b;
    const int n = 1 + (r + s) % 3;
    for (int i = 0; i < n; i++) {
        const size_t len = (size_t)(37 + 1000 * r + 977 * i) * (size_t)(1 + s * s) + (size_t)((r * 7 + s * 3 + i) % 16);
        std::vector<uint8_t> f(len);
        uint32_t x = 0x9E3779B9u * (uint32_t)(r + 1) + 0x85EBCA6Bu * (uint32_t)(s + 1) + (uint32_t)i;
        for (size_t k = 0; k < len; k++) {
            x = x * 1664525u + 1013904223u;
            f[k] = (uint8_t)(x >> 24);
        }
        b.files.push_back(std::move(f));
    }
    return b;
}
std::vector<uint8_t> packed(const HostBatch &b) {   // files back to back at 16-byte aligned offsets (flo_batch_pack_files)
    std::vector<uint8_t> out;
    for (const auto &f : b.files) {
        out.insert(out.end(), f.begin(), f.end());
        out.resize((out.size() + 15) & ~(size_t)15, 0);
    }
    return out;
}

struct SocketBackend {
    struct Buffer {
        std::vector<uint8_t> v;
    };
    int rank = 0, world = 1, root = 0;
    std::vector<int> fd;   // root: fd[r] = socket to rank r; peers: fd[root]
    // messages that arrived before they were asked for: (peer, type, slot) -> bytes
    std::map<std::pair<int, std::pair<int, int>>, std::vector<std::vector<uint8_t>>> stash;
    uint64_t mine[2] = {0, 0};
    std::vector<uint64_t> sizes[2];
    int reserve_calls = 0;

    static int wr(int f, const void *p, size_t n) {
        const uint8_t *b = (const uint8_t *)p;
        while (n) {
            ssize_t k = write(f, b, n);
            if (k < 0) {
                if (errno == EINTR) continue;
                return -1;
            }
            b += k;
            n -= (size_t)k;
        }
        return 0;
    }
    static int rd(int f, void *p, size_t n) {
        uint8_t *b = (uint8_t *)p;
        while (n) {
            ssize_t k = read(f, b, n);
            if (k <= 0) {
                if (k < 0 && errno == EINTR) continue;
                return -1;
            }
            b += k;
            n -= (size_t)k;
        }
        return 0;
    }
    int put(int peer, int type, int slot, const void *p, uint64_t n) {
        uint64_t h[3] = {(uint64_t)type, (uint64_t)slot, n};
        if (wr(fd[peer], h, sizeof h)) return 1;
        return n ? wr(fd[peer], p, n) : 0;
    }
    int get(int peer, int type, int slot, std::vector<uint8_t> &out) {
        auto key = std::make_pair(peer, std::make_pair(type, slot));
        for (;;) {
            auto it = stash.find(key);
            if (it != stash.end() && !it->second.empty()) {
                out = std::move(it->second.front());
                it->second.erase(it->second.begin());
                return 0;
            }
            uint64_t h[3];
            if (rd(fd[peer], h, sizeof h)) return 1;
            std::vector<uint8_t> m(h[2]);
            if (h[2] && rd(fd[peer], m.data(), h[2])) return 1;
            stash[std::make_pair(peer, std::make_pair((int)h[0], (int)h[1]))].push_back(std::move(m));
        }
    }

    int reserve(Buffer &b, size_t need) {
        if (b.v.size() >= need) return 0;
        reserve_calls++;
        std::vector<uint8_t>().swap(b.v);   // like the product: the old contents are gone
        b.v.assign(need + need / 4 + 64, 0xEE);
        return 0;
    }
    int payload_bytes(void *batch, uint64_t *need) {
        *need = packed(*(HostBatch *)batch).size();
        return 0;
    }
    int pack(void *batch, Buffer &dst, uint64_t *bytes) {
        const std::vector<uint8_t> p = packed(*(HostBatch *)batch);
        if (p.size() > dst.v.size()) return 2;
        if (!p.empty()) memcpy(dst.v.data(), p.data(), p.size());
        *bytes = p.size();
        return 0;
    }
    int wait_moved_before_pack(int) { return 0; }
    int mark_packed(int) { return 0; }
    int sizes_exchange(int s, uint64_t m) {
        mine[s] = m;
        if (rank != root) return put(root, 1, s, &m, 8);   // peers: on the wire at once; the vector is read in sizes_wait
        return 0;
    }
    int sizes_wait(int s, const uint64_t **out) {
        sizes[s].assign(world, 0);
        std::vector<uint8_t> m;
        if (rank == root) {
            sizes[s][root] = mine[s];
            for (int r = 0; r < world; r++)
                if (r != root) {
                    if (get(r, 1, s, m) || m.size() != 8) return 3;
                    memcpy(&sizes[s][r], m.data(), 8);
                }
            for (int r = 0; r < world; r++)
                if (r != root && put(r, 2, s, sizes[s].data(), 8 * (uint64_t)world)) return 3;
        } else {
            if (get(root, 2, s, m) || m.size() != 8 * (size_t)world) return 3;
            memcpy(sizes[s].data(), m.data(), m.size());
        }
        *out = sizes[s].data();
        return 0;
    }
    int comm_waits_for_pack(int) { return 0; }
    int copy_own(Buffer &dst, size_t off, Buffer &src, size_t n) {
        if (off + n > dst.v.size() || n > src.v.size()) return 4;
        memcpy(dst.v.data() + off, src.v.data(), n);
        return 0;
    }
    int group_begin() { return 0; }
    int recv(Buffer &dst, size_t off, size_t n, int peer) {
        std::vector<uint8_t> m;
        if (get(peer, 3, 0, m) || m.size() != n || off + n > dst.v.size()) return 5;
        memcpy(dst.v.data() + off, m.data(), n);
        return 0;
    }
    int send(Buffer &src, size_t n, int peer) { return n > src.v.size() ? 6 : put(peer, 3, 0, src.v.data(), n); }
    int group_end() { return 0; }
    int mark_moved(int) { return 0; }
    int drain() { return 0; }
};

int check_root(const flo::DistEngine<SocketBackend> &e, int world, int step) {
    if (e.res_slot != (step & 1)) return 10;
    for (int r = 0; r < world; r++) {
        const std::vector<uint8_t> want = packed(make_batch(r, step));
        if (e.res_size[r] != want.size()) return 11;
        if (e.res_off[r] % 256) return 12;
        if (want.size() && memcmp(e.recv[e.res_slot].v.data() + e.res_off[r], want.data(), want.size())) return 13;
    }
    return 0;
}

int run_rank(int rank, int world, int steps, std::vector<int> fd) {
    SocketBackend be;
    be.rank = rank;
    be.world = world;
    be.fd = std::move(fd);
    flo::DistEngine<SocketBackend> eng;
    eng.init(&be, rank, world, 0);
    for (int s = 0; s < steps; s++) {
        HostBatch b = make_batch(rank, s);
        int rc = eng.submit(&b);
        if (rc) return 100 + rc;
        // after submit(s) the transfers of step s - 1 have been posted (and, with this synchronous transport, have run)
        if (rank == 0 && s >= 1 && (rc = check_root(eng, world, s - 1))) return rc;
        if (s == steps / 2) {   // a flush in the middle of the job: everything pending is posted, the job goes on
            if ((rc = eng.flush())) return 200 + rc;
            if (rank == 0 && (rc = check_root(eng, world, s))) return 20 + rc;
        }
    }
    int rc = eng.flush();
    if (rc) return 300 + rc;
    if (rank == 0 && (rc = check_root(eng, world, steps - 1))) return 40 + rc;
    if (rank == 0 && steps >= 5 && be.reserve_calls < 4) return 60;   // the payloads grow: both slots' buffers had to, on both sides
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 2, steps = argc > 2 ? atoi(argv[2]) : 7;
    if (world < 1 || world > 8 || steps < 1) return 2;
    std::vector<int> root_fd(world, -1);
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 3;
        pid_t p = fork();
        if (p < 0) return 3;
        if (p == 0) {
            close(sv[0]);
            for (int f : root_fd)
                if (f >= 0) close(f);
            std::vector<int> fd(world, -1);
            fd[0] = sv[1];
            _exit(run_rank(r, world, steps, fd));
        }
        close(sv[1]);
        root_fd[r] = sv[0];
        kids.push_back(p);
    }
    int rc = run_rank(0, world, steps, root_fd);
    for (int f : root_fd)
        if (f >= 0) close(f);
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = rc ? rc : 90 + (WIFEXITED(st) ? WEXITSTATUS(st) : 0);
    }
    if (rc) fprintf(stderr, "dist_engine_test: world %d steps %d failed with code %d\n", world, steps, rc);
    else printf("dist_engine_test: world %d, %d steps ok\n", world, steps);
    return rc;
}
