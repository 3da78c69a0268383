// sphmi_shm.h — host shared-memory transport for the rank mode of the slab driver (SPHMI_TRANSPORT=shm).
//
// sphmi_create_rank puts the peers of a slab behind RCCL.  RCCL refuses two ranks on one device, so on a box with fewer
// GPUs than ranks — the one-GPU test box above all — the SAME rank-mode driver (one slab per process, counts and index
// lists negotiated between processes, matching messages for cancelled steps, the collective rebuild) runs with this
// transport instead: messages and reductions are staged through the host and a POSIX shared-memory segment that all
// ranks of the node map.  It is a bring-up and test transport: every exchange synchronises its stream, so nothing
// overlaps and a step costs a host round trip; results are bit-identical to the RCCL path by construction (the same
// bytes travel, the reductions are integer maxima / sums).
//
// Segment layout (all-zero is the valid initial state, so creation needs no hand-shake):
//   header      barrier counter + sense, attach counter
//   reduce[w]   one slot of kReduceWords int64 per rank (host allreduce: write own slot, barrier, read all, barrier)
//   chan[w][2]  single-producer single-consumer byte rings: chan[r][0] carries r → r−1, chan[r][1] carries r → r+1
// Every wait has a deadline (SPHMI_SHM_TIMEOUT seconds, default 120): a dead peer is an error, not a hang.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace sphmi {

struct ShmError : std::runtime_error { using std::runtime_error::runtime_error; };

class ShmWorld {
public:
    static constexpr size_t kReduceWords = 4096;
    static constexpr size_t kRingBytes = 4u << 20;
    struct Header { std::atomic<int> bar_count, bar_sense, attached, pad; };
    struct Ring { std::atomic<uint64_t> head; char pad0[56]; std::atomic<uint64_t> tail; char pad1[56]; char data[kRingBytes]; };
    enum Op { SUM, MAX, MAXU };

    // one posted half of a neighbour message: to_right / from_right select the peer, p/bytes the host buffer
    struct Xfer { bool send; int side; char* p; size_t bytes; size_t done; };

    ShmWorld(const void* unique_id, size_t id_bytes, int rank, int world) : rank_(rank), world_(world) {
        uint64_t h = 1469598103934665603ull;                                  // FNV-1a of the launcher's unique id
        for (size_t i = 0; i < id_bytes; ++i) { h ^= ((const unsigned char*)unique_id)[i]; h *= 1099511628211ull; }
        char nm[64]; snprintf(nm, sizeof nm, "/sphmi_%016llx_w%d", (unsigned long long)h, world);
        name_ = nm;
        bytes_ = sizeof(Header) + (size_t)world * kReduceWords * 8 + (size_t)world * 2 * sizeof(Ring);
        if (const char* t = getenv("SPHMI_SHM_TIMEOUT")) timeout_s_ = atof(t) > 0 ? atof(t) : timeout_s_;
        const int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
        if (fd < 0) throw ShmError(std::string("shm transport: shm_open failed for ") + nm);
        if (ftruncate(fd, (off_t)bytes_) != 0) { close(fd); throw ShmError("shm transport: ftruncate failed (is /dev/shm full?)"); }
        base_ = (char*)mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base_ == (char*)MAP_FAILED) { base_ = nullptr; throw ShmError("shm transport: mmap failed"); }
        hdr_ = (Header*)base_;
        red_ = (int64_t*)(base_ + sizeof(Header));
        rings_ = (Ring*)(base_ + sizeof(Header) + (size_t)world * kReduceWords * 8);
        hdr_->attached.fetch_add(1);
        try { barrier(); } catch (...) { shm_unlink(nm); throw; }
        if (rank == 0) shm_unlink(nm);                                        // everyone has it mapped: no name left behind
    }
    ~ShmWorld() { if (base_) munmap(base_, bytes_); }
    ShmWorld(const ShmWorld&) = delete;
    ShmWorld& operator=(const ShmWorld&) = delete;

    int rank() const { return rank_; }
    int world() const { return world_; }

    void barrier() {
        const auto t0 = now();
        sense_ ^= 1;
        if (hdr_->bar_count.fetch_add(1) + 1 == world_) { hdr_->bar_count.store(0); hdr_->bar_sense.store(sense_); return; }
        while (hdr_->bar_sense.load() != sense_) relax(t0, "barrier");
    }

    // v[0..n) ← reduction over all ranks of their v (every rank passes the same n)
    void allreduce(int64_t* v, size_t n, Op op) {
        for (size_t o = 0; o < n; o += kReduceWords) {
            const size_t m = std::min(kReduceWords, n - o);
            memcpy(red_ + (size_t)rank_ * kReduceWords, v + o, m * 8);
            barrier();
            for (size_t k = 0; k < m; ++k) {
                int64_t acc = red_[k];
                for (int r = 1; r < world_; ++r) {
                    const int64_t u = red_[(size_t)r * kReduceWords + k];
                    if (op == SUM) acc += u;
                    else if (op == MAX) acc = u > acc ? u : acc;
                    else acc = (uint64_t)u > (uint64_t)acc ? u : acc;
                }
                v[o + k] = acc;
            }
            barrier();
        }
    }

    // all posted halves of one phase: sends to / receives from the left (side 0) and right (side 1) neighbour.  Both
    // ends post the same byte counts in the same order per direction (the driver negotiated them), so the rings need no
    // message headers.  Sends and receives progress together: a message longer than a ring cannot dead-lock.
    void exchange(std::vector<Xfer>& x) {
        const auto t0 = now();
        for (;;) {
            bool pending = false, moved = false;
            for (Xfer& t : x) {
                if (t.done == t.bytes) continue;
                // rings are strictly in-order per direction: only the first unfinished transfer of a direction may move
                bool first = true;
                for (Xfer& u : x) { if (&u == &t) break; if (u.send == t.send && u.side == t.side && u.done != u.bytes) { first = false; break; } }
                if (first) {
                    const size_t n = t.send ? push(out_ring(t.side), t.p + t.done, t.bytes - t.done)
                                            : pull(in_ring(t.side), t.p + t.done, t.bytes - t.done);
                    t.done += n; moved |= n != 0;
                }
                pending |= t.done != t.bytes;
            }
            if (!pending) return;
            if (!moved) relax(t0, "neighbour exchange");
        }
    }

private:
    using clock = std::chrono::steady_clock;
    static clock::time_point now() { return clock::now(); }
    void relax(clock::time_point t0, const char* what) {
        sched_yield();
        if (std::chrono::duration<double>(now() - t0).count() > timeout_s_)
            throw ShmError(std::string("shm transport: rank ") + std::to_string(rank_) + " waited " + std::to_string((int)timeout_s_) +
                           " s in " + what + " — a peer process is gone or out of step");
    }
    Ring& out_ring(int side) { return rings_[(size_t)rank_ * 2 + side]; }
    // what the left neighbour sends to its RIGHT arrives from my left, and the other way round
    Ring& in_ring(int side) { return side == 0 ? rings_[(size_t)(rank_ - 1) * 2 + 1] : rings_[(size_t)(rank_ + 1) * 2 + 0]; }
    static size_t push(Ring& r, const char* p, size_t n) {
        const uint64_t h = r.head.load(std::memory_order_relaxed), t = r.tail.load(std::memory_order_acquire);
        const size_t room = kRingBytes - (size_t)(h - t), m = std::min(room, n);
        if (!m) return 0;
        const size_t at = (size_t)(h % kRingBytes), first = std::min(m, kRingBytes - at);
        memcpy(r.data + at, p, first); memcpy(r.data, p + first, m - first);
        r.head.store(h + m, std::memory_order_release);
        return m;
    }
    static size_t pull(Ring& r, char* p, size_t n) {
        const uint64_t t = r.tail.load(std::memory_order_relaxed), h = r.head.load(std::memory_order_acquire);
        const size_t have = (size_t)(h - t), m = std::min(have, n);
        if (!m) return 0;
        const size_t at = (size_t)(t % kRingBytes), first = std::min(m, kRingBytes - at);
        memcpy(p, r.data + at, first); memcpy(p + first, r.data, m - first);
        r.tail.store(t + m, std::memory_order_release);
        return m;
    }

    int rank_, world_, sense_ = 0;
    double timeout_s_ = 120.0;
    std::string name_;
    size_t bytes_ = 0;
    char* base_ = nullptr;
    Header* hdr_ = nullptr;
    int64_t* red_ = nullptr;
    Ring* rings_ = nullptr;
};

}  // namespace sphmi
