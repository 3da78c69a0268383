// mock_rccl — a test double for librccl.so, loaded through $SPHMI_RCCL_LIB (csrc/sphmi_multi.h, Rccl::get).
//
// TEST INFRASTRUCTURE.  Nothing under sphexample_amd/ links, loads or names this file; tests/test_mock_rccl*.py point
// $SPHMI_RCCL_LIB at the built library so that the RCCL branch of the slab driver — the halo ncclSend/ncclRecv groups, the
// neighbour counts, the two communicators, the per-step MAX-allreduce, the id hand-over of the second communicator, the mailbox
// handle table — runs with 2-4 ranks on ONE GPU, where the real library refuses ("two ranks on one device").
//
// It exports exactly the eleven symbols the driver binds (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommDestroy,
// ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclAllReduce, ncclGetErrorString, ncclGetLastError) with the NCCL 2.x
// signatures, moves the bytes through a POSIX shared-memory segment named after the unique id (device -> host -> segment ->
// host -> device, staged copies on the stream each call names), and CHECKS what a real RCCL would answer with a hang:
//   * every ncclRecv meets a ncclSend of the SAME byte count and datatype, in the same position of the pair's message sequence
//     and in the same group of the pair's group sequence (rank a's k-th group that talks to b is rank b's k-th group that talks to a);
//   * a send completes only when its receiver has taken the last byte (rendezvous: no program may rely on eager buffering);
//   * every rank enters the k-th collective of a communicator with the same count, datatype and reduction;
//   * groups are closed: ncclGroupEnd without a start, a communicator destroyed inside an open group or with a message nobody
//     received, a process that ends with an open group;
//   * every wait has a deadline ($MOCK_RCCL_TIMEOUT seconds, default 60): the failing call returns ncclSystemError and
//     ncclGetLastError says which operation of which rank was waiting for which peer.
// A violation poisons the segment: every rank's next call fails with the first violation's text instead of waiting.
// Default mode: an operation has completed when the call (or the closing ncclGroupEnd) returns — STRONGER than RCCL, so stream-ordering
// mistakes of the caller stay hidden; message-list mistakes do not.
// $MOCK_RCCL_ASYNC=1: RCCL's own timing.  The call only records an event on the stream it names and queues a wait on a signal word
// (hipStreamWaitValue32) behind it; a proxy thread waits for the event — everything the caller queued BEFORE the call — reads the send
// buffers THEN, moves the bytes, writes the receive buffers and releases the signal: the caller's later work on that stream sees the
// data, work it queued on OTHER streams without an event does not wait.  A send buffer refilled too early, a receive buffer read
// without waiting for its stream: both become wrong bytes that the parity comparison of the tests sees
// (stream_order_probe.py makes both mistakes on purpose).  $MOCK_RCCL_ASYNC_DELAY_US: the proxy sleeps that long before it reads the
// send buffers — RCCL's kernels are not instant either.  A wait parks a hardware queue: run with GPU_MAX_HW_QUEUES=24 so that the proxy's
// copies and the caller's other streams do not share one with a parked stream.
//
// Build: hipcc (host code only) -shared -fPIC mock_rccl.cpp -o libmockrccl.so     (tests/mock_rccl/build.py)

#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}

namespace {

constexpr int kMaxWorld = 16;
constexpr size_t kChunk = 64 << 10, kSlots = 4, kRedMax = 256 << 10;
constexpr uint64_t kMagic = 0x6d6f636b5243434cull;     // "mockRCCL"

struct alignas(64) Chan {                               // one per ordered pair (src, dst): single producer, single consumer
    std::atomic<uint64_t> head, tail;                   // chunks published / consumed
    char pad[48];
    char slot[kSlots][kChunk];
};
struct MsgHdr { uint64_t magic, bytes, dtype, pair_epoch, index; };
struct alignas(64) RedSlot {
    std::atomic<uint64_t> posted;                       // number of the last collective this rank posted
    uint64_t count[2], dtype[2], op[2], epoch[2][kMaxWorld];
    char data[2][kRedMax];
};
struct Seg {
    std::atomic<uint32_t> state;                        // 0 zero-filled, 1 being initialised, 2 ready
    uint32_t world;
    std::atomic<uint32_t> arrived, departed, poisoned;
    char poison[512];
    RedSlot red[kMaxWorld];
    Chan chan[kMaxWorld * kMaxWorld];
};

struct Stats {
    std::atomic<uint64_t> comms{0}, groups{0}, sends{0}, recvs{0}, send_bytes{0}, recv_bytes{0}, allreduces{0}, violations{0}, max_send{0},
        open_comms{0}, timeouts{0}, proxied{0};
    std::mutex mu;
    std::set<const void*> p2p_streams, coll_streams;
} g_stats;

double timeout_s() {
    static double t = [] { const char* e = getenv("MOCK_RCCL_TIMEOUT"); double v = e ? atof(e) : 0.0; return v > 0 ? v : 60.0; }();
    return t;
}
using Clock = std::chrono::steady_clock;
// $MOCK_RCCL_HOST_BUFFERS=1: the buffers handed to ncclSend / ncclRecv / ncclAllReduce are HOST memory and streams are ignored —
// the mock checks its own checking in CPU processes (tests/test_mock_rccl.py, no GPU needed).
bool host_mode() {
    static bool h = [] { const char* e = getenv("MOCK_RCCL_HOST_BUFFERS"); return e && atoi(e) != 0; }();
    return h;
}

struct Comm {
    Seg* seg = nullptr;
    size_t seg_bytes = 0;
    std::shared_ptr<void> keep;                         // the mapping, shared by the communicators of one ncclCommInitAll
    int rank = 0, world = 1, device = 0, serial = 0;
    uint64_t coll_seq = 0;
    uint64_t epoch[kMaxWorld] = {};                     // groups so far that talked to peer q (collectives talk to everybody)
    uint64_t n_sent[kMaxWorld] = {}, n_recvd[kMaxWorld] = {};
    std::string last_error;
    bool dead = false;
    std::atomic<int> async_error{0};                    // MOCK_RCCL_ASYNC: the result of a batch that failed on the proxy thread; every later call returns it
    std::atomic<uint64_t> queued{0}, retired{0};        // … batches of this communicator handed to / finished by the proxy
};
std::mutex g_comm_mu;
std::set<Comm*> g_live;
thread_local std::string t_last_error;                  // ncclGetLastError(NULL)

enum Kind { SEND, RECV, ALLREDUCE };
struct Op {
    Kind kind; Comm* c; int peer; const void* sbuf; void* rbuf; size_t count; ncclDataType_t dt; ncclRedOp_t rop; hipStream_t stream;
    size_t bytes = 0;
    std::vector<char> host;
    // progress
    int stage = 0; size_t done_bytes = 0; uint64_t end_head = 0; bool finished = false; uint64_t coll_no = 0; uint64_t epoch_tag = 0; uint64_t index = 0;
    std::vector<uint64_t> epochs;                           // ALLREDUCE: the group count with every peer when the call was made (the proxy runs later)
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t dt_size(ncclDataType_t d) { switch (d) { case ncclInt8: case ncclUint8: return 1; case ncclInt32: case ncclUint32: return 4; default: return 8; } }
const char* kind_name(Kind k) { return k == SEND ? "ncclSend" : k == RECV ? "ncclRecv" : "ncclAllReduce"; }

ncclResult_t violation(Comm* c, ncclResult_t code, const std::string& text) {
    std::string m = "[mock-rccl] VIOLATION";
    if (c) { char b[64]; snprintf(b, sizeof b, " (communicator %d, rank %d of %d)", c->serial, c->rank, c->world); m += b; }
    m += ": " + text;
    fprintf(stderr, "%s\n", m.c_str());
    fflush(stderr);
    g_stats.violations += 1;
    t_last_error = m;
    if (c) {
        c->last_error = m;
        if (c->seg) {
            uint32_t z = 0;
            if (c->seg->poisoned.compare_exchange_strong(z, 1u)) { snprintf(c->seg->poison, sizeof c->seg->poison, "%s", m.c_str()); c->seg->poisoned.store(2u, std::memory_order_release); }
        }
    }
    return code;
}
bool poisoned(Comm* c, std::string& why) {
    if (!c->seg) return false;
    const uint32_t p = c->seg->poisoned.load(std::memory_order_acquire);
    if (!p) return false;
    why = p == 2 ? std::string(c->seg->poison) : std::string("[mock-rccl] a peer reported a violation");
    return true;
}

struct Mapping { void* p; size_t n; ~Mapping() { if (p) munmap(p, n); } };

std::string seg_name(const ncclUniqueId& id) {
    char b[80]; char* w = b + snprintf(b, sizeof b, "/mockrccl_");
    for (int i = 0; i < 16; ++i) w += snprintf(w, 4, "%02x", (unsigned char)id.internal[i] ^ (unsigned char)id.internal[16 + i]);
    return b;
}

template <class T> void reduce_t(T* acc, const T* in, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
            case ncclSum: acc[i] = (T)(acc[i] + in[i]); break;
            case ncclProd: acc[i] = (T)(acc[i] * in[i]); break;
            case ncclMax: acc[i] = acc[i] > in[i] ? acc[i] : in[i]; break;
            case ncclMin: acc[i] = acc[i] < in[i] ? acc[i] : in[i]; break;
        }
    }
}
void reduce(void* acc, const void* in, size_t n, ncclDataType_t dt, ncclRedOp_t op) {
    switch (dt) {
        case ncclInt8: reduce_t((int8_t*)acc, (const int8_t*)in, n, op); break;
        case ncclUint8: reduce_t((uint8_t*)acc, (const uint8_t*)in, n, op); break;
        case ncclInt32: reduce_t((int32_t*)acc, (const int32_t*)in, n, op); break;
        case ncclUint32: reduce_t((uint32_t*)acc, (const uint32_t*)in, n, op); break;
        case ncclInt64: reduce_t((int64_t*)acc, (const int64_t*)in, n, op); break;
        case ncclUint64: reduce_t((uint64_t*)acc, (const uint64_t*)in, n, op); break;
    }
}

#define HIPQ(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { return violation(op.c, ncclUnhandledCudaError, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)

std::string describe(const Op& o) {
    char b[256];
    if (o.kind == ALLREDUCE) snprintf(b, sizeof b, "%s #%llu of rank %d (count %zu, datatype %d, op %d)", kind_name(o.kind), (unsigned long long)o.coll_no, o.c->rank, o.count, (int)o.dt, (int)o.rop);
    else snprintf(b, sizeof b, "%s of rank %d %s rank %d (%zu bytes, message #%llu of the pair, %zu bytes moved)", kind_name(o.kind), o.c->rank, o.kind == SEND ? "to" : "from", o.peer, o.bytes,
                  (unsigned long long)o.index, o.done_bytes);
    return b;
}

// one non-blocking step of an operation; returns false on a violation (text in `why`)
bool step(Op& o, std::string& why) {
    Comm* c = o.c;
    Seg* s = c->seg;
    if (o.kind == SEND) {
        Chan& ch = s->chan[c->rank * kMaxWorld + o.peer];
        for (;;) {
            const uint64_t head = ch.head.load(std::memory_order_relaxed);
            if (o.stage == 2) { if (ch.tail.load(std::memory_order_acquire) >= o.end_head) o.finished = true; return true; }
            if (head - ch.tail.load(std::memory_order_acquire) >= kSlots) return true;              // ring full: the receiver has to take something first
            char* dst = ch.slot[head % kSlots];
            if (o.stage == 0) {
                MsgHdr h{kMagic, (uint64_t)o.bytes, (uint64_t)o.dt, o.epoch_tag, o.index};
                memcpy(dst, &h, sizeof h);
                o.stage = o.bytes ? 1 : 2;
            } else {
                const size_t n = std::min(kChunk, o.bytes - o.done_bytes);
                memcpy(dst, o.host.data() + o.done_bytes, n);
                o.done_bytes += n;
                if (o.done_bytes == o.bytes) o.stage = 2;
            }
            ch.head.store(head + 1, std::memory_order_release);
            if (o.stage == 2) o.end_head = head + 1;
        }
    }
    if (o.kind == RECV) {
        Chan& ch = s->chan[o.peer * kMaxWorld + c->rank];
        for (;;) {
            const uint64_t tail = ch.tail.load(std::memory_order_relaxed);
            if (ch.head.load(std::memory_order_acquire) == tail) return true;                       // nothing published yet
            const char* src = ch.slot[tail % kSlots];
            if (o.stage == 0) {
                MsgHdr h; memcpy(&h, src, sizeof h);
                char b[400];
                if (h.magic != kMagic) { why = "ncclRecv found something that is not a message header in the channel (an earlier message was longer than its receive)"; return false; }
                if (h.bytes != o.bytes || h.dtype != (uint64_t)o.dt) {
                    snprintf(b, sizeof b, "message #%llu from rank %d to rank %d: ncclSend posted %llu bytes (datatype %llu), the matching ncclRecv expects %zu bytes (datatype %d)",
                             (unsigned long long)o.index, o.peer, c->rank, (unsigned long long)h.bytes, (unsigned long long)h.dtype, o.bytes, (int)o.dt);
                    why = b; return false;
                }
                if (h.index != o.index) { snprintf(b, sizeof b, "message from rank %d to rank %d: the sender counts it as #%llu of the pair, the receiver as #%llu", o.peer, c->rank, (unsigned long long)h.index, (unsigned long long)o.index); why = b; return false; }
                if (h.pair_epoch != o.epoch_tag) {
                    snprintf(b, sizeof b, "message #%llu from rank %d to rank %d was sent in the sender's group #%llu with this peer and is received in the receiver's group #%llu with that peer: the two ranks' call sequences differ",
                             (unsigned long long)o.index, o.peer, c->rank, (unsigned long long)h.pair_epoch, (unsigned long long)o.epoch_tag);
                    why = b; return false;
                }
                o.stage = 1;
            } else {
                const size_t n = std::min(kChunk, o.bytes - o.done_bytes);
                memcpy(o.host.data() + o.done_bytes, src, n);
                o.done_bytes += n;
            }
            ch.tail.store(tail + 1, std::memory_order_release);
            if (o.stage == 1 && o.done_bytes == o.bytes) { o.finished = true; return true; }
        }
    }
    // ALLREDUCE: stage 0 post, stage 1 wait for everybody and reduce in rank order
    const int par = (int)(o.coll_no & 1);
    RedSlot& mine = s->red[c->rank];
    if (o.stage == 0) {
        mine.count[par] = o.count; mine.dtype[par] = (uint64_t)o.dt; mine.op[par] = (uint64_t)o.rop;
        for (int q = 0; q < kMaxWorld; ++q) mine.epoch[par][q] = o.epochs[q];
        memcpy(mine.data[par], o.host.data(), o.bytes);
        mine.posted.store(o.coll_no, std::memory_order_release);
        o.stage = 1;
    }
    for (int q = 0; q < c->world; ++q) if (s->red[q].posted.load(std::memory_order_acquire) < o.coll_no) return true;
    std::vector<char> acc(o.bytes);
    for (int q = 0; q < c->world; ++q) {
        const RedSlot& r = s->red[q];
        if (r.count[par] != o.count || r.dtype[par] != (uint64_t)o.dt || r.op[par] != (uint64_t)o.rop) {
            char b[300];
            snprintf(b, sizeof b, "collective #%llu: rank %d entered ncclAllReduce with count %llu / datatype %llu / op %llu, rank %d with count %zu / datatype %d / op %d",
                     (unsigned long long)o.coll_no, q, (unsigned long long)r.count[par], (unsigned long long)r.dtype[par], (unsigned long long)r.op[par], c->rank, o.count, (int)o.dt, (int)o.rop);
            why = b; return false;
        }
        if (q != c->rank && r.epoch[par][c->rank] != o.epochs[q]) {
            char b[300];
            snprintf(b, sizeof b, "collective #%llu: rank %d has had %llu groups with rank %d on this communicator, rank %d has had %llu with rank %d: point-to-point and collective calls are interleaved differently on the two ranks",
                     (unsigned long long)o.coll_no, q, (unsigned long long)r.epoch[par][c->rank], c->rank, c->rank, (unsigned long long)o.epochs[q], q);
            why = b; return false;
        }
        if (q == 0) memcpy(acc.data(), r.data[par], o.bytes);
        else reduce(acc.data(), r.data[par], o.count, o.dt, o.rop);
    }
    o.host.swap(acc);
    o.finished = true;
    return true;
}

bool async_mode() {
    static bool a = [] { const char* e = getenv("MOCK_RCCL_ASYNC"); return e && atoi(e) != 0 && !host_mode(); }();
    return a;
}
int async_delay_us() { static int d = [] { const char* e = getenv("MOCK_RCCL_ASYNC_DELAY_US"); return e ? atoi(e) : 0; }(); return d; }
hipStream_t g_proxy_stream = nullptr;                   // the proxy thread's own stream (async mode)

// prepare (caller thread): the bookkeeping that defines WHICH message / collective an operation is
ncclResult_t prepare_ops(std::vector<Op>& ops) {
    {
        std::set<std::pair<Comm*, int>> touched;
        for (Op& o : ops) {
            if (o.kind == ALLREDUCE) { for (int q = 0; q < o.c->world; ++q) if (q != o.c->rank) touched.insert({o.c, q}); }
            else touched.insert({o.c, o.peer});
        }
        for (auto& t : touched) t.first->epoch[t.second] += 1;
    }
    for (Op& op : ops) {
        Comm* c = op.c;
        std::string why;
        if (poisoned(c, why)) { t_last_error = c->last_error = why; return ncclRemoteError; }
        op.bytes = op.count * dt_size(op.dt);
        op.host.resize(op.bytes ? op.bytes : 1);
        if (op.kind == SEND) { op.index = c->n_sent[op.peer]++; op.epoch_tag = c->epoch[op.peer]; }
        if (op.kind == RECV) { op.index = c->n_recvd[op.peer]++; op.epoch_tag = c->epoch[op.peer]; }
        if (op.kind == ALLREDUCE) {
            if (op.bytes > kRedMax) return violation(c, ncclInvalidArgument, "ncclAllReduce of more than 256 KiB: the mock keeps collectives in one slot");
            op.coll_no = ++c->coll_seq;
            op.epochs.assign(c->epoch, c->epoch + kMaxWorld);
        }
        std::lock_guard<std::mutex> g(g_stats.mu);
        (op.kind == ALLREDUCE ? g_stats.coll_streams : g_stats.p2p_streams).insert((const void*)op.stream);
    }
    return ncclSuccess;
}

ncclResult_t transfer_ops(std::vector<Op>& ops, bool on_proxy);

ncclResult_t run_ops(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    ncclResult_t rc = prepare_ops(ops);
    if (rc != ncclSuccess) return rc;
    return transfer_ops(ops, false);
}

// move the bytes: gather what is sent, run the channels, deliver what is received.  on_proxy: the caller's streams are not touched —
// the events of the batch have been waited for, copies run on the proxy's stream.
ncclResult_t transfer_ops(std::vector<Op>& ops, bool on_proxy) {
    int dev0 = 0;
    if (!host_mode()) (void)hipGetDevice(&dev0);
    for (Op& op : ops) {
        Comm* c = op.c;
        if (host_mode()) { if (op.kind != RECV && op.bytes) memcpy(op.host.data(), op.sbuf, op.bytes); continue; }
        HIPQ(hipSetDevice(c->device));
        hipStream_t q = on_proxy ? g_proxy_stream : op.stream;
        if (!on_proxy) HIPQ(hipStreamSynchronize(op.stream));       // stream order: everything queued before the call has run
        if (op.kind != RECV && op.bytes) { HIPQ(hipMemcpyAsync(op.host.data(), op.sbuf, op.bytes, hipMemcpyDeviceToHost, q)); HIPQ(hipStreamSynchronize(q)); }
    }
    const auto t0 = Clock::now();
    size_t left = ops.size();
    unsigned spins = 0;
    while (left) {
        bool moved = false;
        std::set<std::pair<Comm*, long>> busy;                     // one message at a time per channel and direction, in call order
        for (Op& o : ops) {
            if (o.finished) continue;
            const std::pair<Comm*, long> key{o.c, o.kind == ALLREDUCE ? -1L : (long)(o.kind == SEND ? 1000 + o.peer : 2000 + o.peer)};
            if (busy.count(key)) continue;
            busy.insert(key);
            const int st = o.stage; const size_t db = o.done_bytes;
            std::string why;
            if (!step(o, why)) return violation(o.c, ncclInvalidUsage, why);
            if (o.finished) { left -= 1; moved = true; }
            else if (o.stage != st || o.done_bytes != db) moved = true;
        }
        if (moved) { spins = 0; continue; }
        for (Op& o : ops) { std::string why; if (!o.finished && poisoned(o.c, why)) { t_last_error = o.c->last_error = why; return ncclRemoteError; } }
        if (++spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s()) {
            std::string m = "no progress for " + std::to_string((int)timeout_s()) + " s; still waiting:";
            Comm* c0 = nullptr;
            for (Op& o : ops) if (!o.finished) { m += "\n    " + describe(o); c0 = o.c; }
            g_stats.timeouts += 1;
            return violation(c0, ncclSystemError, m);
        }
    }
    for (Op& op : ops) {
        if (op.kind == SEND) { g_stats.sends += 1; g_stats.send_bytes += op.bytes; uint64_t m = g_stats.max_send.load(); while (op.bytes > m && !g_stats.max_send.compare_exchange_weak(m, op.bytes)) {} continue; }
        if (op.kind == RECV) { g_stats.recvs += 1; g_stats.recv_bytes += op.bytes; } else g_stats.allreduces += 1;
        if (!op.bytes) continue;
        if (host_mode()) { memcpy(op.rbuf, op.host.data(), op.bytes); continue; }
        HIPQ(hipSetDevice(op.c->device));
        hipStream_t q = on_proxy ? g_proxy_stream : op.stream;
        HIPQ(hipMemcpyAsync(op.rbuf, op.host.data(), op.bytes, hipMemcpyHostToDevice, q));
        HIPQ(hipStreamSynchronize(q));
    }
    if (!host_mode()) (void)hipSetDevice(dev0);
    g_stats.groups += 1;
    return ncclSuccess;
}

// ---- MOCK_RCCL_ASYNC: the proxy ------------------------------------------------------------------------------------------------
struct Wait { int device; void* signal; uint32_t value; };
struct Batch { std::vector<Op> ops; std::vector<hipEvent_t> ready; std::vector<Wait> waits; };
struct Proxy {
    std::mutex mu; std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> q;
    std::thread th; bool started = false, stop = false;
    struct Slot { void* p = nullptr; uint32_t next = 0; };
    std::vector<Slot> slots; size_t rr = 0;
    ~Proxy() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
    void run() {
        for (;;) {
            std::unique_ptr<Batch> b;
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return stop || !q.empty(); }); if (q.empty()) return; b = std::move(q.front()); q.pop_front(); }
            ncclResult_t rc = ncclSuccess;
            g_stats.proxied += 1;
            for (size_t k = 0; k < b->ops.size(); ++k) {             // everything the caller queued before the call has run
                (void)hipSetDevice(b->ops[k].c->device);
                if (hipEventSynchronize(b->ready[k]) != hipSuccess) rc = ncclUnhandledCudaError;
                (void)hipEventDestroy(b->ready[k]);
            }
            if (async_delay_us() > 0) std::this_thread::sleep_for(std::chrono::microseconds(async_delay_us()));   // RCCL's kernels are not instant either
            if (rc == ncclSuccess) rc = transfer_ops(b->ops, true);
            std::set<Comm*> comms;
            for (Op& o : b->ops) comms.insert(o.c);
            for (Comm* c : comms) { if (rc != ncclSuccess) { int z = 0; c->async_error.compare_exchange_strong(z, (int)rc); } }
            // release the streams WHATEVER happened: a parked stream is a hung device
            for (const Wait& w : b->waits) { (void)hipSetDevice(w.device); (void)hipStreamWriteValue32(g_proxy_stream, w.signal, w.value, 0); }
            (void)hipStreamSynchronize(g_proxy_stream);
            for (Comm* c : comms) c->retired.fetch_add(1, std::memory_order_release);
        }
    }
    ncclResult_t submit(std::vector<Op>& ops) {
        if (!started) {
            if (hipStreamCreateWithFlags(&g_proxy_stream, hipStreamNonBlocking) != hipSuccess) return ncclUnhandledCudaError;
            slots.resize(64);
            th = std::thread([this] { run(); }); started = true;
        }
        for (Op& o : ops) { const int e = o.c->async_error.load(); if (e) { t_last_error = o.c->last_error; return (ncclResult_t)e; } }
        ncclResult_t rc = prepare_ops(ops);
        if (rc != ncclSuccess) return rc;
        auto b = std::make_unique<Batch>();
        std::set<std::pair<int, hipStream_t>> streams;
        for (Op& o : ops) {
            if (hipSetDevice(o.c->device) != hipSuccess) return ncclUnhandledCudaError;
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, o.stream) != hipSuccess) return ncclUnhandledCudaError;
            b->ready.push_back(ev);
            streams.insert({o.c->device, o.stream});
        }
        for (auto& ds : streams) {                                    // later work on the stream waits for the batch
            Slot& sl = slots[rr++ % slots.size()];
            (void)hipSetDevice(ds.first);
            if (!sl.p) {
                if (hipExtMallocWithFlags(&sl.p, 8, hipMallocSignalMemory) != hipSuccess) return ncclUnhandledCudaError;
                *(volatile uint64_t*)sl.p = 0;                       // signal memory comes uninitialised: a stale word ≥ 1 would let the first wait through
            }
            const uint32_t v = ++sl.next;
            if (hipStreamWaitValue32(ds.second, sl.p, v, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) return ncclUnhandledCudaError;
            b->waits.push_back({ds.first, sl.p, v});
        }
        std::set<Comm*> comms;
        for (Op& o : ops) comms.insert(o.c);
        for (Comm* c : comms) c->queued.fetch_add(1);
        b->ops = std::move(ops);
        { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(b)); }
        cv.notify_one();
        return ncclSuccess;
    }
    void drain(Comm* c) {                                             // every batch of this communicator has left the proxy
        const auto t0 = Clock::now();
        while (c->retired.load(std::memory_order_acquire) < c->queued.load()) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::duration<double>(Clock::now() - t0).count() > 2 * timeout_s() + 5) break;
        }
    }
} g_proxy;

ncclResult_t dispatch(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    if (async_mode()) { int dev0 = 0; (void)hipGetDevice(&dev0); const ncclResult_t rc = g_proxy.submit(ops); (void)hipSetDevice(dev0); return rc; }
    return run_ops(ops);
}

ncclResult_t submit(Op&& op) {
    if (!op.c) { t_last_error = "[mock-rccl] null communicator"; return ncclInvalidArgument; }
    {
        std::lock_guard<std::mutex> g(g_comm_mu);
        if (!g_live.count(op.c)) { g_stats.violations += 1; t_last_error = "[mock-rccl] VIOLATION: call on a communicator that was destroyed (or never made)"; fprintf(stderr, "%s\n", t_last_error.c_str()); return ncclInvalidArgument; }
    }
    if (op.kind != ALLREDUCE && (op.peer < 0 || op.peer >= op.c->world || op.peer == op.c->rank))
        return violation(op.c, ncclInvalidArgument, std::string(kind_name(op.kind)) + ": peer " + std::to_string(op.peer) + " is not another rank of this communicator");
    if (op.count && ((op.kind != RECV && !op.sbuf) || (op.kind != SEND && !op.rbuf))) return violation(op.c, ncclInvalidArgument, std::string(kind_name(op.kind)) + ": null buffer");
    if (t_depth > 0) { t_ops.push_back(std::move(op)); return ncclSuccess; }
    std::vector<Op> one; one.push_back(std::move(op));
    return dispatch(one);
}

Seg* attach(const ncclUniqueId* id, int world, int rank, std::shared_ptr<void>& keep, std::string& err) {
    const size_t n = sizeof(Seg);
    void* p = MAP_FAILED;
    std::string name;
    if (id) {
        name = seg_name(*id);
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) { err = "shm_open(" + name + ") failed"; return nullptr; }
        if (ftruncate(fd, (off_t)n) != 0) { close(fd); err = "ftruncate failed"; return nullptr; }
        p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
    } else p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { err = "mmap of the segment failed"; return nullptr; }
    auto m = std::make_shared<Mapping>(); m->p = p; m->n = n; keep = m;
    Seg* s = (Seg*)p;
    uint32_t z = 0;
    if (s->state.compare_exchange_strong(z, 1u)) { s->world = (uint32_t)world; s->state.store(2u, std::memory_order_release); }
    const auto t0 = Clock::now();
    while (s->state.load(std::memory_order_acquire) != 2u) {
        if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s()) { err = "the segment was never initialised"; return nullptr; }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if ((int)s->world != world) { err = "ncclCommInitRank: this rank says world = " + std::to_string(world) + ", the rank that made the segment said " + std::to_string(s->world); return nullptr; }
    if (id) {
        s->arrived.fetch_add(1);
        while ((int)s->arrived.load() < world) {
            if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s()) {
                err = "ncclCommInitRank: rank " + std::to_string(rank) + " waited " + std::to_string((int)timeout_s()) + " s for its peers (" + std::to_string(s->arrived.load()) + " of " + std::to_string(world) + " arrived)";
                shm_unlink(name.c_str());
                return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        shm_unlink(name.c_str());                                  // everybody has it mapped: the name can go (ENOENT for all but the first)
    }
    return s;
}

struct AtExit {
    ~AtExit() {
        if (t_depth != 0) { fprintf(stderr, "[mock-rccl] VIOLATION: the process ends inside an open ncclGroupStart (depth %d)\n", t_depth); g_stats.violations += 1; }
    }
} g_at_exit;

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    FILE* f = fopen("/dev/urandom", "rb");
    size_t got = f ? fread(id->internal, 1, sizeof id->internal, f) : 0;
    if (f) fclose(f);
    if (got != sizeof id->internal) { t_last_error = "[mock-rccl] /dev/urandom unreadable"; return ncclSystemError; }
    return ncclSuccess;
}

static ncclResult_t make_comm(ncclComm_t* out, Seg* s, std::shared_ptr<void> keep, int world, int rank, int device) {
    Comm* c = new Comm;
    c->seg = s; c->keep = std::move(keep); c->world = world; c->rank = rank; c->device = device;
    c->serial = (int)(g_stats.comms.fetch_add(1) + 1);
    g_stats.open_comms += 1;
    { std::lock_guard<std::mutex> g(g_comm_mu); g_live.insert(c); }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxWorld || rank < 0 || rank >= nranks) { t_last_error = "[mock-rccl] ncclCommInitRank: bad arguments"; return ncclInvalidArgument; }
    if (t_depth) return violation(nullptr, ncclInvalidUsage, "ncclCommInitRank inside an open group (the mock does not defer communicator creation)");
    int dev = 0;
    if (!host_mode() && hipGetDevice(&dev) != hipSuccess) { t_last_error = "[mock-rccl] no HIP device"; return ncclUnhandledCudaError; }
    std::shared_ptr<void> keep; std::string err;
    Seg* s = attach(&id, nranks, rank, keep, err);
    if (!s) return violation(nullptr, ncclSystemError, err);
    return make_comm(comm, s, keep, nranks, rank, dev);
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1 || ndev > kMaxWorld) { t_last_error = "[mock-rccl] ncclCommInitAll: bad arguments"; return ncclInvalidArgument; }
    std::shared_ptr<void> keep; std::string err;
    Seg* s = attach(nullptr, ndev, 0, keep, err);
    if (!s) return violation(nullptr, ncclSystemError, err);
    for (int r = 0; r < ndev; ++r) make_comm(&comms[r], s, keep, ndev, r, devlist ? devlist[r] : r);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    {
        std::lock_guard<std::mutex> g(g_comm_mu);
        if (!g_live.count(c)) { g_stats.violations += 1; t_last_error = "[mock-rccl] VIOLATION: ncclCommDestroy of a communicator that is not alive"; fprintf(stderr, "%s\n", t_last_error.c_str()); return ncclInvalidArgument; }
    }
    ncclResult_t rc = ncclSuccess;
    if (async_mode()) { g_proxy.drain(c); if (c->async_error.load()) rc = (ncclResult_t)c->async_error.load(); }
    if (t_depth) rc = violation(c, ncclInvalidUsage, "ncclCommDestroy inside an open ncclGroupStart");
    if (!c->seg->poisoned.load()) {
        for (int q = 0; q < c->world; ++q) {
            if (q == c->rank) continue;
            Chan& in = c->seg->chan[q * kMaxWorld + c->rank];
            if (in.head.load(std::memory_order_acquire) != in.tail.load(std::memory_order_acquire))
                rc = violation(c, ncclInvalidUsage, "ncclCommDestroy: rank " + std::to_string(q) + " sent a message to rank " + std::to_string(c->rank) + " that was never received");
        }
    }
    { std::lock_guard<std::mutex> g(g_comm_mu); g_live.erase(c); }
    g_stats.open_comms -= 1;
    delete c;
    return rc;
}

ncclResult_t ncclGroupStart() { t_depth += 1; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return violation(nullptr, ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops; ops.swap(t_ops);
    return dispatch(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Op o{SEND, (Comm*)comm, peer, sendbuff, nullptr, count, datatype, ncclSum, stream};
    return submit(std::move(o));
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Op o{RECV, (Comm*)comm, peer, nullptr, recvbuff, count, datatype, ncclSum, stream};
    return submit(std::move(o));
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Op o{ALLREDUCE, (Comm*)comm, -1, sendbuff, recvbuff, count, datatype, op, stream};
    return submit(std::move(o));
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled hip error (mock-rccl)";
        case ncclSystemError: return "unhandled system error (mock-rccl: a wait ran into its deadline)";
        case ncclInternalError: return "internal error (mock-rccl)";
        case ncclInvalidArgument: return "invalid argument (mock-rccl)";
        case ncclInvalidUsage: return "invalid usage (mock-rccl: the ranks' call sequences do not match)";
        case ncclRemoteError: return "remote error (mock-rccl: a peer reported a violation)";
        default: return "unknown result code (mock-rccl)";
    }
}
const char* ncclGetLastError(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (c) { std::lock_guard<std::mutex> g(g_comm_mu); if (g_live.count(c) && !c->last_error.empty()) return c->last_error.c_str(); }
    return t_last_error.c_str();
}

// ---- what the tests read (not part of the RCCL surface) --------------------------------------------------------------
// out[0] communicators made, [1] groups (or lone calls) completed, [2] sends, [3] receives, [4] bytes sent, [5] bytes received,
// [6] allreduces, [7] violations, [8] distinct streams named by point-to-point calls, [9] by collectives, [10] longest send in bytes,
// [11] communicators alive, [12] current group depth of the calling thread, [13] deadline expiries, [14] batches the proxy thread ran
// ($MOCK_RCCL_ASYNC=1; 0 in the default mode)
int mockrccl_stats(uint64_t* out, int cap) {
    uint64_t v[15] = {g_stats.comms, g_stats.groups, g_stats.sends, g_stats.recvs, g_stats.send_bytes, g_stats.recv_bytes, g_stats.allreduces, g_stats.violations, 0, 0,
                      g_stats.max_send, g_stats.open_comms, (uint64_t)t_depth, g_stats.timeouts, g_stats.proxied};
    { std::lock_guard<std::mutex> g(g_stats.mu); v[8] = g_stats.p2p_streams.size(); v[9] = g_stats.coll_streams.size(); }
    for (int i = 0; i < cap && i < 15; ++i) out[i] = v[i];
    return 15;
}
const char* mockrccl_version(void) { return "mock-rccl 1 (tests/mock_rccl/mock_rccl.cpp): staged copies over POSIX shared memory, checked message lists"; }

}  // extern "C"
