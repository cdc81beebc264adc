// Test stand-in for librccl: the entry points the fluid library binds with dlopen (nine, and ncclCommSplit), over POSIX
// shared memory between processes that may all sit on ONE GPU.
//
// RCCL refuses two ranks of a communicator on the same device, so the rank-to-rank side of the z-slab code (who
// sends which planes to whom, in which order, on which stream; the all-reduced stopping scalars) cannot run
// against the real library on a one-GPU machine.  Loaded through FSAMD_RCCL_LIBRARY this file lets N processes
// drive the unchanged library code on one card.  It is test infrastructure: nothing here is fast, and nothing
// in the product loads it unless that variable names it.
//
// Semantics kept: operations of one communicator complete in issue order; a send matches the receive the peer
// posts in the same position of its own sequence towards this rank; grouped sends/receives make progress
// together; data is in device memory before any work queued later on the same stream runs (every call drains
// the stream it is given, moves the bytes, and returns).  Every wait has a deadline and fails with
// ncclSystemError instead of hanging, so a mismatched sequence shows up as an error text.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlot = size_t(1) << 20;   // bytes per mailbox fill; longer messages go in pieces
constexpr int kMaxRed = 16;
constexpr double kDeadline = 60.0;          // seconds before a wait gives up

struct Mailbox {
    std::atomic<uint32_t> full;
    uint32_t bytes;
    char pad[56];
    char data[kSlot];
};

struct Shared {
    std::atomic<int> attached;
    std::atomic<int> bar_count;
    std::atomic<int> bar_gen;
    std::atomic<int> failed;
    double red[kMaxRanks][kMaxRed];
    // a second set of barrier and reduction slots for the communicator ncclCommSplit hands out (the library
    // uses it for all-reduces only, concurrently with send/receive traffic on the first)
    std::atomic<int> bar2_count;
    std::atomic<int> bar2_gen;
    double red2[kMaxRanks][kMaxRed];
    Mailbox box[kMaxRanks * kMaxRanks];   // [src * kMaxRanks + dst]
};

struct Op {
    bool send;
    char* dev;
    size_t bytes, done;
    int peer;
    hipStream_t stream;
};

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

thread_local int group_depth = 0;
thread_local std::vector<std::pair<struct ncclComm*, Op>> group_ops;

}  // namespace

struct ncclComm {
    Shared* sh = nullptr;
    int rank = 0, nranks = 1;
    bool second = false;   // made by ncclCommSplit
    char name[64];
};

namespace {

ncclResult_t barrier(ncclComm* c) {
    Shared* sh = c->sh;
    std::atomic<int>& count = c->second ? sh->bar2_count : sh->bar_count;
    std::atomic<int>& bgen = c->second ? sh->bar2_gen : sh->bar_gen;
    const int gen = bgen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        count.store(0, std::memory_order_relaxed);
        bgen.fetch_add(1, std::memory_order_acq_rel);
        return ncclSuccess;
    }
    const double t0 = now();
    while (bgen.load(std::memory_order_acquire) == gen) {
        if (sh->failed.load() || now() - t0 > kDeadline) { sh->failed.store(1); return ncclSystemError; }
        usleep(20);
    }
    return ncclSuccess;
}

// Runs a set of sends and receives of one communicator to completion, the head of every (peer, direction)
// queue making progress whenever its mailbox allows.
ncclResult_t run_ops(ncclComm* c, std::vector<Op>& ops) {
    for (Op& o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::deque<Op*> sendq[kMaxRanks], recvq[kMaxRanks];
    for (Op& o : ops) (o.send ? sendq : recvq)[o.peer].push_back(&o);
    size_t left = ops.size();
    double last_progress = now();
    while (left) {
        bool moved = false;
        for (int p = 0; p < c->nranks; ++p) {
            if (!sendq[p].empty()) {
                Op* o = sendq[p].front();
                Mailbox& mb = c->sh->box[c->rank * kMaxRanks + p];
                if (!mb.full.load(std::memory_order_acquire)) {
                    const size_t n = std::min(kSlot, o->bytes - o->done);
                    if (n && hipMemcpy(mb.data, o->dev + o->done, n, hipMemcpyDeviceToHost) != hipSuccess)
                        return ncclUnhandledCudaError;
                    mb.bytes = (uint32_t)n;
                    mb.full.store(1, std::memory_order_release);
                    o->done += n;
                    moved = true;
                    if (o->done == o->bytes) { sendq[p].pop_front(); --left; }
                }
            }
            if (!recvq[p].empty()) {
                Op* o = recvq[p].front();
                Mailbox& mb = c->sh->box[p * kMaxRanks + c->rank];
                if (mb.full.load(std::memory_order_acquire)) {
                    const size_t n = mb.bytes;
                    if (n > o->bytes - o->done) { c->sh->failed.store(1); return ncclInvalidArgument; }   // sizes disagree
                    if (n && hipMemcpy(o->dev + o->done, mb.data, n, hipMemcpyHostToDevice) != hipSuccess)
                        return ncclUnhandledCudaError;
                    mb.full.store(0, std::memory_order_release);
                    o->done += n;
                    moved = true;
                    if (o->done == o->bytes) { recvq[p].pop_front(); --left; }
                }
            }
        }
        if (moved) last_progress = now();
        else {
            if (c->sh->failed.load() || now() - last_progress > kDeadline) { c->sh->failed.store(1); return ncclSystemError; }
            usleep(10);
        }
    }
    return ncclSuccess;
}

ncclResult_t post(ncclComm* c, bool send, const void* buf, size_t count, ncclDataType_t dt, int peer, hipStream_t st) {
    if (!c || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    size_t esz = 1;
    switch (dt) {
        case ncclInt8: case ncclUint8: esz = 1; break;
        case ncclFloat16: case ncclBfloat16: esz = 2; break;
        case ncclInt32: case ncclUint32: case ncclFloat32: esz = 4; break;
        case ncclInt64: case ncclUint64: case ncclFloat64: esz = 8; break;
        default: return ncclInvalidArgument;
    }
    Op o{send, (char*)const_cast<void*>(buf), count * esz, 0, peer, st};
    if (group_depth > 0) { group_ops.emplace_back(c, o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(c, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<int> serial{0};
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/fsamd_fake_rccl_%d_%d_%ld", (int)getpid(), serial.fetch_add(1),
                  (long)(now() * 1e3) % 1000000L);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    const bool ok = ftruncate(fd, sizeof(Shared)) == 0;   // fresh pages read as zero: every atomic starts at 0
    close(fd);
    return ok ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ncclComm* c = new ncclComm();
    c->sh = (Shared*)p;
    c->rank = rank;
    c->nranks = nranks;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->sh->attached.fetch_add(1);
    const double t0 = now();
    while (c->sh->attached.load() < nranks) {
        if (now() - t0 > kDeadline) { munmap(p, sizeof(Shared)); delete c; return ncclSystemError; }
        usleep(100);
    }
    const ncclResult_t r = barrier(c);
    if (rank == 0) shm_unlink(c->name);   // everybody holds a mapping now; the name can go
    if (r != ncclSuccess) { munmap(p, sizeof(Shared)); delete c; return r; }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (!c->second) munmap(c->sh, sizeof(Shared));   // a split communicator borrows its parent's mapping
    delete c;
    return ncclSuccess;
}

// Only what the library asks for: one colour, ranks in their order, all-reduces on the new communicator.  It must
// be destroyed before its parent.
ncclResult_t ncclCommSplit(ncclComm_t c, int color, int key, ncclComm_t* out, ncclConfig_t* config) {
    (void)config;
    if (!c || !out || c->second || color != 0 || key != c->rank) return ncclInvalidArgument;
    ncclComm* n = new ncclComm(*c);
    n->second = true;
    *out = n;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (group_depth <= 0) return ncclInvalidUsage;
    if (--group_depth > 0) return ncclSuccess;
    ncclResult_t res = ncclSuccess;
    while (!group_ops.empty() && res == ncclSuccess) {   // one communicator at a time, in order of first appearance
        ncclComm* c = group_ops.front().first;
        std::vector<Op> mine;
        std::vector<std::pair<ncclComm*, Op>> rest;
        for (auto& e : group_ops) (e.first == c ? (void)mine.push_back(e.second) : (void)rest.push_back(e));
        group_ops.swap(rest);
        res = run_ops(c, mine);
    }
    group_ops.clear();
    return res;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    return post(c, true, buf, count, dt, peer, st);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    return post(c, false, buf, count, dt, peer, st);
}

ncclResult_t ncclAllReduce(const void* in, void* out, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
    if (!c || dt != ncclFloat64 || count > (size_t)kMaxRed || (op != ncclSum && op != ncclMax && op != ncclMin))
        return ncclInvalidArgument;   // the library reduces fp64 scalars only
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    double (*red)[kMaxRed] = c->second ? c->sh->red2 : c->sh->red;
    double* mine = red[c->rank];
    if (hipMemcpy(mine, in, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = barrier(c);
    if (r != ncclSuccess) return r;
    double acc[kMaxRed];
    for (size_t i = 0; i < count; ++i) {
        double a = red[0][i];
        for (int k = 1; k < c->nranks; ++k) {   // rank order: every rank forms the same sum
            const double b = red[k][i];
            a = op == ncclSum ? a + b : op == ncclMax ? (b > a ? b : a) : (b < a ? b : a);
        }
        acc[i] = a;
    }
    r = barrier(c);   // nobody overwrites its slot before everybody has read it
    if (r != ncclSuccess) return r;
    if (hipMemcpy(out, acc, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "HIP call failed (stand-in transport)";
        case ncclSystemError: return "peer did not arrive within the deadline (stand-in transport)";
        case ncclInvalidArgument: return "invalid argument (stand-in transport)";
        case ncclInvalidUsage: return "invalid usage (stand-in transport)";
        default: return "error (stand-in transport)";
    }
}

}  // extern "C"
