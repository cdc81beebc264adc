// C ABI of the 3D fluid step (include/fluidsim_amd.h): host logic that owns the
// device state, launches the kernels of fs_kernels3d.hpp on the handle's
// stream and moves halo planes between z-slabs.
//
// State layout in HBM.  The global nx*ny*nz grid is cut along z (the slowest
// axis, so a slab is one contiguous range of planes) into P = nranks *
// local_slabs parts.  A part holds its owned planes plus H halo planes on
// either side when P > 1 (H = 2: one plane for the stencil, one for a
// backtrace of up to two cells; where every slab owns at least four planes the
// allocations carry D = 2 more planes beyond those, which only the pass of two
// sweeps reads, see launch_rows2), as dense x-fastest arrays:
//   f[2][4]   u, v, w, density -- two generations; advection writes the other
//             generation and the roles swap, which replaces the reference's
//             snapshot copies (main.cpp:122-127);
//   tmp[3]    the unprojected half-step velocity of the reflection step, also
//             the divergence output;
//   cells     the reference's m_s as one byte per cell; flags = the neighbour
//             bits derived from it.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "fs_comm.hpp"
#include "fs_common.hpp"
#include "fs_kernels3d.hpp"
#include "fs_sweep_fused.hpp"
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include "fs_sweep_rows.hpp"
#include "fs_sweep_rows2.hpp"
#include "fs_advect_tiled.hpp"
#include "fs_sweep_advect.hpp"
#include "fs_diag3d.hpp"
#include "fs_mg3d.hpp"

using namespace fsamd;

namespace {

constexpr int kReduceBlocks = 1024;

struct Part {
    Grid g{};
    int z0 = 0, z1 = 0;   // owned planes, global
    int H = 0;            // halo planes on either side
    int D = 0;            // "deep" halo planes beyond those: allocated (every array starts D planes before the plane its
                          // pointer addresses and ends D planes after local plane nzl - 1) but not counted in g.nzl
    int zb = 0, ze = 0;   // local range of the interior planes this part relaxes / advects
    int gp = 0;           // global part index
    bool has_lo = false, has_hi = false;
    size_t n = 0;         // local element count
    void* f[2][4] = {{nullptr}};
    void* tmp[3] = {nullptr};
    void* mc[4] = {nullptr};   // outputs of the MacCormack pass, allocated when the scheme is first used
    struct MgRaw { int nx, ny, nz; void* a[6]; };   // wx, wy, wz, diag, e, r of one coarse level
    std::vector<MgRaw> mg;     // multigrid levels, allocated when the solver is first used
    bool mg_weights_ok = false;
    uint8_t* cells = nullptr;
    uint8_t* flags = nullptr;
    uint8_t* vflags = nullptr;   // visit bytes of the register-resident sweeps (k_build_flags)
    void* recv_lo = nullptr;   // staging for the interface w plane from the lower / upper rank
    void* recv_hi = nullptr;
    double* partials = nullptr;
    double* adv_part = nullptr;   // per-workgroup residual partials of an advection that verifies a projection
    size_t adv_cap = 0;           // pairs it holds
    int adv_blocks = 0;           // pairs the last such advection wrote
    int own_planes() const { return z1 - z0; }
};

}  // namespace

struct fs3d_sim {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;        // halo traffic of the one-pass sweep, overlapped with its interior
    hipEvent_t ev_bnd = nullptr, ev_comm = nullptr, ev_b = nullptr;
    int mg_backoff[2] = {0, 0};  // projections to wait before plain sweeps are tried again under the multigrid solver
    int mg_fails[2] = {0, 0};    // consecutive failed tries
    int spec_backoff[2] = {0, 0};   // deferred residual check: projections to wait after a failed one,
    int spec_fails[2] = {0, 0};     // consecutive failures
    int fuse_backoff[2] = {0, 0};   // one sweep + advection in one launch: projections to wait after a launch that could
    int fuse_fails[2] = {0, 0};     // not serve every cell (or missed the tolerance), and consecutive such launches
    bool halo_pending = false;   // work queued on comm_stream that the main stream has not waited for yet
    bool deep = false;           // the slabs carry deep halo planes (Part::D > 0): every slab of the grid owns >= 4 planes
    bool deep_ok = false;        // the deep halo planes of the current velocity generation are current (or on their way
                                 // on comm_stream): true after a pass of two sweeps, false after anything else that
                                 // changes the velocity
    fs3d_config cfg{};
    int nx = 0, ny = 0, nz = 0;
    size_t esz = 4;
    int rank = 0, nranks = 1, nlocal = 1, nparts_global = 1;
    std::vector<Part> parts;
    int curv = 0, curd = 0;      // generation holding the current velocity (u,v,w) / density
    int tilde_gen = -1;          // inside a reflection step: the generation whose velocity arrays still hold the
                                 // unprojected half-step field (else it is in the scratch fields, or not needed)
    bool cells_dirty = true;
    int halo_dirty = 7;          // halo planes that lag their owners: bit 0 velocity, bit 1 density, bit 2 scratch velocity
    double* dres = nullptr;      // 2 doubles per local part, then the combined record: sum, out-of-halo count, max
    double* hres = nullptr;      // pinned mirror of the combined record; [3] = ticket of the record (single-part handles)
    double ticket = 0.0;         // ticket of the record queued last
    int* doob = nullptr;         // device flag the advection kernels raise: a backtrace left the slab's halo
    ncclComm_t comm = nullptr;
    ncclComm_t comm_red = nullptr;   // the communicator of the scalar all-reduces: a copy of `comm` where the library
                                     // can make one (the reductions then need not queue behind the halo traffic), else `comm`
    Rccl* rccl = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    fs3d_stats stats{};
    int num_cus = 256;
    int sweep_hint[2] = {1, 1};  // sweeps the first / second projection of the previous step needed
    int sweep_hint_prev[2] = {0, 0};   // and of the step before that
    double resid_dt = 1.0;       // the dt in b = -div/dt of the L2 residual (3_fluid/FluidSim.cpp:17)
    std::string err;
    // Communication accounting (fs3d_get_comm_stats).  Byte and call counts are always kept; device times need
    // event pairs around every exchange and every wait, which is switched on by fs3d_comm_timing.
    struct Timed { hipEvent_t a = nullptr, b = nullptr; int kind = 0; bool busy = false; };
    std::vector<Timed> timed;    // ring of event pairs
    size_t timed_next = 0;
    bool comm_timing = false;
    fs3d_comm_stats cstats{};
};

// Raises a kernel's dynamic LDS limit once per kernel and device (the call is not free, and a sweep
// launches the same kernel thousands of times).
inline hipError_t allow_lds(fs3d_sim* s, const void* kern, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kern, s->device})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.insert({kern, s->device});
    return e;
}

namespace {

template <typename T> T* F(Part& p, int gen, int k) { return (T*)p.f[gen][k]; }

int set_err(fs3d_sim* s, int code, const std::string& msg) { s->err = msg; return code; }

#define SIM_HIP(call) FS_HIP(s->err, call)
#define SIM_GUARD(s)                                                                \
    if (!(s)) return FS_EINVAL;                                                     \
    DeviceGuard guard__((s)->device);                                               \
    if (!guard__.ok) return set_err((s), FS_EHIP, "hipSetDevice failed")

int nccl_check(fs3d_sim* s, ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return FS_OK;
    return set_err(s, FS_ECOMM, std::string(what) + ": " + s->rccl->GetErrorString(r));
}

void split_planes(int nz, int P, int gp, int* z0, int* z1) {
    const int base = nz / P, rem = nz % P;
    *z0 = gp * base + std::min(gp, rem);
    *z1 = *z0 + base + (gp < rem ? 1 : 0);
}

// Every per-cell array of a part is allocated with D more planes on either side than its local grid counts, and the
// pointer the part keeps addresses local plane 0.
size_t deep_bytes(const Part& p, size_t esz) { return (size_t)p.D * (size_t)p.g.sz * esz; }
size_t alloc_bytes(const Part& p, size_t esz) { return p.n * esz + 2 * deep_bytes(p, esz); }

int alloc_field(fs3d_sim* s, const Part& p, void** out, size_t esz) {
    char* base = nullptr;
    SIM_HIP(hipMalloc((void**)&base, alloc_bytes(p, esz)));
    *out = base + deep_bytes(p, esz);
    return FS_OK;
}

void free_field(const Part& p, void* ptr, size_t esz) {
    if (ptr) (void)hipFree((char*)ptr - deep_bytes(p, esz));
}

// The local grid widened by the deep planes, and an array pointer moved to its first deep plane.
Grid deep_grid(const Part& p) {
    Grid g = p.g;
    g.nzl += 2 * p.D;
    g.gz0 -= p.D;
    return g;
}
template <typename P> P* deep_ptr(const Part& p, P* a) { return a - (ptrdiff_t)p.D * p.g.sz; }

void free_parts(fs3d_sim* s) {
    for (Part& p : s->parts) {
        for (int g = 0; g < 2; ++g)
            for (int k = 0; k < 4; ++k) free_field(p, p.f[g][k], s->esz);
        for (int k = 0; k < 3; ++k) free_field(p, p.tmp[k], s->esz);
        for (int k = 0; k < 4; ++k) free_field(p, p.mc[k], s->esz);
        for (auto& l : p.mg) for (void* a : l.a) (void)hipFree(a);
        free_field(p, p.cells, 1); free_field(p, p.flags, 1); free_field(p, p.vflags, 1);
        (void)hipFree(p.recv_lo); (void)hipFree(p.recv_hi);
        (void)hipFree(p.partials);
        (void)hipFree(p.adv_part);
    }
    s->parts.clear();
}

int alloc_parts(fs3d_sim* s) {
    const int P = s->nparts_global;
    s->parts.resize(s->nlocal);
    // deep halo planes (for the pass of two sweeps) where every slab of the grid can fill four planes of a neighbour
    // from its own planes; every rank comes to the same answer
    s->deep = P > 1 && s->nz / P >= 4;
    for (int k = 0; k < s->nlocal; ++k) {
        Part& p = s->parts[k];
        p.gp = s->rank * s->nlocal + k;
        split_planes(s->nz, P, p.gp, &p.z0, &p.z1);
        p.H = P > 1 ? 2 : 0;
        p.D = s->deep ? 2 : 0;
        p.has_lo = p.gp > 0;
        p.has_hi = p.gp < P - 1;
        if (P > 1 && p.own_planes() < p.H)
            return set_err(s, FS_EINVAL, "too many slabs: every slab needs at least 2 planes");
        p.g.nx = s->nx; p.g.ny = s->ny; p.g.nzl = p.own_planes() + 2 * p.H;
        p.g.gz0 = p.z0 - p.H; p.g.gnz = s->nz;
        p.g.sy = s->nx; p.g.sz = (int64_t)s->nx * s->ny;
        p.zb = std::max(p.z0, 1) - p.g.gz0;
        p.ze = std::min(p.z1, s->nz - 1) - p.g.gz0;
        if (p.ze < p.zb) p.ze = p.zb;
        p.n = (size_t)p.g.sz * p.g.nzl;
        for (int g = 0; g < 2; ++g)
            for (int q = 0; q < 4; ++q) FS_TRY(alloc_field(s, p, &p.f[g][q], s->esz));
        for (int q = 0; q < 3; ++q) FS_TRY(alloc_field(s, p, &p.tmp[q], s->esz));
        FS_TRY(alloc_field(s, p, (void**)&p.cells, 1));
        FS_TRY(alloc_field(s, p, (void**)&p.flags, 1));
        FS_TRY(alloc_field(s, p, (void**)&p.vflags, 1));
        SIM_HIP(hipMalloc((void**)&p.partials, sizeof(double) * 2 * kReduceBlocks));
        if (p.H) {
            SIM_HIP(hipMalloc(&p.recv_lo, (size_t)p.g.sz * s->esz));
            SIM_HIP(hipMalloc(&p.recv_hi, (size_t)p.g.sz * s->esz));
        }
    }
    return FS_OK;
}

// ---- cell types ----------------------------------------------------------
// FluidSim::FluidSim (main.cpp:28-33): shell = Object, rest Fluid.  Planes of a
// halo that lie outside the global grid are Object as well.
int init_cells(fs3d_sim* s) {
    for (Part& p : s->parts) {
        std::vector<uint8_t> h(alloc_bytes(p, 1));
        for (int lz = -p.D; lz < p.g.nzl + p.D; ++lz) {
            const int gz = lz + p.g.gz0;
            for (int y = 0; y < s->ny; ++y)
                for (int x = 0; x < s->nx; ++x) {
                    const bool shell = gz <= 0 || gz >= s->nz - 1 || y == 0 || y == s->ny - 1 || x == 0 ||
                                       x == s->nx - 1;
                    h[(size_t)(lz + p.D) * p.g.sz + (size_t)y * s->nx + x] = shell ? FS_CELL_OBJECT : FS_CELL_FLUID;
                }
        }
        SIM_HIP(hipMemcpy(deep_ptr(p, p.cells), h.data(), h.size(), hipMemcpyHostToDevice));
    }
    s->cells_dirty = true;
    return FS_OK;
}

int zero_fields(fs3d_sim* s) {
    for (Part& p : s->parts) {
        for (int g = 0; g < 2; ++g)
            for (int q = 0; q < 4; ++q)
                SIM_HIP(hipMemsetAsync((char*)p.f[g][q] - deep_bytes(p, s->esz), 0, alloc_bytes(p, s->esz), s->stream));
        for (int q = 0; q < 3; ++q)
            SIM_HIP(hipMemsetAsync((char*)p.tmp[q] - deep_bytes(p, s->esz), 0, alloc_bytes(p, s->esz), s->stream));
    }
    s->deep_ok = false;
    return FS_OK;
}

// ---- communication accounting ------------------------------------------------
void timed_resolve(fs3d_sim* s, fs3d_sim::Timed& t) {
    if (!t.busy) return;
    float ms = 0;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess)
        (t.kind == 0 ? s->cstats.exchange_ms : s->cstats.wait_ms) += ms;
    t.busy = false;
}

// Records the first event of a pair on `st` and returns the pair (nullptr when timing is off); timed_end
// records the second.  kind 0: an exchange on the communication stream, 1: the main stream waiting for one.
fs3d_sim::Timed* timed_begin(fs3d_sim* s, hipStream_t st, int kind) {
    if (!s->comm_timing) return nullptr;
    if (s->timed.empty()) {
        s->timed.resize(256);
        for (auto& t : s->timed)
            if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { s->comm_timing = false; return nullptr; }
    }
    fs3d_sim::Timed& t = s->timed[s->timed_next];
    s->timed_next = (s->timed_next + 1) % s->timed.size();
    timed_resolve(s, t);   // the oldest pair: long finished
    t.kind = kind;
    if (hipEventRecord(t.a, st) != hipSuccess) return nullptr;
    return &t;
}

void timed_end(fs3d_sim::Timed* t, hipStream_t st) {
    if (t && hipEventRecord(t->b, st) == hipSuccess) t->busy = true;
}

// ---- halo transport -------------------------------------------------------
// Copies `planes` planes of `bytes_per_plane` bytes between neighbouring parts:
// from local plane src_lz of part a into local plane dst_lz of part b, for
// every interface, in both directions.  Parts in this process copy device to
// device on the stream; parts of another rank go through ncclSend/ncclRecv.
struct HaloOp {
    // array selector: returns the base pointer of the array in part p
    void* (*base)(fs3d_sim*, Part&, int) = nullptr;
    int which = 0;
    size_t elem = 1;
    // the planes moved, by their distance from the interface: [d0, d1), d1 = 0 meaning the H halo planes [0, H);
    // d1 up to H + D reaches the deep planes
    int d0 = 0, d1 = 0;
};

int exchange_halos_impl(fs3d_sim* s, const std::vector<HaloOp>& ops, hipStream_t st);

int exchange_halos(fs3d_sim* s, const std::vector<HaloOp>& ops, hipStream_t st = nullptr) {
    if (s->nparts_global == 1) return FS_OK;
    if (!st) st = s->stream;
    fs3d_sim::Timed* t = timed_begin(s, st, 0);
    const int rc = exchange_halos_impl(s, ops, st);
    timed_end(t, st);
    ++s->cstats.exchanges;
    return rc;
}

// Up to kCopyBatch device-to-device copies in one launch: the halo planes of local slabs.  (An exchange between the
// slabs of one process is 2 copies per interface and field - 42 for eight slabs and three fields - and as many calls
// of hipMemcpyAsync take 6 us each on their stream, whatever they move.)
constexpr int kCopyBatch = 48;
struct CopyBatch { const char* src[kCopyBatch]; char* dst[kCopyBatch]; size_t bytes[kCopyBatch]; };

__global__ void __launch_bounds__(256) k_copy_batch(CopyBatch b) {
    const char* src = b.src[blockIdx.y];
    char* dst = b.dst[blockIdx.y];
    const size_t bytes = b.bytes[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    if ((((uintptr_t)src | (uintptr_t)dst | bytes) & 15u) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = tid; i < bytes / 16; i += nthreads) d4[i] = s4[i];
    } else if ((((uintptr_t)src | (uintptr_t)dst | bytes) & 3u) == 0) {   // (planes of an odd number of 4-byte cells)
        const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
        uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
        for (size_t i = tid; i < bytes / 4; i += nthreads) d1[i] = s1[i];
    } else {
        for (size_t i = tid; i < bytes; i += nthreads) dst[i] = src[i];
    }
}

int exchange_halos_impl(fs3d_sim* s, const std::vector<HaloOp>& ops, hipStream_t st) {
    // Planes at distance [d0, d1) from an interface: the owned planes H + d (below: H + own - 1 - d) of a part fill
    // the halo planes H + own + d (H - 1 - d) of its lower (upper) neighbour.  All parts hold the same H.
    bool remote = false;
    CopyBatch batch;
    int queued = 0;
    size_t largest = 0;
    auto flush = [&]() -> int {
        if (queued == 0) return FS_OK;
        const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>(largest / (16 * 256 * 4), 1), 256);
        hipLaunchKernelGGL(k_copy_batch, dim3(blocks, queued, 1), dim3(256), 0, st, batch);
        SIM_HIP(hipGetLastError());
        queued = 0;
        largest = 0;
        return FS_OK;
    };
    auto copy = [&](char* dst, const char* src, size_t bytes) -> int {
        batch.dst[queued] = dst; batch.src[queued] = src; batch.bytes[queued] = bytes;
        largest = std::max(largest, bytes);
        if (++queued == kCopyBatch) return flush();
        return FS_OK;
    };
    for (Part& p : s->parts) {
        const ptrdiff_t H = p.H, own = p.own_planes();
        for (const HaloOp& op : ops) {
            const ptrdiff_t plane = (ptrdiff_t)p.g.sz * (ptrdiff_t)op.elem, d0 = op.d0, d1 = op.d1 > 0 ? op.d1 : H;
            char* mine = (char*)op.base(s, p, op.which);
            // upper neighbour in this process: fill its low halo from my top owned planes, and my
            // high halo from its first owned planes
            if (p.has_hi && (&p != &s->parts.back())) {
                Part& q = *(&p + 1);
                char* theirs = (char*)op.base(s, q, op.which);
                FS_TRY(copy(theirs + (H - d1) * plane, mine + (H + own - d1) * plane, (size_t)((d1 - d0) * plane)));
                FS_TRY(copy(mine + (H + own + d0) * plane, theirs + (H + d0) * plane, (size_t)((d1 - d0) * plane)));
            }
        }
        if ((p.has_lo && &p == &s->parts.front()) || (p.has_hi && &p == &s->parts.back())) remote = true;
    }
    FS_TRY(flush());
    if (!remote || s->nranks == 1) return FS_OK;
    Rccl* R = s->rccl;
    FS_TRY(nccl_check(s, R->GroupStart(), "ncclGroupStart"));
    for (const HaloOp& op : ops) {
        Part& lo = s->parts.front();
        Part& hi = s->parts.back();
        if (lo.has_lo) {
            const ptrdiff_t plane = (ptrdiff_t)lo.g.sz * (ptrdiff_t)op.elem, H = lo.H, d0 = op.d0, d1 = op.d1 > 0 ? op.d1 : H;
            const size_t bytes = (size_t)((d1 - d0) * plane);
            char* a = (char*)op.base(s, lo, op.which);
            FS_TRY(nccl_check(s, R->Send(a + (H + d0) * plane, bytes, ncclUint8, s->rank - 1, s->comm, st), "ncclSend"));
            FS_TRY(nccl_check(s, R->Recv(a + (H - d1) * plane, bytes, ncclUint8, s->rank - 1, s->comm, st), "ncclRecv"));
            s->cstats.bytes_sent_lo += (int64_t)bytes;
        }
        if (hi.has_hi) {
            const ptrdiff_t plane = (ptrdiff_t)hi.g.sz * (ptrdiff_t)op.elem, H = hi.H, own = hi.own_planes(), d0 = op.d0,
                            d1 = op.d1 > 0 ? op.d1 : H;
            const size_t bytes = (size_t)((d1 - d0) * plane);
            char* a = (char*)op.base(s, hi, op.which);
            FS_TRY(nccl_check(s, R->Send(a + (H + own - d1) * plane, bytes, ncclUint8, s->rank + 1, s->comm, st), "ncclSend"));
            FS_TRY(nccl_check(s, R->Recv(a + (H + own + d0) * plane, bytes, ncclUint8, s->rank + 1, s->comm, st), "ncclRecv"));
            s->cstats.bytes_sent_hi += (int64_t)bytes;
        }
    }
    return nccl_check(s, R->GroupEnd(), "ncclGroupEnd");
}

void* base_field_cur(fs3d_sim* s, Part& p, int k) { return p.f[k < 3 ? s->curv : s->curd][k]; }
void* base_flags(fs3d_sim*, Part& p, int which) { return which == 0 ? p.flags : p.vflags; }
void* base_tmp(fs3d_sim*, Part& p, int k) { return p.tmp[k]; }
void* base_cells(fs3d_sim*, Part& p, int) { return p.cells; }

enum { HALO_VEL = 1, HALO_DEN = 2, HALO_ALL = 3, HALO_TMP = 4 };

// Halo traffic runs on comm_stream and the main stream does not wait for it when it is queued: the next
// kernel may be the interior of a one-pass sweep, which reads no halo plane.  Everything else that reads
// halo planes, touches the communicator or hands data to the host calls join_halos first.
int join_halos(fs3d_sim* s) {
    if (!s->halo_pending) return FS_OK;
    fs3d_sim::Timed* t = timed_begin(s, s->stream, 1);
    SIM_HIP(hipStreamWaitEvent(s->stream, s->ev_comm, 0));
    timed_end(t, s->stream);
    ++s->cstats.waits;
    s->halo_pending = false;
    return FS_OK;
}

// Blocks the host until the handle's two streams are idle.
int quiesce(fs3d_sim* s) {
    FS_TRY(join_halos(s));
    SIM_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

// Queues an exchange behind everything the main stream holds so far.
int exchange_async(fs3d_sim* s, const std::vector<HaloOp>& ops) {
    SIM_HIP(hipEventRecord(s->ev_bnd, s->stream));
    SIM_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_bnd, 0));
    FS_TRY(exchange_halos(s, ops, s->comm_stream));
    SIM_HIP(hipEventRecord(s->ev_comm, s->comm_stream));
    s->halo_pending = true;
    return FS_OK;
}

// Brings the halo planes named in `need` (HALO_* bits) up to date if they lag.  With `async` the main
// stream is not made to wait for the exchange (see join_halos).
int refresh_halos(fs3d_sim* s, int need, bool async = false) {
    if (s->nparts_global == 1) { s->halo_dirty = 0; return FS_OK; }
    const int todo = s->halo_dirty & need;
    std::vector<HaloOp> ops;
    if (todo & HALO_VEL) s->deep_ok = false;   // (the owners' planes have changed since the halos were filled)
    if (todo & HALO_VEL) for (int k = 0; k < 3; ++k) ops.push_back({base_field_cur, k, s->esz});
    if (todo & HALO_DEN) ops.push_back({base_field_cur, 3, s->esz});
    if (todo & HALO_TMP) for (int k = 0; k < 3; ++k) ops.push_back({base_tmp, k, s->esz});
    if (!ops.empty()) FS_TRY(exchange_async(s, ops));
    s->halo_dirty &= ~todo;
    if (!async) FS_TRY(join_halos(s));
    return FS_OK;
}

// After sweeps: the one-pass sweeps leave the velocity halos exchanged, the two-pass and the
// lexicographic ones only keep the interface w plane current.
void halos_after_sweeps(fs3d_sim* s) {
    const bool one_pass = s->cfg.ordering == FS_ORDER_RED_BLACK && s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS;
    if (s->nparts_global > 1 && !one_pass) s->halo_dirty |= HALO_VEL;
}

int refresh_flags(fs3d_sim* s) {
    if (!s->cells_dirty) return FS_OK;
    FS_TRY(join_halos(s));
    const int depth = s->parts[0].H + s->parts[0].D;   // the deep planes too: the pass of two sweeps visits their cells
    FS_TRY(exchange_halos(s, {{base_cells, 0, 1, 0, depth}}));
    for (Part& p : s->parts) {
        const Grid dg = deep_grid(p);
        dim3 b(64, 4, 1), g(div_up(s->nx, 64), div_up(s->ny, 4), dg.nzl);
        hipLaunchKernelGGL(k_build_flags, g, b, 0, s->stream, dg, deep_ptr(p, p.cells), deep_ptr(p, p.flags), deep_ptr(p, p.vflags));
    }
    SIM_HIP(hipGetLastError());
    // the flags of a halo plane depend on cells beyond the halo: take them from the plane's owner
    FS_TRY(exchange_halos(s, {{base_flags, 0, 1, 0, depth}, {base_flags, 1, 1, 0, depth}}));
    s->cells_dirty = false;
    for (Part& p : s->parts) p.mg_weights_ok = false;
    return FS_OK;
}

// Everything a compute entry point needs before its first kernel.
int prepare(fs3d_sim* s, int need = HALO_VEL, bool async = false) {
    FS_TRY(refresh_flags(s));
    return refresh_halos(s, need, async);
}

// ---- projection -----------------------------------------------------------
// After a half-sweep of colour c the slab below an interface holds the fresh w
// where the cell under the interface has colour c, the slab above where the
// cell over it has; each takes the other half from its neighbour.
template <typename T>
int exchange_w_interfaces(fs3d_sim* s, int colour) {
    if (s->nparts_global == 1) return FS_OK;
    dim3 b(64, 4, 1), g(div_up(s->nx, 64), div_up(s->ny, 4), 1);
    // interfaces inside this process: read the neighbour's plane directly
    for (size_t k = 0; k + 1 < s->parts.size(); ++k) {
        Part& lo = s->parts[k];
        Part& hi = s->parts[k + 1];
        const int lz_lo = lo.H + lo.own_planes(), lz_hi = hi.H;
        T* wl = F<T>(lo, s->curv, 2);
        T* wh = F<T>(hi, s->curv, 2);
        // the cell over the interface (global plane z1 of `lo`) has parity (x+y+z1); the upper slab
        // updated the locations of parity == colour, the lower slab the others
        hipLaunchKernelGGL(k_merge_w_plane<T>, g, b, 0, s->stream, lo.g, wl, wh + (size_t)lz_hi * hi.g.sz, lz_lo, colour);
        hipLaunchKernelGGL(k_merge_w_plane<T>, g, b, 0, s->stream, hi.g, wh, wl + (size_t)lz_lo * lo.g.sz, lz_hi, colour ^ 1);
    }
    if (s->nranks == 1) { SIM_HIP(hipGetLastError()); return FS_OK; }
    Rccl* R = s->rccl;
    Part& first = s->parts.front();
    Part& last = s->parts.back();
    const size_t bytes = (size_t)first.g.sz * s->esz;
    FS_TRY(nccl_check(s, R->GroupStart(), "ncclGroupStart"));
    if (first.has_lo) {
        T* w = F<T>(first, s->curv, 2) + (size_t)first.H * first.g.sz;
        FS_TRY(nccl_check(s, R->Send(w, bytes, ncclUint8, s->rank - 1, s->comm, s->stream), "ncclSend"));
        FS_TRY(nccl_check(s, R->Recv(first.recv_lo, bytes, ncclUint8, s->rank - 1, s->comm, s->stream), "ncclRecv"));
    }
    if (last.has_hi) {
        T* w = F<T>(last, s->curv, 2) + (size_t)(last.H + last.own_planes()) * last.g.sz;
        FS_TRY(nccl_check(s, R->Send(w, bytes, ncclUint8, s->rank + 1, s->comm, s->stream), "ncclSend"));
        FS_TRY(nccl_check(s, R->Recv(last.recv_hi, bytes, ncclUint8, s->rank + 1, s->comm, s->stream), "ncclRecv"));
    }
    FS_TRY(nccl_check(s, R->GroupEnd(), "ncclGroupEnd"));
    if (first.has_lo)   // I am the upper slab of this interface: take what the lower slab updated
        hipLaunchKernelGGL(k_merge_w_plane<T>, g, b, 0, s->stream, first.g, F<T>(first, s->curv, 2), (const T*)first.recv_lo, first.H, colour ^ 1);
    if (last.has_hi)
        hipLaunchKernelGGL(k_merge_w_plane<T>, g, b, 0, s->stream, last.g, F<T>(last, s->curv, 2), (const T*)last.recv_hi, last.H + last.own_planes(), colour);
    SIM_HIP(hipGetLastError());
    return FS_OK;
}

template <typename T>
int launch_halfsweep(fs3d_sim* s, int colour) {
    const T omega = (T)s->cfg.omega;
    for (Part& p : s->parts) {
        if (p.ze - p.zb <= 0) continue;
        dim3 b(64, 4, 1), g(div_up(div_up(s->nx - 2, 2), 64), div_up(s->ny - 2, 4), p.ze - p.zb);
        hipLaunchKernelGGL(k_rb_halfsweep<T>, g, b, 0, s->stream, p.g, F<T>(p, s->curv, 0), F<T>(p, s->curv, 1),
                           F<T>(p, s->curv, 2), p.flags, colour, omega, p.zb, p.ze);
    }
    SIM_HIP(hipGetLastError());
    return exchange_w_interfaces<T>(s, colour);
}

// One full red-black sweep in a single pass over HBM: the kernels read the current velocity generation
// and write the other one, then the roles swap.  `launch(part, z0, z1, stream)` runs the kernel for the
// stored plane range [z0, z1).  Slabs recompute the two ghost planes below them (and one above) instead
// of exchanging per colour.  With more than one slab the two owned planes next to each interface are
// produced on the second stream, followed there by their exchange, while the main stream sweeps the
// interior of the slab; the main stream picks the halos up before anything reads them.
// The reflection needs the unprojected half-step velocity after the projection that follows it.  The first
// half advection leaves it in the generation it writes; a one-pass sweep reads one generation and writes the
// other, so the field survives the first sweep of that projection untouched.  Whoever is about to overwrite
// that generation (a second sweep, or the second half advection) first trades its velocity arrays for the
// scratch arrays: the field is then "in the scratch fields" without a copy, halo planes included.
void tilde_to_tmp(fs3d_sim* s) {
    if (s->tilde_gen < 0) return;
    for (Part& p : s->parts)
        for (int k = 0; k < 3; ++k) std::swap(p.f[s->tilde_gen][k], p.tmp[k]);
    s->halo_dirty &= ~HALO_TMP;   // its halo planes came with the exchange that followed the advection
    s->tilde_gen = -1;
}

template <typename Launch>
int one_pass_driver(fs3d_sim* s, Launch launch) {
    if (s->tilde_gen == (s->curv ^ 1)) tilde_to_tmp(s);
    if (s->nparts_global == 1) {
        for (Part& p : s->parts) launch(p, 0, p.g.nzl, s->stream, 0);
        SIM_HIP(hipGetLastError());
        s->curv ^= 1;
        return FS_OK;
    }
    // Sweep k on two streams.  comm_stream: B_k (the two owned planes next to each interface), then X_k
    // (their exchange).  Main stream: I_k (the interior).  I_k reads and overwrites owned planes only, so it
    // waits for B_{k-1} (which read the generation I_k overwrites and produced planes I_k reads) but not
    // for X_{k-1}; B_k reads halo planes and follows X_{k-1} in stream order.  Whoever needs the halos of
    // the last sweep calls join_halos.
    s->deep_ok = false;   // (this sweep's exchange fills the H halo planes only)
    SIM_HIP(hipEventRecord(s->ev_bnd, s->stream));               // I_{k-1} and everything queued before it
    SIM_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_bnd, 0));
    if (s->halo_pending) SIM_HIP(hipStreamWaitEvent(s->stream, s->ev_b, 0));   // B_{k-1}, or the exchange queued before sweep 1
    for (Part& p : s->parts) {
        const int own_lo = p.H, own_hi = p.H + p.own_planes();
        const int lo_end = p.has_lo ? std::min(own_lo + 2, own_hi) : own_lo;
        const int hi_beg = p.has_hi ? std::max(own_hi - 2, lo_end) : own_hi;
        // (a slab with two interfaces: the planes next to both in one launch)
        if (lo_end - own_lo == 2 && own_hi - hi_beg == 2 && hi_beg > lo_end) launch(p, own_lo, own_hi, s->comm_stream, 2);
        else {
            if (lo_end > own_lo) launch(p, own_lo, lo_end, s->comm_stream, 0);
            if (own_hi > hi_beg) launch(p, hi_beg, own_hi, s->comm_stream, 0);
        }
        if (hi_beg > lo_end) launch(p, lo_end, hi_beg, s->stream, 0);
    }
    SIM_HIP(hipGetLastError());
    SIM_HIP(hipEventRecord(s->ev_b, s->comm_stream));
    s->curv ^= 1;   // the halo operators below address the new generation
    std::vector<HaloOp> ops;
    for (int k = 0; k < 3; ++k) ops.push_back({base_field_cur, k, s->esz});
    FS_TRY(exchange_halos(s, ops, s->comm_stream));
    SIM_HIP(hipEventRecord(s->ev_comm, s->comm_stream));
    s->halo_pending = true;
    return FS_OK;
}

template <typename T>
FusedRanges fused_ranges(const Part& p) {
    FusedRanges r;
    r.s1b = p.zb - (p.has_lo ? 1 : 0); r.s1e = p.ze;
    r.s0b = p.zb - (p.has_lo ? 2 : 0); r.s0e = p.ze + (p.has_hi ? 1 : 0);
    r.z0 = 0; r.z1 = 0; r.zchunk = 1; r.zlen = 1;
    return r;
}

// z-chunks of one launch: pick the chunk count that minimises (rounds of workgroups over the CUs) x
// (planes a workgroup marches through, its overlap and pipeline planes included).
inline int pick_chunks(int planes, int tiles, int slots, int max_len = 1 << 30) {
    int best_c = 1;
    double best = 1e300;
    for (int c = div_up(planes, max_len); c <= std::max(1, planes / 2); ++c) {   // (chunks of two planes at least: small grids are bound by the steps of a workgroup)
        const int zc = div_up(planes, c);
        const int rounds = div_up(tiles * div_up(planes, zc), slots);
        const double cost = (double)rounds * (zc + 6);
        if (cost < best) { best = cost; best_c = c; }
    }
    return best_c;
}

// The LDS-window form (fs_sweep_fused.hpp); any nx.
template <typename T>
int launch_fused_sweep(fs3d_sim* s) {
    constexpr int TX = 64, TY = 8;
    using C = FusedTile<T, TX, TY>;
    auto kern = k_rb_fused<T, TX, TY>;
    SIM_HIP(allow_lds(s, (const void*)kern, C::lds_bytes()));
    const T omega = (T)s->cfg.omega;
    const int cv = s->curv, ov = s->curv ^ 1;
    const int tiles = div_up(s->nx, TX) * div_up(s->ny, TY);
    return one_pass_driver(s, [&](Part& p, int z0, int z1, hipStream_t st, int ends) {
        FusedRanges r = fused_ranges<T>(p);
        r.z0 = z0; r.z1 = z1;
        if (ends > 0) {   // two workgroups along z: the first and the last `ends` planes of [z0, z1)
            r.zchunk = z1 - ends - z0; r.zlen = ends;
        } else {
            const int nchunks = pick_chunks(z1 - z0, tiles, 2 * s->num_cus);
            r.zchunk = r.zlen = div_up(z1 - z0, nchunks);
        }
        dim3 b(256, 1, 1), g(div_up(s->nx, TX), div_up(s->ny, TY), ends > 0 ? 2 : div_up(z1 - z0, r.zchunk));
        hipLaunchKernelGGL(kern, g, b, C::lds_bytes(), st, p.g, F<T>(p, cv, 0), F<T>(p, cv, 1), F<T>(p, cv, 2),
                           F<T>(p, ov, 0), F<T>(p, ov, 1), F<T>(p, ov, 2), p.flags, omega, r);
    });
}

// The register-resident form (fs_sweep_rows.hpp); needs an even nx.  Measured at 512^3: the time of a
// launch follows the total number of (row, plane) steps, not the number of busy CUs, so the tile keeps
// the most rows a workgroup can have (the smallest share of halo rows) and the z-chunking only has to
// provide about one workgroup per CU.
// One launch of the rows kernel over `ntx` tiles of a row starting at column xbase (see the kernel for xedges).
template <typename T, int NSEG, int CP>
int launch_rows_tiles(fs3d_sim* s, Part& p, int z0, int z1, hipStream_t st, int ends, int ntx, int xbase, int xedges) {
    using C = RowsCfg<T, NSEG, CP>;
    auto kern = k_rb_rows<T, NSEG, CP>;
    SIM_HIP(allow_lds(s, (const void*)kern, C::lds_bytes()));
    const int cv = s->curv, ov = s->curv ^ 1;
    const int rows = std::min(C::RY, s->ny);
    const int nty = s->ny <= rows ? 1 : div_up(s->ny - 4, rows - 4);
    FusedRanges r = fused_ranges<T>(p);
    r.z0 = z0; r.z1 = z1;
    // the narrow variants fit 64 registers: two workgroups of 16 waves share a CU
    constexpr int per_cu = NSEG * CP * sizeof(T) <= 8 ? 2 : 1;
    int zgroups;
    if (ends > 0) {   // two workgroups along z: the first and the last `ends` planes of [z0, z1)
        r.zchunk = z1 - ends - z0; r.zlen = ends; zgroups = 2;
    } else {
        const int nchunks = pick_chunks(z1 - z0, ntx * nty, per_cu * s->num_cus);
        r.zchunk = r.zlen = div_up(z1 - z0, nchunks);
        zgroups = div_up(z1 - z0, r.zchunk);
    }
    dim3 b(64, rows, 1), g(ntx * nty * zgroups, 1, 1);
    hipLaunchKernelGGL(kern, g, b, C::lds_bytes(), st, p.g, F<T>(p, cv, 0), F<T>(p, cv, 1), F<T>(p, cv, 2), F<T>(p, ov, 0),
                       F<T>(p, ov, 1), F<T>(p, ov, 2), p.flags, p.vflags, (T)s->cfg.omega, r, ntx, nty, rows, xbase, xedges);
    return FS_OK;
}

// The register-resident form (fs_sweep_rows.hpp); needs an even nx.  Measured at 512^3: the time of a
// launch follows the total number of (row, plane) steps, not the number of busy CUs, so the tile keeps
// the most rows a workgroup can have (the smallest share of halo rows) and the z-chunking only has to
// provide about one workgroup per CU.  Tiles of W columns overlap by 4, so a row of nx columns takes
// ceil((nx - 4) / (W - 4)) of them: where the last of these would hold few columns (nx = 1024 is two tiles
// of 512 and four columns) it is given to a launch of a narrower variant instead, which marches through a
// quarter of the lanes (1.17 -> 0.97 ms per sweep at 1024 x 1024 x 128).
template <typename T, int NSEG, int CP>
int launch_rows_sweep(fs3d_sim* s) {
    using C = RowsCfg<T, NSEG, CP>;
    const int ntx = s->nx <= C::W ? 1 : div_up(s->nx - 4, C::W - 4);
    const int last_origin = (ntx - 1) * (C::W - 4), last_cols = s->nx - last_origin;
    const bool split = ntx >= 2 && C::W > 128 && last_cols <= C::W / 2;
    int rc = FS_OK;
    auto keep = [&](int r) { if (rc == FS_OK) rc = r; };
    const int drc = one_pass_driver(s, [&](Part& p, int z0, int z1, hipStream_t st, int ends) {
        if (!split) { keep(launch_rows_tiles<T, NSEG, CP>(s, p, z0, z1, st, ends, ntx, 0, 3)); return; }
        keep(launch_rows_tiles<T, NSEG, CP>(s, p, z0, z1, st, ends, ntx - 1, 0, 1));
        if (last_cols <= 128) keep(launch_rows_tiles<T, 1, 1>(s, p, z0, z1, st, ends, 1, last_origin, 2));
        else if (sizeof(T) == 4 && s->nx % 4 == 0) keep(launch_rows_tiles<T, 1, 2>(s, p, z0, z1, st, ends, 1, last_origin, 2));
        else keep(launch_rows_tiles<T, 2, 1>(s, p, z0, z1, st, ends, 1, last_origin, 2));
    });
    return rc != FS_OK ? rc : drc;
}

template <typename T>
int launch_one_pass_sweep(fs3d_sim* s) {
    if (s->nx % 2 == 0 && s->cfg.sweep_kernel == FS_SWEEP_FUSED) {
        // A lane moves 16 bytes per access where that is free: 8-byte elements, or rows of 4-byte elements of
        // at most 256 columns (104 instead of 113 us per sweep at 256^3).  At 512 columns the 16-byte form of
        // the 4-byte kernel spills more registers than the 8-byte form (five planes of u and w per column
        // against 128 registers) and runs 1.35 instead of 0.88 ms, so wide rows keep 8-byte accesses.
        if constexpr (sizeof(T) == 8) {
            if (s->nx <= 128) return launch_rows_sweep<T, 1, 1>(s);
            return launch_rows_sweep<T, 2, 1>(s);
        } else {
        if (s->nx % 4 == 0 && s->nx <= 256) return launch_rows_sweep<T, 1, 2>(s);
        if (s->nx % 4 == 0) return launch_rows_sweep<T, 2, 2>(s);
        if (s->nx <= 128) return launch_rows_sweep<T, 1, 1>(s);
        if (s->nx <= 256) return launch_rows_sweep<T, 2, 1>(s);
        return launch_rows_sweep<T, 4, 1>(s);
        }
    }
    return launch_fused_sweep<T>(s);
}

// Two sweeps in one pass over HBM (fs_sweep_rows2.hpp): even nx, the register-resident kernel; a single domain, or
// z-slabs that carry deep halo planes (every slab of the grid owns at least four planes).
bool pairs_possible(const fs3d_sim* s) {
    const char* no = getenv("FSAMD_NO_SWEEP_PAIRS");   // read per call: tests switch it between simulations
    const bool off = no && no[0] == '1';
    return !off && s->cfg.ordering == FS_ORDER_RED_BLACK && s->cfg.sweep_kernel == FS_SWEEP_FUSED && s->nx % 2 == 0 &&
           (s->nparts_global == 1 || s->deep);
}

// The launch of the two-sweep kernel for one choice of lanes per row: tiles of W columns x `rows` rows overlapping by 8,
// z-chunks of the `planes` planes it stores, and the number of steps the slowest CU marches (rounds of workgroups x
// planes of a workgroup: its chunk, the eight planes of overlap and the pipeline).
struct Rows2Plan { int waves, rows, ntx, nty, zchunk, zlen, zgroups; double steps; };

// ends > 0: the launch stores the first and the last `ends` planes of the `planes` it is given, one chunk each.
template <typename C>
Rows2Plan rows2_plan(const fs3d_sim* s, int planes, int per_cu, int ntx = 0, int ends = 0) {   // ntx: that many tiles of a row instead of all of it
    Rows2Plan t;
    // an even number of waves: the rows of a wave then have one colour
    t.waves = std::min(C::WAVES, 2 * div_up(div_up(s->ny, C::RPW), 2));
    t.rows = t.waves * C::RPW;
    t.nty = s->ny <= t.rows ? 1 : div_up(s->ny - 2 * C::S, t.rows - 2 * C::S);
    t.ntx = ntx > 0 ? ntx : s->nx <= C::W ? 1 : div_up(s->nx - 2 * C::S, C::W - 2 * C::S);
    int best_c = 1;
    t.steps = 1e300;
    for (int c = 1; c <= std::max(1, planes / 2); ++c) {   // (chunks of two planes at least: small grids are bound by the steps of a workgroup)
        const int zc = div_up(planes, c);
        const double cost = (double)div_up(t.ntx * t.nty * div_up(planes, zc), per_cu * s->num_cus) * (zc + 12);
        if (cost < t.steps) { t.steps = cost; best_c = c; }
    }
    t.zchunk = t.zlen = div_up(planes, best_c);
    t.zgroups = div_up(planes, t.zchunk);
    if (ends > 0) {
        t.zchunk = planes - ends; t.zlen = ends; t.zgroups = 2;
        t.steps = (double)div_up(t.ntx * t.nty * 2, per_cu * s->num_cus) * (ends + 12);
    }
    return t;
}

// One launch: the planes [z0, z1) of part p (local planes) are stored; the kernel reads four planes more on either
// side, which for a slab are its halo planes, the deep ones included.
template <typename T, int NSEG, int CP, int LPR>
int launch_rows2_tiles(fs3d_sim* s, Part& p, const Rows2Plan& t, int cv, int ov, int z0, int z1, hipStream_t st, int xbase = 0,
                       int xedges = 3) {
    using C = Rows2Cfg<T, NSEG, CP, LPR>;
    const void* kern = rows2_kernel<T, NSEG, CP, LPR>();   // (instantiated in fs_rows2.hip)
    SIM_HIP(allow_lds(s, kern, C::lds_bytes()));
    dim3 b(64, t.waves, 1), g(t.ntx * t.nty * t.zgroups, 1, 1);
    // the arguments of k_rb_rows2, in its order and types; plane numbers count from the first deep plane
    Grid grid = deep_grid(p);
    const T *ui = deep_ptr(p, F<T>(p, cv, 0)), *vi = deep_ptr(p, F<T>(p, cv, 1)), *wi = deep_ptr(p, F<T>(p, cv, 2));
    T *uo = deep_ptr(p, F<T>(p, ov, 0)), *vo = deep_ptr(p, F<T>(p, ov, 1)), *wo = deep_ptr(p, F<T>(p, ov, 2));
    const uint8_t* vflags = deep_ptr(p, p.vflags);
    T omega = (T)s->cfg.omega;
    // every plane of the arrays that is an interior plane of the grid is relaxed (a slab relaxes its halo planes too:
    // they go stale ring by ring, exactly like the overlap of two z-chunks)
    int zb = std::max(1 - grid.gz0, 0), ze = std::max(zb, std::min(s->nz - 1 - grid.gz0, grid.nzl));
    int zs0 = z0 + p.D, zs1 = z1 + p.D, zchunk = t.zchunk, zlen = t.zlen, ntx = t.ntx, nty = t.nty, rows = t.rows;
    void* args[] = {&grid, &ui, &vi, &wi, &uo, &vo, &wo, &vflags, &omega, &zb, &ze, &zs0, &zs1, &zchunk, &zlen, &ntx, &nty, &rows, &xbase, &xedges};
    SIM_HIP(hipLaunchKernel(kern, g, b, args, C::lds_bytes(), st));
    return FS_OK;
}

// The tile shape is chosen per grid.  A workgroup is 16 waves whatever its shape (256 x 16, 128 x 32 or 64 x 64 cells at
// 4-byte elements), and squarer tiles lose a smaller share of their cells to the stale rings; the estimate is the
// number of steps of the plan times the measured cost of a step of that shape: 2.85 us with one row per wave at any
// size (waves whose row is stale in a half-sweep skip it, and a wave reads a kilobyte of a row at a time), 3.2 to 4.7 us
// with two rows per wave and 3.7 to 7.7 us with four, growing with the length of a row between 1 and 2 KiB (the
// shorter pieces of longer rows are served worse by the memory system).
// Where the last 256-column tile of a row would hold at most 64 (128) columns - nx = 512 is two tiles and sixteen
// columns - those can go to a second launch of 64 x 64 (128 x 32) tiles instead: "64+16", "64+32".
// FSAMD_ROWS2_LANES_PER_ROW = 64 | 32 | 16 | 64+16 | 64+32 fixes the shape (tests, measurements).
// ends > 0: only the first and the last `ends` planes of [z0, z1) (the planes next to the two interfaces of a slab,
// in one launch: its workgroups run side by side instead of one launch after the other).
template <typename T, int NSEG, int CP>
int launch_rows2_range(fs3d_sim* s, Part& p, int z0, int z1, hipStream_t st, int ends = 0) {
    if (z1 <= z0) return FS_OK;
    const int cv = s->curv, ov = s->curv ^ 1;
    using C64 = Rows2Cfg<T, NSEG, CP, 64>;
    using C32 = Rows2Cfg<T, NSEG, CP, 32>;
    using C16 = Rows2Cfg<T, NSEG, CP, 16>;
    const int per_cu = 1, span = z1 - z0;   // (with ends, a plan covers the span and stores its two ends)
    const Rows2Plan t64 = rows2_plan<C64>(s, span, per_cu, 0, ends), t32 = rows2_plan<C32>(s, span, per_cu, 0, ends), t16 = rows2_plan<C16>(s, span, per_cu, 0, ends);
    const int last_origin = (t64.ntx - 1) * (C64::W - 2 * C64::S), last_cols = s->nx - last_origin;
    const int rest = t64.ntx < 2 ? 0 : last_cols <= C16::W ? 16 : last_cols <= C32::W ? 32 : 0;   // shape for the last columns
    const Rows2Plan main64 = rest ? rows2_plan<C64>(s, span, per_cu, t64.ntx - 1, ends) : t64;
    const Rows2Plan rest_plan = rest == 16 ? rows2_plan<C16>(s, span, per_cu, 1, ends) : rest == 32 ? rows2_plan<C32>(s, span, per_cu, 1, ends) : t64;
    enum { ONE64 = 64, ONE32 = 32, ONE16 = 16, SPLIT = 1 };
    int shape = 0;
    if (const char* want = getenv("FSAMD_ROWS2_LANES_PER_ROW")) {   // read per call: tests switch it between simulations
        const std::string w(want);
        shape = w == "64" ? ONE64 : w == "32" ? ONE32 : w == "16" ? ONE16 : (w == "64+16" && rest == 16) || (w == "64+32" && rest == 32) ? SPLIT : 0;
    }
    if (!shape) {
        const double over = std::min(1.0, std::max(0.0, ((double)s->nx * sizeof(T) - 1024.0) / 1024.0));
        const double k32 = 3.2 + 1.5 * over, k16 = 3.7 + 4.0 * over;
        const double c64 = 2.85 * t64.steps, c32 = k32 * t32.steps, c16 = k16 * t16.steps;
        const double csplit = rest ? 2.85 * main64.steps + (rest == 16 ? k16 : k32) * rest_plan.steps : 1e300;
        shape = csplit <= c64 && csplit <= c32 && csplit <= c16 ? SPLIT : c64 <= c32 && c64 <= c16 ? ONE64 : c32 <= c16 ? ONE32 : ONE16;
    }
    if (shape == SPLIT) {
        FS_TRY((launch_rows2_tiles<T, NSEG, CP, 64>(s, p, main64, cv, ov, z0, z1, st, 0, 1)));
        if (rest == 16) FS_TRY((launch_rows2_tiles<T, NSEG, CP, 16>(s, p, rest_plan, cv, ov, z0, z1, st, last_origin, 2)));
        else FS_TRY((launch_rows2_tiles<T, NSEG, CP, 32>(s, p, rest_plan, cv, ov, z0, z1, st, last_origin, 2)));
    } else if (shape == ONE64) FS_TRY((launch_rows2_tiles<T, NSEG, CP, 64>(s, p, t64, cv, ov, z0, z1, st)));
    else if (shape == ONE32) FS_TRY((launch_rows2_tiles<T, NSEG, CP, 32>(s, p, t32, cv, ov, z0, z1, st)));
    else FS_TRY((launch_rows2_tiles<T, NSEG, CP, 16>(s, p, t16, cv, ov, z0, z1, st)));
    return FS_OK;
}

// Brings the deep halo planes of the current velocity up to date (they lag after anything but a pass of two sweeps).
int refresh_deep_halos(fs3d_sim* s) {
    if (s->deep_ok && !(s->halo_dirty & HALO_VEL)) return FS_OK;
    const int H = s->parts[0].H, D = s->parts[0].D;
    // only the deep planes where the H halo planes are current
    const int d0 = (s->halo_dirty & HALO_VEL) ? 0 : H;
    std::vector<HaloOp> ops;
    for (int k = 0; k < 3; ++k) ops.push_back({base_field_cur, k, s->esz, d0, H + D});
    FS_TRY(exchange_async(s, ops));
    s->halo_dirty &= ~HALO_VEL;
    s->deep_ok = true;
    return FS_OK;
}

// One pass of two sweeps.  On z-slabs the schedule is that of the one-pass sweep (one_pass_driver) with everything twice
// as deep: a slab stores its owned planes and reads four planes beyond them, so the four owned planes next to each
// interface are produced on the second stream (B), followed there by the exchange of four planes of u, v and w with
// each neighbour (X), while the main stream sweeps the rest of the slab (I), which reads owned planes only.  One
// exchange per two sweeps.
template <typename T, int NSEG, int CP>
int launch_rows2(fs3d_sim* s) {
    if (s->tilde_gen == (s->curv ^ 1)) tilde_to_tmp(s);
    if (s->nparts_global == 1) {
        Part& p = s->parts[0];
        FS_TRY((launch_rows2_range<T, NSEG, CP>(s, p, 0, p.g.nzl, s->stream)));
        SIM_HIP(hipGetLastError());
        s->curv ^= 1;
        return FS_OK;
    }
    FS_TRY(refresh_deep_halos(s));
    SIM_HIP(hipEventRecord(s->ev_bnd, s->stream));               // I_{k-1} and everything queued before it
    SIM_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_bnd, 0));
    if (s->halo_pending) SIM_HIP(hipStreamWaitEvent(s->stream, s->ev_b, 0));   // B_{k-1} (see one_pass_driver)
    const int depth = s->parts[0].H + s->parts[0].D;
    for (Part& p : s->parts) {
        const int own_lo = p.H, own_hi = p.H + p.own_planes();
        const int lo_end = p.has_lo ? std::min(own_lo + depth, own_hi) : own_lo;
        const int hi_beg = p.has_hi ? std::max(own_hi - depth, lo_end) : own_hi;
        // (a slab with two interfaces: the planes next to both in one launch)
        if (lo_end - own_lo == depth && own_hi - hi_beg == depth && hi_beg >= lo_end)
            FS_TRY((launch_rows2_range<T, NSEG, CP>(s, p, own_lo, own_hi, s->comm_stream, depth)));
        else {
            FS_TRY((launch_rows2_range<T, NSEG, CP>(s, p, own_lo, lo_end, s->comm_stream)));
            FS_TRY((launch_rows2_range<T, NSEG, CP>(s, p, hi_beg, own_hi, s->comm_stream)));
        }
        FS_TRY((launch_rows2_range<T, NSEG, CP>(s, p, lo_end, hi_beg, s->stream)));
    }
    SIM_HIP(hipGetLastError());
    SIM_HIP(hipEventRecord(s->ev_b, s->comm_stream));
    s->curv ^= 1;   // the halo operators below address the new generation
    std::vector<HaloOp> ops;
    for (int k = 0; k < 3; ++k) ops.push_back({base_field_cur, k, s->esz, 0, depth});
    FS_TRY(exchange_halos(s, ops, s->comm_stream));
    SIM_HIP(hipEventRecord(s->ev_comm, s->comm_stream));
    s->halo_pending = true;
    s->deep_ok = true;
    return FS_OK;
}

// Two sweeps; the caller has checked pairs_possible().
template <typename T>
int sweep_pair(fs3d_sim* s) {
    if constexpr (sizeof(T) == 8) return launch_rows2<T, 1, 1>(s);
    else {
        if (s->nx % 4 == 0) return launch_rows2<T, 1, 2>(s);
        return launch_rows2<T, 1, 1>(s);
    }
}

template <typename T> int sweep_once(fs3d_sim* s);

// n sweeps without any evaluation, two per pass where that is possible.
template <typename T>
int sweeps_n(fs3d_sim* s, int n) {
    const bool pairs = pairs_possible(s);
    while (n > 0) {
        if (pairs && n >= 2) { FS_TRY(sweep_pair<T>(s)); n -= 2; }
        else { FS_TRY(sweep_once<T>(s)); n -= 1; }
    }
    return FS_OK;
}

template <typename T>
int sweep_once(fs3d_sim* s) {
    if (s->cfg.ordering == FS_ORDER_RED_BLACK && s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS) return launch_one_pass_sweep<T>(s);
    if (s->cfg.ordering == FS_ORDER_RED_BLACK) {
        FS_TRY(launch_halfsweep<T>(s, 0));
        return launch_halfsweep<T>(s, 1);
    }
    // the reference's lexicographic order: one launch per diagonal of 16^3-cell blocks (single part only)
    Part& p = s->parts[0];
    using C = LexiBlock<T>;
    SIM_HIP(allow_lds(s, (const void*)k_lexi_blocks<T>, C::lds_bytes()));
    const int nbx = div_up(s->nx - 2, C::LB), nby = div_up(s->ny - 2, C::LB), nbz = div_up(s->nz - 2, C::LB);
    for (int K = 0; K <= nbx + nby + nbz - 3; ++K)
        hipLaunchKernelGGL(k_lexi_blocks<T>, dim3(nby, nbz, 1), dim3(256), C::lds_bytes(), s->stream, p.g, F<T>(p, s->curv, 0),
                           F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), p.flags, K, nbx, (T)s->cfg.omega);
    SIM_HIP(hipGetLastError());
    return FS_OK;
}

// Combines the per-part (sum, max) pairs, across ranks too, into host values.
// out[0] = sum, out[1] = 1 if this process's out-of-halo flag is up (added over the ranks with the sum: every rank
// then sees the same count), out[2] = max.
__global__ void k_combine_parts(const double* __restrict__ dres, int nparts, const int* __restrict__ oob,
                                double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double sum = 0, mx = 0;
        for (int k = 0; k < nparts; ++k) { sum += dres[2 * k]; mx = fmax(mx, dres[2 * k + 1]); }
        out[0] = sum; out[1] = *oob ? 1.0 : 0.0; out[2] = mx;
    }
}

// The combined record written into the pinned host mirror, and a ticket behind it that the host can poll.
__global__ void k_publish_record(const double* __restrict__ comb, volatile double* __restrict__ host, double ticket) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        host[0] = comb[0]; host[1] = comb[1]; host[2] = comb[2];
        __threadfence_system();
        host[3] = ticket;
    }
}

// An advection of a slab does not wait for its out-of-halo flag; the flag comes back with the next
// reduction result or synchronisation and stays set until the handle is reset or loaded.  The flag is part of
// what the ranks add up, so every rank reports the failure at the same call (a rank that stopped alone would
// leave the others waiting in their next exchange).
int oob_status(fs3d_sim* s) {
    if (s->nparts_global > 1 && s->hres[1] > 0.0)
        return set_err(s, FS_ESTATE, "a backtrace left the slab halo (|v|*dt > 2 cells) on this or another rank");
    return FS_OK;
}

// Queues, on the main stream, the combination of the per-part pairs (across ranks too) and their copy to
// the host mirror.
int queue_reduction(fs3d_sim* s, bool want_max) {
    // one stream at a time on a communicator: with a communicator of their own the reductions run beside
    // the halo traffic of the other stream (every rank issues the two in the same order)
    if (s->nranks > 1 && s->comm_red == s->comm) FS_TRY(join_halos(s));
    double* comb = s->dres + 2 * s->nlocal;
    hipLaunchKernelGGL(k_combine_parts, dim3(1), dim3(64), 0, s->stream, s->dres, s->nlocal, s->doob, comb);
    SIM_HIP(hipGetLastError());
    if (s->nranks > 1) {
        // the sum and the out-of-halo count travel together
        FS_TRY(nccl_check(s, s->rccl->AllReduce(comb, comb, 2, ncclDouble, ncclSum, s->comm_red, s->stream), "ncclAllReduce"));
        if (want_max)   // every rank makes the same sequence of calls, so all of them ask for the maximum or none does
            FS_TRY(nccl_check(s, s->rccl->AllReduce(comb + 2, comb + 2, 1, ncclDouble, ncclMax, s->comm_red, s->stream), "ncclAllReduce"));
    }
    // the record goes to the pinned host mirror by a kernel that writes a ticket behind it (see k_reduce_to_host)
    s->ticket += 1.0;
    hipLaunchKernelGGL(k_publish_record, dim3(1), dim3(64), 0, s->stream, comb, s->hres, s->ticket);
    SIM_HIP(hipGetLastError());
    return FS_OK;
}

// One part on one rank (the usual single-GPU case): the partials of a reduction kernel are added up, combined and
// written into the pinned host record by ONE kernel - in the order and with the operations of k_reduce_final followed
// by k_combine_parts - instead of two kernels and a copy, and the host does not wait for the stream to drain: the
// kernel writes a ticket behind the values (host[3], after a system-scope fence) and the host polls the pinned word.
// A stream synchronisation sleeps and wakes on an interrupt, which leaves the GPU idle for about 30 us per
// evaluation of the stopping rule (a tenth of the 256^3 step); the poll sees the ticket within a few microseconds.
__global__ void __launch_bounds__(256) k_reduce_to_host(const double* __restrict__ part, int nblocks, const int* __restrict__ oob,
                                                        double* __restrict__ dres, volatile double* __restrict__ host, double ticket) {
    double sm = 0, m = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { sm += part[2 * b]; m = fmax(m, part[2 * b + 1]); }
    block_sum_max(sm, m);
    if (threadIdx.x == 0) {
        dres[0] = sm; dres[1] = m;
        double sum = 0, mx = 0;
        sum += sm; mx = fmax(mx, m);
        const double flag = *oob ? 1.0 : 0.0;
        dres[2] = sum; dres[3] = flag; dres[4] = mx;
        host[0] = sum; host[1] = flag; host[2] = mx;
        __threadfence_system();
        host[3] = ticket;
    }
}

bool single_part(const fs3d_sim* s) { return s->nlocal == 1 && s->nranks == 1; }

// Queues the reduction of part 0's partials into the host record (single-part handles).
int queue_reduction_single(fs3d_sim* s, const double* partials, int nblocks) {
    s->ticket += 1.0;
    hipLaunchKernelGGL(k_reduce_to_host, dim3(1), dim3(256), 0, s->stream, partials, nblocks, s->doob, s->dres, s->hres, s->ticket);
    SIM_HIP(hipGetLastError());
    return FS_OK;
}

// Waits for the record queued last: polls the ticket for a while, then falls back to a stream synchronisation (which
// also reports a failed launch or kernel).
int wait_host_record(fs3d_sim* s) {
    const auto t0 = std::chrono::steady_clock::now();
    volatile double* h = s->hres;
    const char* np = getenv("FSAMD_NO_POLL");   // (the polling thread keeps a core busy; read per call: tests switch it between simulations)
    const bool no_poll = np && np[0] == '1';
    bool seen = false;
    if (no_poll) { SIM_HIP(hipStreamSynchronize(s->stream)); seen = true; }
    for (int spin = 0; !seen; ++spin) {
        seen = h[3] == s->ticket;
        if (!seen && (spin & 1023) == 1023 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 20.0) break;
    }
    if (!seen) {
        if (!no_poll) ++s->cstats.poll_fallbacks;   // the ticket did not show within 20 ms: a timing regression, not an error
        SIM_HIP(hipStreamSynchronize(s->stream));
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->cstats.reduction_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++s->cstats.reductions;
    return oob_status(s);
}

int finish_reduction_single(fs3d_sim* s, const double* partials, int nblocks, double* sum, double* mx) {
    FS_TRY(queue_reduction_single(s, partials, nblocks));
    FS_TRY(wait_host_record(s));
    if (sum) *sum = s->hres[0];
    if (mx) *mx = s->hres[2];
    return FS_OK;
}

int finish_reduction(fs3d_sim* s, double* sum, double* mx) {
    FS_TRY(queue_reduction(s, mx != nullptr));
    FS_TRY(join_halos(s));
    FS_TRY(wait_host_record(s));
    if (sum) *sum = s->hres[0];
    if (mx) *mx = s->hres[2];
    return FS_OK;
}

template <typename T>
int snapshot_velocity(fs3d_sim* s) {   // main.cpp:192-196; the other generation serves as the snapshot
    for (Part& p : s->parts)
        for (int k = 0; k < 3; ++k)
            SIM_HIP(hipMemcpyAsync(p.f[s->curv ^ 1][k], p.f[s->curv][k], p.n * s->esz, hipMemcpyDeviceToDevice, s->stream));
    return FS_OK;
}

template <typename T>
int change_since_snapshot(fs3d_sim* s, double* change) {   // main.cpp:243-249
    ++s->stats.evaluations_step;
    FS_TRY(join_halos(s));   // the planes next to a slab interface were swept on the second stream
    if (s->cfg.ordering == FS_ORDER_LEXICOGRAPHIC) {
        Part& p = s->parts[0];
        hipLaunchKernelGGL(k_change_serial<T>, dim3(1), dim3(64), 0, s->stream, p.g, F<T>(p, s->curv ^ 1, 0),
                           F<T>(p, s->curv ^ 1, 1), F<T>(p, s->curv ^ 1, 2), F<T>(p, s->curv, 0), F<T>(p, s->curv, 1),
                           F<T>(p, s->curv, 2), s->dres);
        SIM_HIP(hipGetLastError());
        SIM_HIP(hipMemcpyAsync(s->hres, s->dres, sizeof(double), hipMemcpyDeviceToHost, s->stream));
        FS_TRY(quiesce(s));
        *change = s->hres[0];
        return FS_OK;
    }
    for (int k = 0; k < s->nlocal; ++k) {
        Part& p = s->parts[k];
        hipLaunchKernelGGL(k_change_l1<T>, dim3(kReduceBlocks), dim3(256), 0, s->stream, p.g, F<T>(p, s->curv ^ 1, 0),
                           F<T>(p, s->curv ^ 1, 1), F<T>(p, s->curv ^ 1, 2), F<T>(p, s->curv, 0), F<T>(p, s->curv, 1),
                           F<T>(p, s->curv, 2), p.zb, p.ze, p.partials);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s->stream, p.partials, kReduceBlocks, s->dres + 2 * k);
    }
    SIM_HIP(hipGetLastError());
    return finish_reduction(s, change, nullptr);
}

template <typename T>
int residual(fs3d_sim* s, double* l2, double* max_abs, bool store_div) {
    ++s->stats.evaluations_step;
    FS_TRY(join_halos(s));   // the top owned plane reads w of the first halo plane
    for (int k = 0; k < s->nlocal; ++k) {
        Part& p = s->parts[k];
        hipLaunchKernelGGL(k_div_sumsq<T>, dim3(kReduceBlocks), dim3(256), 0, s->stream, p.g, F<T>(p, s->curv, 0),
                           F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), p.flags, p.zb, p.ze, p.partials,
                           store_div ? (T*)p.tmp[0] : (T*)nullptr);
        if (!single_part(s))
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s->stream, p.partials, kReduceBlocks, s->dres + 2 * k);
    }
    SIM_HIP(hipGetLastError());
    double sum = 0, mx = 0;
    if (single_part(s)) FS_TRY(finish_reduction_single(s, s->parts[0].partials, kReduceBlocks, &sum, max_abs ? &mx : nullptr));
    else FS_TRY(finish_reduction(s, &sum, max_abs ? &mx : nullptr));
    const double ninterior = (double)(s->nx - 2) * (s->ny - 2) * (s->nz - 2);
    if (l2) *l2 = sqrt(sum) / ninterior;   // 3_fluid/FluidSim.cpp:40-43
    if (max_abs) *max_abs = mx;
    return FS_OK;
}

// ---- multigrid V-cycle (fs_mg3d.hpp) ---------------------------------------------------------------
constexpr int kMgNu1 = 2, kMgNu2 = 2, kMgBottom = 30;
constexpr double kMgAlpha = 2.0;
constexpr int64_t kMgSmallCells = 4096;   // levels up to this size run inside one workgroup

template <typename T>
MgLevel<T> mg_level(const Part::MgRaw& r) {
    MgLevel<T> L;
    L.d = MgDims{r.nx, r.ny, r.nz};
    L.wx = (T*)r.a[0]; L.wy = (T*)r.a[1]; L.wz = (T*)r.a[2]; L.diag = (T*)r.a[3]; L.e = (T*)r.a[4]; L.r = (T*)r.a[5];
    return L;
}

// The coarse levels span the whole grid and live with the first local part.  With several slabs every rank
// holds (and solves) all of them: a slab contributes the planes of level 1 that its own fine planes make up
// (weights once, right-hand side per cycle), ranks swap those planes, and everything below level 1 is
// computed redundantly, identically, everywhere.  The fine level, where the work is, stays distributed.

// A slab's level-1 cells must be made of its own fine cells: every slab starts on an even plane.
inline bool mg_slabs_ok(int nz, int nparts) {
    for (int gp = 0; gp < nparts; ++gp) {
        int z0, z1;
        split_planes(nz, nparts, gp, &z0, &z1);
        if (z0 % 2 != 0) return false;
    }
    return true;
}

// Level-1 planes made of the fine planes [z0, z1) (z0 even).
inline void mg_coarse_planes(int z0, int z1, int* c0, int* c1) { *c0 = z0 / 2; *c1 = (z1 + 1) / 2; }

// Every rank sends its planes of a level-1 array to every other rank and receives theirs in place.
int mg_allgather_level1(fs3d_sim* s, void* base, const Part::MgRaw& L) {
    if (s->nranks == 1) return FS_OK;
    FS_TRY(join_halos(s));   // one stream at a time on the communicator
    Rccl* R = s->rccl;
    const size_t plane = (size_t)L.nx * L.ny * s->esz;
    auto range_of = [&](int rank, int* c0, int* c1) {
        int z0, z1, t;
        split_planes(s->nz, s->nparts_global, rank * s->nlocal, &z0, &t);
        split_planes(s->nz, s->nparts_global, (rank + 1) * s->nlocal - 1, &t, &z1);
        mg_coarse_planes(z0, z1, c0, c1);
    };
    int m0, m1;
    range_of(s->rank, &m0, &m1);
    FS_TRY(nccl_check(s, R->GroupStart(), "ncclGroupStart"));
    for (int r = 0; r < s->nranks; ++r) {
        if (r == s->rank) continue;
        int c0, c1;
        range_of(r, &c0, &c1);
        FS_TRY(nccl_check(s, R->Send((char*)base + (size_t)m0 * plane, (size_t)(m1 - m0) * plane, ncclUint8, r, s->comm, s->stream), "ncclSend"));
        FS_TRY(nccl_check(s, R->Recv((char*)base + (size_t)c0 * plane, (size_t)(c1 - c0) * plane, ncclUint8, r, s->comm, s->stream), "ncclRecv"));
    }
    return nccl_check(s, R->GroupEnd(), "ncclGroupEnd");
}

template <typename T>
int mg_prepare(fs3d_sim* s) {
    Part& p = s->parts[0];
    auto grid_of = [](const Part::MgRaw& r, dim3* b, dim3* g) { *b = dim3(64, 4, 1); *g = dim3(div_up(r.nx, 64), div_up(r.ny, 4), r.nz); };
    if (p.mg.empty()) {
        for (Part& q : s->parts)
            if (q.z0 % 2 != 0) return set_err(s, FS_EINVAL, "the multigrid solver needs slabs that start on even planes");
        int nx = s->nx, ny = s->ny, nz = s->nz;
        while (p.mg.size() < 16) {
            nx = (nx + 1) / 2; ny = (ny + 1) / 2; nz = (nz + 1) / 2;
            Part::MgRaw r{nx, ny, nz, {nullptr}};
            for (void*& a : r.a) {
                SIM_HIP(hipMalloc(&a, (size_t)nx * ny * nz * sizeof(T)));
                SIM_HIP(hipMemsetAsync(a, 0, (size_t)nx * ny * nz * sizeof(T), s->stream));
            }
            p.mg.push_back(r);
            if (nx <= 4 || ny <= 4 || nz <= 4) break;
        }
        p.mg_weights_ok = false;
    }
    if (p.mg_weights_ok) return FS_OK;
    for (Part& q : s->parts) {   // level-1 weights of the planes this slab makes up (its cell halos are current)
        int c0, c1;
        mg_coarse_planes(q.z0, q.z1, &c0, &c1);
        const Part::MgRaw& r = p.mg[0];
        dim3 b(64, 4, 1), g(div_up(r.nx, 64), div_up(r.ny, 4), c1 - c0);
        hipLaunchKernelGGL(k_mg_weights_cells<T>, g, b, 0, s->stream, q.g, q.cells, mg_level<T>(r), c0);
    }
    SIM_HIP(hipGetLastError());
    for (int k = 0; k < 3; ++k) FS_TRY(mg_allgather_level1(s, p.mg[0].a[k], p.mg[0]));
    for (size_t l = 0; l < p.mg.size(); ++l) {
        dim3 b, g;
        grid_of(p.mg[l], &b, &g);
        if (l > 0) hipLaunchKernelGGL(k_mg_weights_level<T>, g, b, 0, s->stream, mg_level<T>(p.mg[l - 1]), mg_level<T>(p.mg[l]));
        hipLaunchKernelGGL(k_mg_diag<T>, g, b, 0, s->stream, mg_level<T>(p.mg[l]));
    }
    SIM_HIP(hipGetLastError());
    p.mg_weights_ok = true;
    return FS_OK;
}

template <typename T>
int mg_smooth(fs3d_sim* s, const Part::MgRaw& r, int sweeps) {
    dim3 b(64, 4, 1), g(div_up(div_up(r.nx, 2), 64), div_up(r.ny, 4), r.nz);
    for (int it = 0; it < sweeps; ++it)
        for (int colour = 0; colour < 2; ++colour)
            hipLaunchKernelGGL(k_mg_smooth<T>, g, b, 0, s->stream, mg_level<T>(r), colour);
    return FS_OK;
}

template <typename T>
int mg_coarse_cycle(fs3d_sim* s, Part& p, size_t l) {
    const Part::MgRaw& r = p.mg[l];
    dim3 b(64, 4, 1), g(div_up(r.nx, 64), div_up(r.ny, 4), r.nz);
    if ((int64_t)r.nx * r.ny * r.nz <= kMgSmallCells) {   // this level and everything below it in one workgroup
        MgSmall<T, 8> S;
        S.n = 0;
        for (size_t k = l; k < p.mg.size() && S.n < 8; ++k) S.L[S.n++] = mg_level<T>(p.mg[k]);
        if ((size_t)S.n == p.mg.size() - l) {
            const size_t lds = mg_small_lds_bytes(S);
            SIM_HIP(allow_lds(s, (const void*)k_mg_small_cycle<T, 8>, lds));
            hipLaunchKernelGGL((k_mg_small_cycle<T, 8>), dim3(1), dim3(1024), lds, s->stream, S, kMgNu1, kMgNu2, kMgBottom, (T)kMgAlpha);
            return FS_OK;
        }
    }
    SIM_HIP(hipMemsetAsync(r.a[4], 0, (size_t)r.nx * r.ny * r.nz * sizeof(T), s->stream));
    if (l + 1 == p.mg.size()) return mg_smooth<T>(s, r, kMgBottom);
    FS_TRY(mg_smooth<T>(s, r, kMgNu1));
    const Part::MgRaw& c = p.mg[l + 1];
    dim3 gc(div_up(c.nx, 64), div_up(c.ny, 4), c.nz);
    hipLaunchKernelGGL(k_mg_restrict<T>, gc, b, 0, s->stream, mg_level<T>(r), mg_level<T>(c));
    FS_TRY(mg_coarse_cycle<T>(s, p, l + 1));
    hipLaunchKernelGGL(k_mg_prolong<T>, g, b, 0, s->stream, mg_level<T>(r), mg_level<T>(c), (T)kMgAlpha);
    return mg_smooth<T>(s, r, kMgNu2);
}

// One V-cycle on the current velocity (red-black order).
template <typename T>
int mg_vcycle(fs3d_sim* s) {
    Part& p = s->parts[0];
    const bool slabs = s->nparts_global > 1;
    const bool one_pass = s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS;
    FS_TRY(mg_prepare<T>(s));
    // the cycle smooths with plain Gauss-Seidel: over-relaxation belongs to the sweeps-only solver
    struct OmegaOne {
        fs3d_sim* s; double saved;
        explicit OmegaOne(fs3d_sim* sim) : s(sim), saved(sim->cfg.omega) { s->cfg.omega = 1.0; }
        ~OmegaOne() { s->cfg.omega = saved; }
    } plain(s);
    FS_TRY(sweeps_n<T>(s, kMgNu1));
    if (slabs) {   // the residual of a slab's top plane reads w of its first halo plane
        halos_after_sweeps(s);
        FS_TRY(refresh_halos(s, HALO_VEL));
    }
    const Part::MgRaw& c = p.mg[0];
    dim3 b(64, 4, 1);
    for (Part& q : s->parts) {
        int c0, c1;
        mg_coarse_planes(q.z0, q.z1, &c0, &c1);
        dim3 gc(div_up(c.nx, 64), div_up(c.ny, 4), c1 - c0);
        hipLaunchKernelGGL(k_mg_fine_residual<T>, gc, b, 0, s->stream, q.g, F<T>(q, s->curv, 0), F<T>(q, s->curv, 1),
                           F<T>(q, s->curv, 2), q.flags, mg_level<T>(c), c0);
    }
    SIM_HIP(hipGetLastError());
    FS_TRY(mg_allgather_level1(s, c.a[5], c));
    FS_TRY(mg_coarse_cycle<T>(s, p, 0));
    for (Part& q : s->parts) {
        if (q.ze <= q.zb) continue;
        dim3 gf(div_up(s->nx - 2, 64), div_up(s->ny - 2, 4), q.ze - q.zb);
        hipLaunchKernelGGL(k_mg_apply<T>, gf, b, 0, s->stream, q.g, F<T>(q, s->curv, 0), F<T>(q, s->curv, 1), F<T>(q, s->curv, 2),
                           q.flags, mg_level<T>(c), (T)kMgAlpha, q.zb);
    }
    SIM_HIP(hipGetLastError());
    if (slabs) {   // the corrected planes next to the interfaces go to the neighbours before the sweeps read them
        s->halo_dirty |= HALO_VEL;
        FS_TRY(refresh_halos(s, HALO_VEL, one_pass));
    }
    return sweeps_n<T>(s, kMgNu2);
}

// One sweep with the reference's change metric (main.cpp:189-254).
template <typename T>
int sweep_with_change(fs3d_sim* s, double* change) {
    // the fused sweep leaves the previous field in the other generation: that is the snapshot
    const bool fused = s->cfg.ordering == FS_ORDER_RED_BLACK && s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS;
    if (!fused) FS_TRY(snapshot_velocity<T>(s));
    FS_TRY(sweep_once<T>(s));
    return change_since_snapshot<T>(s, change);
}

// The loop of update() (main.cpp:105-111) under the configured stopping rule.
// check_every == 0 selects the adaptive schedule: the metric is evaluated from one sweep before the count the
// same projection of the previous step needed, then after every sweep (consecutive steps need nearly the
// same number of sweeps, so this costs one or two evaluations and stops at the exact sweep), and twice on
// the way there.
// unchecked > 0: exactly that many sweeps and no evaluation (the caller verifies the stopping rule afterwards);
// resume_from > 0: the projection already holds that many sweeps and goes on, evaluating after every sweep;
// deferred != nullptr: if the evaluation one sweep before the expected count fails, the expected last sweep
// runs unverified and *deferred is set: the caller verifies the stopping rule (and resumes if it is not met).
template <typename T>
int project(fs3d_sim* s, int* sweeps_out, double* metric_out, int slot = 0, int unchecked = 0, int resume_from = 0,
            bool* deferred = nullptr) {
    // the interior of a one-pass sweep reads no halo plane: let the exchange run beside it
    const bool one_pass = s->cfg.ordering == FS_ORDER_RED_BLACK && s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS;
    FS_TRY(prepare(s, HALO_VEL, one_pass));
    const fs3d_config& c = s->cfg;
    const bool adaptive = c.check_every <= 0;
    const int every = std::max(1, c.check_every);
    const int hint = s->sweep_hint[slot & 1];
    const int first = resume_from > 0 ? resume_from + 1 : adaptive ? std::max(1, hint - 1) : 1;
    // two early evaluations, at a quarter and at half of the expected count: a count that collapses (the step
    // after a start-up transient) is then found within a step or two, and a stale hint cannot cost much
    // past the expected count the evaluations thin out (every sweep for four sweeps, then one sweep in five
    // of the excess): a projection that runs far beyond its hint, such as the first of a run, is not charged
    // an evaluation per sweep, and overshoots its stopping sweep by at most a quarter of that excess
    // (not while the count is steady: within an eighth of the one before)
    const bool steady = std::abs(hint - s->sweep_hint_prev[slot & 1]) <= std::max(1, hint / 8);
    const int early1 = steady ? 0 : first / 4, early2 = steady ? 0 : first / 2;
    auto check_after = [&](int done) {   // the sweep count of the next evaluation once `done` sweeps are behind
        if (done < early1) return early1;
        if (done < early2) return early2;
        if (done < first) return first;
        return done + 1 + (done - first) / 4;
    };
    int next_check = check_after(resume_from);
    int it = resume_from;
    double metric = 0;
    int last_it = -1;          // the last failed evaluation of this projection
    double last_metric = 0;
    const bool pairs = pairs_possible(s);
    if (unchecked > 0) {
        FS_TRY(sweeps_n<T>(s, unchecked));
        s->stats.sweeps = unchecked;
        s->stats.sweeps_step += unchecked;
        halos_after_sweeps(s);
        return FS_OK;
    }
    if (c.solver == FS_SOLVER_MULTIGRID) {
        // V-cycles; the stopping rule is evaluated after every cycle, a cycle counts as its fine sweeps.
        // A cycle costs more than four sweeps, so where sweeps alone may do (the same projection of the
        // previous step needed at most two of them, or a single cycle: worth a try now and then) up to
        // two plain sweeps are tried first.
        bool done = false;
        int& backoff = s->mg_backoff[slot & 1];
        const bool try_plain = c.stop_mode == FS_STOP_L2_RESIDUAL && c.max_iters > 2 &&
                               (hint <= 2 || (hint <= kMgNu1 + kMgNu2 + 2 && backoff == 0));
        if (backoff > 0) --backoff;
        if (try_plain) {
            for (int k = 0; k < 2 && !done; ++k) {
                FS_TRY(sweep_once<T>(s));
                ++it;
                FS_TRY(residual<T>(s, &metric, nullptr, false));
                metric /= s->resid_dt;
                done = !(metric > c.tol);
            }
            // a failed try is paid for by waiting: 2, 4, 8, 8, ... projections
            int& fails = s->mg_fails[slot & 1];
            fails = done ? 0 : std::min(fails + 1, 3);
            if (!done) backoff = 1 << fails;
        }
        while (!done) {
            FS_TRY(mg_vcycle<T>(s));
            it += kMgNu1 + kMgNu2;
            if (c.stop_mode == FS_STOP_L2_RESIDUAL) {
                FS_TRY(residual<T>(s, &metric, nullptr, false));
                metric /= s->resid_dt;
                if (!(metric > c.tol)) break;
            }
            if (it >= c.max_iters) break;
        }
    } else
    for (;;) {
        if (deferred && it == first && it + 1 == hint && it + 1 < c.max_iters) {
            // the evaluation at `first` has just failed: the sweep the hint expects to be the last
            FS_TRY(sweep_once<T>(s));
            ++it;
            *deferred = true;
            break;
        }
        auto is_check = [&](int n) { return (adaptive ? n == next_check : (n % every == 0)) || n >= c.max_iters; };
        bool check = is_check(it + 1);
        // two sweeps in one pass where the first of them is not evaluated (and, for the change of one sweep, the
        // second is not either: that metric needs the field one sweep back)
        if (pairs && !check && it + 2 <= c.max_iters && !(c.stop_mode == FS_STOP_L1_CHANGE && is_check(it + 2))) {
            FS_TRY(sweep_pair<T>(s));
            it += 2;
            check = is_check(it);
        } else {
            if (c.stop_mode == FS_STOP_L1_CHANGE && check) {
                FS_TRY(sweep_with_change<T>(s, &metric));
            } else {
                FS_TRY(sweep_once<T>(s));
            }
            ++it;
        }
        if (check) next_check = check_after(it);
        if (c.stop_mode == FS_STOP_L1_CHANGE && check && !(metric > c.tol)) break;
        if (c.stop_mode == FS_STOP_L2_RESIDUAL && check) {
            FS_TRY(residual<T>(s, &metric, nullptr, false));
            metric /= s->resid_dt;   // b = -div/dt (3_fluid/FluidSim.cpp:17), dx = rho = 1
            if (!(metric > c.tol)) break;
            // Two failed evaluations give a decay rate, and plain Gauss-Seidel only slows down as it goes (the smooth
            // error components outlive the others), so the rate promises at least n more sweeps: the next evaluation
            // can wait for most of them.  Close to the tolerance n falls below 2 and every sweep is evaluated again,
            // so the projection still stops on its first sweep at or below the tolerance.  Over-relaxed sweeps can
            // plateau and then speed up, so with omega != 1 nothing is extrapolated.
            if (adaptive && !deferred && c.omega == 1.0 && last_it >= 0 && metric < last_metric && metric > 0) {
                const double n = log(c.tol / metric) / (log(metric / last_metric) / (it - last_it));
                if (n >= 2.5) next_check = std::max(next_check, (int)std::min<double>(c.max_iters, it + floor(0.8 * n)));
            }
            last_it = it;
            last_metric = metric;
        }
        if (it >= c.max_iters) break;
    }
    // a projection that ran into the cap says nothing about the next one: start checking at once
    const bool capped = it >= c.max_iters && metric > c.tol;
    s->sweep_hint_prev[slot & 1] = (resume_from > 0 || capped) ? 0 : hint;   // (mispredicted or capped: not steady)
    s->sweep_hint[slot & 1] = capped ? 1 : it;
    s->stats.sweeps = it;
    s->stats.sweeps_step += it - resume_from;
    s->stats.metric = metric;
    halos_after_sweeps(s);
    if (sweeps_out) *sweeps_out = it;
    if (metric_out) *metric_out = metric;
    return FS_OK;
}

// ---- advection / reflection ------------------------------------------------
// Carries q = (qu,qv,qw from `qsrc`, density of the current generation) along the current
// generation's velocity into the other generation, then swaps generations.
template <typename T, typename OFF, bool SLAB>
int launch_advect(fs3d_sim* s, Part& p, const T* qu, const T* qv, const T* qw, int cv, int ov, int cd, int od, T t_step,
                  int density_mode, bool copy_to_tmp, bool reflect, bool with_res, int zb, int ze, int ends = 0) {
    if (ze <= zb) return FS_OK;
    T* c0 = copy_to_tmp ? (T*)p.tmp[0] : nullptr;
    T* c1 = copy_to_tmp ? (T*)p.tmp[1] : nullptr;
    T* c2 = copy_to_tmp ? (T*)p.tmp[2] : nullptr;
    const T *U = F<T>(p, cv, 0), *V = F<T>(p, cv, 1), *W = F<T>(p, cv, 2);
    if (s->cfg.advect_kernel == FS_ADVECT_DIRECT) {
        if (with_res) return set_err(s, FS_ESTATE, "only the staged advection kernel evaluates the residual");
        if (ends > 0) {   // one thread per location and no march along z: two plain launches
            FS_TRY((launch_advect<T, OFF, SLAB>(s, p, qu, qv, qw, cv, ov, cd, od, t_step, density_mode, copy_to_tmp, reflect, false, zb, zb + ends)));
            return launch_advect<T, OFF, SLAB>(s, p, qu, qv, qw, cv, ov, cd, od, t_step, density_mode, copy_to_tmp, reflect, false, ze - ends, ze);
        }
        dim3 b(64, 4, 1), g(div_up(s->nx, 64), div_up(s->ny, 4), ze - zb);
        hipLaunchKernelGGL((k_advect<T, OFF, SLAB>), g, b, 0, s->stream, p.g, qu, qv, qw, F<T>(p, cd, 3), U, V, W, F<T>(p, ov, 0),
                           F<T>(p, ov, 1), F<T>(p, ov, 2), F<T>(p, od, 3), c0, c1, c2, p.flags, t_step, density_mode, zb,
                           s->doob);
        return FS_OK;
    }
    auto go = [&](auto kern, size_t lds) -> int {
        SIM_HIP(allow_lds(s, (const void*)kern, lds));
        constexpr int TY = AdvTile<T, false>::TY;
        const int tiles = div_up(s->nx, 64) * div_up(s->ny, TY);
        // workgroups a CU holds at once: limited by LDS (160 KiB) and, at 72 registers, to 3 x 8 waves
        const int per_cu = std::max(1, std::min(3, (int)((size_t)160 * 1024 / lds)));
        int zchunk, zlen, zgroups;
        if (ends > 0) {   // two workgroups along z: the first and the last `ends` planes of [zb, ze)
            zchunk = ze - ends - zb; zlen = ends; zgroups = 2;
        } else {
            // Chunks of at most 128 planes: the workgroups of a round start together and the front they form stays
            // compact, so that the apron rows a tile shares with its neighbours are still in L2 when the neighbour
            // asks; tiles that march the whole of a 512-plane grid drift apart (512^3: reflected advection 1.45 ms
            // with one chunk, 1.34 with chunks of 64 to 128 planes, 1.36 with 32, 1.45 with 16).
            const int nchunks = pick_chunks(ze - zb, tiles, per_cu * s->num_cus, AdvTile<T, false>::kMaxChunk);
            zchunk = zlen = div_up(ze - zb, nchunks);
            zgroups = div_up(ze - zb, zchunk);
        }
        if (zlen > AdvTile<T, false>::kMaxChunk) return set_err(s, FS_ESTATE, "an advection chunk of more planes than the kernel keeps notes for");
        dim3 b(64, TY, 1), g(div_up(s->nx, 64), div_up(s->ny, TY), zgroups);
        double* res_part = nullptr;
        if (with_res) {   // the launches of one advection write their partials one after the other
            const size_t blocks = (size_t)g.x * g.y * g.z;
            if (!p.adv_part) {
                // at most planes/2 chunks per launch, three launches per advection
                const size_t cap = (size_t)tiles * (size_t)(p.own_planes() / 2 + 6);
                SIM_HIP(hipMalloc((void**)&p.adv_part, cap * 2 * sizeof(double)));
                p.adv_cap = cap;
            }
            if (p.adv_blocks + blocks > p.adv_cap) return set_err(s, FS_ESTATE, "residual partials of the advection do not fit");
            res_part = p.adv_part + 2 * (size_t)p.adv_blocks;
            p.adv_blocks += (int)blocks;
        }
        hipLaunchKernelGGL(kern, g, b, lds, s->stream, p.g, qu, qv, qw, F<T>(p, cd, 3), U, V, W, F<T>(p, ov, 0), F<T>(p, ov, 1),
                           F<T>(p, ov, 2), F<T>(p, od, 3), c0, c1, c2, p.flags, t_step, density_mode, zb, ze, zchunk, zlen, s->doob,
                           res_part);
        return FS_OK;
    };
    if (reflect) return go(k_advect_tiled<T, OFF, true, SLAB, true>, AdvTile<T, true>::lds_bytes());
    if (qu == U) return go(k_advect_tiled<T, OFF, false, SLAB>, AdvTile<T, false>::lds_bytes());
    return go(k_advect_tiled<T, OFF, true, SLAB>, AdvTile<T, true>::lds_bytes());
}

// copy_to_tmp: the new velocity is also written to the scratch fields (the unprojected half-step
// field the reflection needs later).
// res_sum: also returns the sum of squared divergences of the carrying velocity (what residual() adds up),
// evaluated by the advection kernel itself; the call then waits for that number.
template <typename T>
int advect(fs3d_sim* s, double t_step, bool q_from_tmp, int density_mode, bool copy_to_tmp = false, bool reflect = false,
           double* res_sum = nullptr) {
    FS_TRY(prepare(s, HALO_ALL | (q_from_tmp ? HALO_TMP : 0)));
    const bool slab = s->nparts_global > 1;
    const int cv = s->curv, ov = s->curv ^ 1, cd = s->curd, od = s->curd ^ 1;
    // The owned planes, shell included, so that the new generation is complete.  With several slabs the two
    // planes next to every interface come first and their exchange starts at once, beside the advection of
    // the slab interiors (pass 0: those planes, pass 1: the rest).
    for (Part& p : s->parts) p.adv_blocks = 0;
    for (int pass = slab ? 0 : 1; pass < 2; ++pass) {
        for (Part& p : s->parts) {
            const T* qu = q_from_tmp ? (const T*)p.tmp[0] : F<T>(p, cv, 0);
            const T* qv = q_from_tmp ? (const T*)p.tmp[1] : F<T>(p, cv, 1);
            const T* qw = q_from_tmp ? (const T*)p.tmp[2] : F<T>(p, cv, 2);
            // byte offsets into a field fit 32 bits (FSAMD_WIDE_OFFSETS=1 selects the 64-bit kernels regardless:
            // the tests use it to run them on small grids)
            static const bool force_wide = getenv("FSAMD_WIDE_OFFSETS") && getenv("FSAMD_WIDE_OFFSETS")[0] == '1';
            const bool small = !force_wide && (size_t)p.n * sizeof(T) < ((size_t)1 << 32);
            const int a = p.H, b = p.H + p.own_planes();
            const int lo_end = slab && p.has_lo ? std::min(a + p.H, b) : a;
            const int hi_beg = slab && p.has_hi ? std::max(b - p.H, lo_end) : b;
            // pass 0: the planes next to the interfaces, both ends in one launch where the slab has two of the
            // same depth; pass 1: the rest
            const bool both = lo_end - a == p.H && b - hi_beg == p.H && hi_beg > lo_end;
            const int ranges[3][3] = {{a, both ? b : lo_end, both ? p.H : 0}, {hi_beg, both ? hi_beg : b, 0}, {lo_end, hi_beg, 0}};
            for (int r = pass == 0 ? 0 : 2; r < (pass == 0 ? 2 : 3); ++r) {
                const int zb = ranges[r][0], ze = ranges[r][1], ends = ranges[r][2];
                if (small && slab) FS_TRY((launch_advect<T, uint32_t, true>(s, p, qu, qv, qw, cv, ov, cd, od, (T)t_step, density_mode, copy_to_tmp, reflect, res_sum != nullptr, zb, ze, ends)));
                else if (small) FS_TRY((launch_advect<T, uint32_t, false>(s, p, qu, qv, qw, cv, ov, cd, od, (T)t_step, density_mode, copy_to_tmp, reflect, res_sum != nullptr, zb, ze, ends)));
                else if (slab) FS_TRY((launch_advect<T, int64_t, true>(s, p, qu, qv, qw, cv, ov, cd, od, (T)t_step, density_mode, copy_to_tmp, reflect, res_sum != nullptr, zb, ze, ends)));
                else FS_TRY((launch_advect<T, int64_t, false>(s, p, qu, qv, qw, cv, ov, cd, od, (T)t_step, density_mode, copy_to_tmp, reflect, res_sum != nullptr, zb, ze, ends)));
            }
        }
        SIM_HIP(hipGetLastError());
        if (pass == 0) {
            // the planes the neighbours need exist: send them (exchange_async queues behind what the main stream
            // holds now; the base pointers of the exchange are those of the new generation)
            s->curv = ov;
            s->curd = od;
            s->halo_dirty |= HALO_ALL | (copy_to_tmp ? HALO_TMP : 0);
            FS_TRY(refresh_halos(s, HALO_ALL | HALO_TMP, true));
        }
    }
    if (!slab) {
        s->curv = ov;
        s->curd = od;
        s->halo_dirty = 0;
    }
    if (res_sum && single_part(s)) {
        FS_TRY(queue_reduction_single(s, s->parts[0].adv_part, s->parts[0].adv_blocks));
        FS_TRY(wait_host_record(s));
        *res_sum = s->hres[0];
        return FS_OK;
    }
    if (res_sum) {
        for (int k = 0; k < s->nlocal; ++k)
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s->stream, s->parts[k].adv_part, s->parts[k].adv_blocks,
                               s->dres + 2 * k);
        SIM_HIP(hipGetLastError());
        FS_TRY(queue_reduction(s, false));
        FS_TRY(wait_host_record(s));
        *res_sum = s->hres[0];
        return FS_OK;
    }
    return FS_OK;   // a backtrace beyond the halo raises the flag behind s->dres: see oob_status
}

// One sweep and the advection that follows it in ONE launch (fs_sweep_advect.hpp): the current velocity is swept in
// LDS and advected along itself into the other generation, with the residual of the swept field added up on the way.
// *bad: some cell could not be served (a backtrace of a cell or more, a non-finite sample): the caller drops the
// result.  4-byte elements, single domain.
bool fused_step_possible(const fs3d_sim* s) {
    return single_part(s) && s->cfg.dtype == FS_F32 && s->cfg.ordering == FS_ORDER_RED_BLACK &&
           s->cfg.sweep_kernel == FS_SWEEP_FUSED && s->cfg.advect_kernel == FS_ADVECT_TILED && s->cfg.solver == FS_SOLVER_SWEEPS;
}

// Whether fs3d_advance uses the launch by itself.  It does not unless FSAMD_FUSED_STEP=1: the launch moves 33 instead
// of 58 bytes per cell but is bound by instruction issue, not by memory (one workgroup of 16 waves per CU, the visits
// and the advection of a step on the same lanes), and at 512^3 takes 1.9 ms where the two launches take 1.65 ms.
bool fused_step_wanted(const fs3d_sim* s) {
    const char* yes = getenv("FSAMD_FUSED_STEP");   // read per call: tests switch it between simulations
    return yes && yes[0] == '1' && fused_step_possible(s);
}

int sweep_advect(fs3d_sim* s, double t_step, int density_mode, double* res_sum, bool* bad) {
    using C = SweepAdvTile;
    FS_TRY(prepare(s, HALO_ALL));
    Part& p = s->parts[0];
    const int cv = s->curv, ov = s->curv ^ 1, cd = s->curd, od = s->curd ^ 1;
    if (s->tilde_gen == ov) tilde_to_tmp(s);   // the other generation is about to be overwritten
    auto kern = k_sweep_advect<int64_t>;
    SIM_HIP(allow_lds(s, (const void*)kern, C::lds_bytes()));
    const int tiles = div_up(s->nx, C::TX) * div_up(s->ny, C::TY), planes = p.g.nzl;
    // z-chunks: (rounds of workgroups over the CUs) x (planes a workgroup marches: its chunk and seven more)
    int best_c = 1;
    double best = 1e300;
    for (int c = 1; c <= std::max(1, planes / 2); ++c) {
        const int zc = div_up(planes, c);
        const double cost = (double)div_up(tiles * div_up(planes, zc), s->num_cus) * (zc + 7);
        if (cost < best) { best = cost; best_c = c; }
    }
    const int zchunk = div_up(planes, best_c), zgroups = div_up(planes, zchunk);
    dim3 b(C::TX, C::TY, 1), g(div_up(s->nx, C::TX), div_up(s->ny, C::TY), zgroups);
    const size_t blocks = (size_t)g.x * g.y * g.z;
    if (!p.adv_part || p.adv_cap < blocks) {
        if (p.adv_part) { FS_TRY(quiesce(s)); (void)hipFree(p.adv_part); p.adv_part = nullptr; }
        const size_t cap = std::max(blocks, (size_t)div_up(s->nx, 64) * div_up(s->ny, 8) * (size_t)(p.own_planes() / 2 + 6));
        SIM_HIP(hipMalloc((void**)&p.adv_part, cap * 2 * sizeof(double)));
        p.adv_cap = cap;
    }
    p.adv_blocks = (int)blocks;
    hipLaunchKernelGGL(kern, g, b, C::lds_bytes(), s->stream, p.g, F<float>(p, cv, 0), F<float>(p, cv, 1), F<float>(p, cv, 2),
                       F<float>(p, cd, 3), F<float>(p, ov, 0), F<float>(p, ov, 1), F<float>(p, ov, 2), F<float>(p, od, 3), p.flags,
                       (float)s->cfg.omega, (float)t_step, density_mode, 0, planes, zchunk, s->doob, p.adv_part);
    SIM_HIP(hipGetLastError());
    s->curv = ov;
    s->curd = od;
    s->halo_dirty = 0;
    FS_TRY(queue_reduction_single(s, p.adv_part, p.adv_blocks));
    FS_TRY(wait_host_record(s));
    *res_sum = s->hres[0];
    *bad = s->hres[1] > 0.0;
    if (*bad) {   // (the flag is the out-of-halo flag of the slab path, which a single domain never raises)
        SIM_HIP(hipMemsetAsync(s->doob, 0, sizeof(int), s->stream));
        s->hres[1] = 0.0;
    }
    return FS_OK;
}

template <typename T>
int buoyancy(fs3d_sim* s, double dt) {
    FS_TRY(prepare(s));
    for (Part& p : s->parts) {
        if (p.ze - p.zb <= 0) continue;
        dim3 b(64, 4, 1), g(div_up(s->nx - 2, 64), div_up(s->ny - 2, 4), p.ze - p.zb);
        hipLaunchKernelGGL(k_buoyancy<T>, g, b, 0, s->stream, p.g, F<T>(p, s->curv, 1), F<T>(p, s->curd, 3), p.flags, (T)dt,
                           (T)(0.1 * s->cfg.buoyancy), p.zb, p.ze);
    }
    SIM_HIP(hipGetLastError());
    s->halo_dirty |= HALO_VEL;
    return FS_OK;
}

// The MacCormack pass after an advection (fs_kernels3d.hpp, k_maccormack): q = what the advection
// carried (the previous generation, or the scratch velocity with the previous density), the carrying
// velocity = the previous generation's, f = the current generation.  The corrected fields become the
// current generation; copy_to_tmp also leaves the corrected velocity in the scratch fields.
template <typename T>
int maccormack(fs3d_sim* s, double t_step, bool q_from_tmp, bool copy_to_tmp) {
    for (Part& p : s->parts)
        for (int k = 0; k < 4; ++k)
            if (!p.mc[k]) FS_TRY(alloc_field(s, p, &p.mc[k], s->esz));
    // the backward trace reads the halo planes of the forward result
    FS_TRY(prepare(s, HALO_ALL | (q_from_tmp ? HALO_TMP : 0)));
    const int cv = s->curv, pv = s->curv ^ 1, cd = s->curd, pd = s->curd ^ 1;
    for (Part& p : s->parts) {
        const int zb = p.H, ze = p.H + p.own_planes();
        dim3 b(64, 4, 1), g(div_up(s->nx, 64), div_up(s->ny, 4), ze - zb);
        const T* qu = q_from_tmp ? (const T*)p.tmp[0] : F<T>(p, pv, 0);
        const T* qv = q_from_tmp ? (const T*)p.tmp[1] : F<T>(p, pv, 1);
        const T* qw = q_from_tmp ? (const T*)p.tmp[2] : F<T>(p, pv, 2);
        T* c0 = copy_to_tmp ? (T*)p.tmp[0] : nullptr;
        T* c1 = copy_to_tmp ? (T*)p.tmp[1] : nullptr;
        T* c2 = copy_to_tmp ? (T*)p.tmp[2] : nullptr;
        hipLaunchKernelGGL(k_maccormack<T>, g, b, 0, s->stream, p.g, qu, qv, qw, F<T>(p, pd, 3), F<T>(p, pv, 0), F<T>(p, pv, 1),
                           F<T>(p, pv, 2), F<T>(p, cv, 0), F<T>(p, cv, 1), F<T>(p, cv, 2), F<T>(p, cd, 3), (T*)p.mc[0],
                           (T*)p.mc[1], (T*)p.mc[2], (T*)p.mc[3], c0, c1, c2, p.flags, (T)t_step, zb, s->doob);
        for (int k = 0; k < 3; ++k) std::swap(p.f[cv][k], p.mc[k]);
        std::swap(p.f[cd][3], p.mc[3]);
    }
    SIM_HIP(hipGetLastError());
    s->halo_dirty |= HALO_ALL | (copy_to_tmp ? HALO_TMP : 0);
    return refresh_halos(s, HALO_ALL | HALO_TMP, true);
}

// Deferred check (check_every = FS_CHECK_DEFERRED).  The advection that follows a projection reads the projected
// field anyway and can add up its squared divergences on the way (fs_advect_tiled.hpp), so the last evaluation
// of a projection need not be a pass of its own:
//  - where the same projection of the previous step needed one sweep, that sweep runs unverified;
//  - else the residual is evaluated one sweep before the expected count (the count may have dropped) and, if
//    that fails, the expected last sweep runs unverified.
// If the advection then reports a residual <= tol (nearly always: consecutive steps need the same count) the
// step has saved residual passes; if not, the advected generation is dropped (the projected field is still
// intact in the other one), the projection goes on with evaluations of its own and the advection is repeated.
// Every projection still ends on a verified residual <= tol.  A miss is paid for by 2, 4, 8 projections
// without deferral.
// Returns -1: no deferral; 0: deferral of the last evaluation only; 1: one sweep, unverified.
int deferral(fs3d_sim* s, int slot, bool mac) {
    const fs3d_config& c = s->cfg;
    if (c.check_every != FS_CHECK_DEFERRED || c.stop_mode != FS_STOP_L2_RESIDUAL ||
        c.ordering != FS_ORDER_RED_BLACK || c.advect_kernel != FS_ADVECT_TILED || mac)
        return -1;
    if (s->spec_backoff[slot] > 0) { --s->spec_backoff[slot]; return -1; }
    const int h = s->sweep_hint[slot];
    // a miss costs an advection, several evaluations' worth: only where the last two steps agreed on the count
    // and where an evaluation is a good part of the projection's cost (at twenty sweeps the count wanders by one
    // from step to step, and one evaluation in three is not worth the risk)
    if (h >= c.max_iters || h != s->sweep_hint_prev[slot] || h > 4) return -1;
    if (c.solver == FS_SOLVER_MULTIGRID) return h == 1 ? 1 : -1;   // a cycle is evaluated by the solver itself
    return h > 1 ? 0 : 1;
}

// A projection followed by the advection `adv(res_sum)` that reads the projected field.
// `fused(res_sum, bad)`, where the caller has one: the single unverified sweep and the advection in one launch.
template <typename T, typename Adv, typename Fused>
int project_then_advect(fs3d_sim* s, int slot, bool mac, hipEvent_t after_projection, Adv adv, Fused fused, bool can_fuse) {
    const int mode = deferral(s, slot, mac);
    if (mode == 1 && can_fuse && fused_step_wanted(s) && s->cfg.max_iters > 1) {
        if (s->fuse_backoff[slot] > 0) --s->fuse_backoff[slot];
        else {
            if (after_projection) SIM_HIP(hipEventRecord(after_projection, s->stream));
            double sum = 0;
            bool bad = false;
            FS_TRY(fused(&sum, &bad));
            const double ninterior = (double)(s->nx - 2) * (s->ny - 2) * (s->nz - 2);
            const double metric = sqrt(sum) / ninterior / s->resid_dt;
            if (!bad && !(metric > s->cfg.tol)) {
                s->stats.sweeps = 1;
                s->stats.sweeps_step += 1;
                s->stats.metric = metric;
                s->sweep_hint_prev[slot] = s->sweep_hint[slot];
                s->sweep_hint[slot] = 1;
                s->spec_fails[slot] = 0;
                s->fuse_fails[slot] = 0;
                ++s->stats.fused;
                return FS_OK;
            }
            // not served, or not converged: back to the field before the sweep, and on with separate launches (which
            // deal with a missed tolerance themselves)
            s->curv ^= 1;
            s->curd ^= 1;
            if (slot == 0) s->tilde_gen = -1;
            s->fuse_fails[slot] = std::min(s->fuse_fails[slot] + 1, 5);
            s->fuse_backoff[slot] = 2 << s->fuse_fails[slot];
            ++s->stats.redone;
        }
    }
    bool deferred = mode > 0;
    FS_TRY(project<T>(s, nullptr, nullptr, slot, std::max(mode, 0), 0, mode == 0 ? &deferred : nullptr));
    if (after_projection) SIM_HIP(hipEventRecord(after_projection, s->stream));
    if (!deferred) return adv(nullptr);
    const int n = s->stats.sweeps;
    double sum = 0;
    FS_TRY(adv(&sum));
    const double ninterior = (double)(s->nx - 2) * (s->ny - 2) * (s->nz - 2);
    const double metric = sqrt(sum) / ninterior / s->resid_dt;   // as residual() and project() form it
    if (!(metric > s->cfg.tol)) {
        s->stats.metric = metric;
        if (mode > 0) {
            s->sweep_hint_prev[slot] = s->sweep_hint[slot];
            s->sweep_hint[slot] = n;
        }
        s->spec_fails[slot] = 0;
        return FS_OK;
    }
    // not converged after all: back to the projected field, which the advection only read
    FS_TRY(join_halos(s));
    s->curv ^= 1;
    s->curd ^= 1;
    s->halo_dirty |= HALO_ALL;
    if (slot == 0) s->tilde_gen = -1;   // the dropped generation held ut
    s->spec_fails[slot] = std::min(s->spec_fails[slot] + 1, 3);
    s->spec_backoff[slot] = 1 << s->spec_fails[slot];
    ++s->stats.redone;
    FS_TRY(project<T>(s, nullptr, nullptr, slot, 0, n));
    return adv(nullptr);
}

template <typename T>
int step(fs3d_sim* s, double dt) {
    s->stats.sweeps_step = 0;
    s->stats.evaluations_step = 0;
    s->tilde_gen = -1;
    s->resid_dt = dt > 0 ? dt : 1.0;
    SIM_HIP(hipEventRecord(s->ev[0], s->stream));
    if (s->cfg.buoyancy != 0.0) FS_TRY(buoyancy<T>(s, dt));
    const bool mac = s->cfg.advection == FS_ADVECTION_MACCORMACK;
    if (s->cfg.step_mode == FS_STEP_REFERENCE) {
        // FluidSim::update (main.cpp:102-130)
        FS_TRY((project_then_advect<T>(s, 0, mac, s->ev[1], [&](double* res) -> int {
            FS_TRY(advect<T>(s, dt, false, mac ? 0 : 1, false, false, res));
            if (mac) FS_TRY(maccormack<T>(s, dt, false, false));
            return (int)FS_OK;
        }, [&](double* res, bool* bad) -> int { return sweep_advect(s, dt, 1, res, bad); }, !mac)));
        SIM_HIP(hipEventRecord(s->ev[2], s->stream));
        SIM_HIP(hipEventRecord(s->ev[3], s->stream));
    } else {
        // Advection-reflection, phase-shifted to the reference's convention of projecting first and
        // leaving the advected field for the next step to project:
        //   u0 = P(state); ut = A(u0; u0, dt/2); uh = P(ut); ur = 2 uh - ut; state = A(ur; uh, dt/2)
        // ut is needed again after the second projection.  Where that projection sweeps out of place and the
        // staged kernel does the second advection, ut stays in the generation the first advection writes and
        // reaches the scratch fields by a trade of arrays (tilde_to_tmp); else the advection writes it twice.
        const bool keep = !mac && s->cfg.advect_kernel == FS_ADVECT_TILED && s->cfg.ordering == FS_ORDER_RED_BLACK &&
                          s->cfg.sweep_kernel != FS_SWEEP_TWO_PASS;
        FS_TRY((project_then_advect<T>(s, 0, mac, nullptr, [&](double* res) -> int {
            FS_TRY(advect<T>(s, 0.5 * dt, false, 0, !mac && !keep, false, res));
            if (keep) s->tilde_gen = s->curv;
            if (mac) FS_TRY(maccormack<T>(s, 0.5 * dt, false, true));
            return (int)FS_OK;
        }, [&](double* res, bool* bad) -> int {
            FS_TRY(sweep_advect(s, 0.5 * dt, 0, res, bad));
            s->tilde_gen = s->curv;   // (keep holds wherever this launch applies)
            return (int)FS_OK;
        }, keep)));
        FS_TRY((project_then_advect<T>(s, 1, mac, s->ev[1], [&](double* res) -> int {
            if (s->cfg.advect_kernel == FS_ADVECT_TILED && !mac) {
                // the staged kernel forms ur while it loads ut and uh: no reflection pass
                tilde_to_tmp(s);   // (after a single sweep ut is still in the generation this advection writes)
                return advect<T>(s, 0.5 * dt, true, 0, false, true, res);
            }
            FS_TRY(prepare(s, HALO_VEL | HALO_TMP));   // the reflection runs over halo planes too
            for (Part& p : s->parts)
                hipLaunchKernelGGL(k_reflect<T>, dim3(2048), dim3(256), 0, s->stream, (int64_t)p.n, F<T>(p, s->curv, 0),
                                   F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), (const T*)p.tmp[0], (const T*)p.tmp[1],
                                   (const T*)p.tmp[2], (T*)p.tmp[0], (T*)p.tmp[1], (T*)p.tmp[2]);
            SIM_HIP(hipGetLastError());
            FS_TRY(advect<T>(s, 0.5 * dt, true, 0));
            if (mac) FS_TRY(maccormack<T>(s, 0.5 * dt, true, false));
            return (int)FS_OK;
        }, [&](double*, bool*) -> int { return (int)FS_OK; }, false)));
        SIM_HIP(hipEventRecord(s->ev[2], s->stream));
        SIM_HIP(hipEventRecord(s->ev[3], s->stream));
    }
    s->stats.time += dt;
    s->stats.step += 1;
    s->stats.sweeps_total += s->stats.sweeps_step;
    s->stats.evaluations_total += s->stats.evaluations_step;
    return FS_OK;
}

void owned_range(const fs3d_sim* s, int* z0, int* z1);

// ---- diagnostics -------------------------------------------------------------
template <typename T>
int vorticity(fs3d_sim* s, bool store, double* enstrophy) {
    FS_TRY(prepare(s));
    for (int k = 0; k < s->nlocal; ++k) {
        Part& p = s->parts[k];
        const int zb = p.H, ze = p.H + p.own_planes();
        hipLaunchKernelGGL(k_vorticity<T>, dim3(kReduceBlocks), dim3(256), 0, s->stream, p.g, F<T>(p, s->curv, 0),
                           F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), zb, ze, store ? (T*)p.tmp[0] : (T*)nullptr, p.partials);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s->stream, p.partials, kReduceBlocks, s->dres + 2 * k);
    }
    SIM_HIP(hipGetLastError());
    return finish_reduction(s, enstrophy, nullptr);
}

template <typename T>
int energy_mass(fs3d_sim* s, int what, double* sum, double* mx) {
    FS_TRY(prepare(s, HALO_ALL));
    for (int k = 0; k < s->nlocal; ++k) {
        Part& p = s->parts[k];
        if (p.ze > p.zb)
            hipLaunchKernelGGL(k_energy_mass<T>, dim3(kReduceBlocks), dim3(256), 0, s->stream, p.g, F<T>(p, s->curv, 0),
                               F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), F<T>(p, s->curd, 3), p.flags, p.zb, p.ze, what,
                               p.partials);
        else
            SIM_HIP(hipMemsetAsync(p.partials, 0, sizeof(double) * 2 * kReduceBlocks, s->stream));
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s->stream, p.partials, kReduceBlocks, s->dres + 2 * k);
    }
    SIM_HIP(hipGetLastError());
    return finish_reduction(s, sum, mx);
}

template <typename T>
int downsample(fs3d_sim* s, int f, void* host, size_t count) {
    int z0, z1;
    owned_range(s, &z0, &z1);
    const int mx = div_up(s->nx, f), my = div_up(s->ny, f);
    for (Part& p : s->parts)
        if (p.z0 % f != 0) return set_err(s, FS_EINVAL, "every slab must start on a multiple of the factor");
    size_t total = 0;
    for (Part& p : s->parts) total += (size_t)div_up(p.own_planes(), f) * my * mx;
    if (count != total) return set_err(s, FS_EINVAL, "count must be the number of blocks of the owned planes");
    T* dev = nullptr;
    SIM_HIP(hipMalloc((void**)&dev, total * sizeof(T)));
    size_t off = 0;
    for (Part& p : s->parts) {
        const int mz = div_up(p.own_planes(), f);
        dim3 b(64, 4, 1), g(div_up(mx, 64), div_up(my, 4), mz);
        hipLaunchKernelGGL(k_downsample<T>, g, b, 0, s->stream, p.g, F<T>(p, s->curd, 3), f, p.H, p.H + p.own_planes(), mx, my,
                           dev + off);
        off += (size_t)mz * my * mx;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, total * sizeof(T), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);   // (halo traffic in flight touches halo planes only)
    (void)hipFree(dev);
    SIM_HIP(e);
    return FS_OK;
}

#define DISPATCH(s, fn, ...) ((s)->cfg.dtype == FS_F64 ? fn<double>(__VA_ARGS__) : fn<float>(__VA_ARGS__))

// ---- host <-> device field access -----------------------------------------
// Plane range addressed by set/get: the whole grid on a single rank, this
// rank's owned planes otherwise.
void owned_range(const fs3d_sim* s, int* z0, int* z1) {
    *z0 = s->parts.front().z0;
    *z1 = s->parts.back().z1;
}

void* field_ptr(fs3d_sim* s, Part& p, int which) {
    if (which >= FS_FIELD_U && which <= FS_FIELD_W) return p.f[s->curv][which];
    if (which == FS_FIELD_DENSITY) return p.f[s->curd][which];
    return p.tmp[0];
}

// ---- scene helpers ----------------------------------------------------------
struct Stamp { std::vector<int64_t> idx; std::vector<float> val; };

// Cells of the sphere (main.cpp:46-52) that fall into a part's local array, with their weights.
void sphere_cells(const fs3d_sim* s, const Part& p, float radius, float cx, float cy, float cz, Stamp* out) {
    const float r2 = powf(radius, 2.0f);
    auto lo = [](float c, float r) { return (int)floorf(c - r) - 1; };
    auto hi = [](float c, float r) { return (int)ceilf(c + r) + 1; };
    const int x0 = std::max(1, lo(cx, radius)), x1 = std::min(s->nx - 2, hi(cx, radius));
    const int y0 = std::max(1, lo(cy, radius)), y1 = std::min(s->ny - 2, hi(cy, radius));
    const int z0 = std::max(1, lo(cz, radius)), z1 = std::min(s->nz - 2, hi(cz, radius));
    for (int z = z0; z <= z1; ++z) {
        const int lz = z - p.g.gz0;
        if (lz < 0 || lz >= p.g.nzl) continue;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const float xc = x - cx, yc = y - cy, zc = z - cz;
                const float dist = powf(xc, 2.0f) + powf(yc, 2.0f) + powf(zc, 2.0f);
                if (dist < r2) {
                    out->idx.push_back((int64_t)lz * p.g.sz + (int64_t)y * s->nx + x);
                    out->val.push_back(1.0f - (dist / r2));
                }
            }
    }
}

template <typename T>
int scatter_to(fs3d_sim* s, Part& p, const Stamp& st, int field, float scale, int cell_type) {
    const int n = (int)st.idx.size();
    if (!n) return FS_OK;
    int64_t* didx = nullptr; float* dval = nullptr;
    hipError_t e = hipMalloc((void**)&didx, n * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&dval, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(didx, st.idx.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dval, st.val.data(), n * sizeof(float), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) {
        if (field >= 0)
            hipLaunchKernelGGL(k_scatter<T>, dim3(div_up(n, 256)), dim3(256), 0, s->stream, n, didx, dval, scale, F<T>(p, field < 3 ? s->curv : s->curd, field));
        if (cell_type >= 0)
            hipLaunchKernelGGL(k_scatter_cells, dim3(div_up(n, 256)), dim3(256), 0, s->stream, n, didx, (uint8_t)cell_type, p.cells);
        e = hipGetLastError();
    }
    const int rc = e == hipSuccess ? quiesce(s) : FS_OK;   // the host vectors and the device copies must outlive the kernels
    (void)hipFree(didx); (void)hipFree(dval);
    SIM_HIP(e);
    return rc;
}

template <typename T>
int sphere_emitter(fs3d_sim* s, float radius, float cx, float cy, float cz, float strength) {
    for (Part& p : s->parts) {
        Stamp st;
        sphere_cells(s, p, radius, cx, cy, cz, &st);
        // main.cpp:53-55: StaticFluid, density = w, v = 20 w
        FS_TRY(scatter_to<T>(s, p, st, 3, 1.0f, FS_CELL_STATIC_FLUID));
        FS_TRY(scatter_to<T>(s, p, st, 1, strength, -1));
    }
    s->cells_dirty = true;
    return FS_OK;
}

template <typename T>
int fill_sphere(fs3d_sim* s, float radius, float cx, float cy, float cz) {
    for (Part& p : s->parts) {
        Stamp st;
        sphere_cells(s, p, radius, cx, cy, cz, &st);
        FS_TRY(scatter_to<T>(s, p, st, 3, 1.0f, -1));   // main.cpp:69-71
    }
    return FS_OK;
}

template <typename T>
__global__ void k_bottom_influence(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w, uint8_t* __restrict__ cells) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z;
    const int gz = z + g.gz0;
    if (x < 1 || y < 1 || gz < 1 || x > g.nx - 2 || y > g.ny - 2 || gz > g.gnz - 2) return;
    const int64_t i = gidx(g, x, y, z);
    // main.cpp:78-85
    u[i] = (T)((1.0f - ((float)x / (float)g.nx)) * 1.0f);
    v[i] = (T)((1.0f - ((float)y / (float)g.ny)) * 1.0f);
    w[i] = (T)((1.0f - ((float)gz / (float)g.gnz)) * 1.0f);
    if (x == 1 || y == 1 || gz == 1) cells[i] = FS_CELL_STATIC_FLUID;
}

template <typename T>
__global__ void k_clear_interior(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w, T* __restrict__ d) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, z = blockIdx.z;
    const int gz = z + g.gz0;
    if (x < 1 || y < 1 || gz < 1 || x > g.nx - 2 || y > g.ny - 2 || gz > g.gnz - 2) return;
    const int64_t i = gidx(g, x, y, z);
    u[i] = 0; v[i] = 0; w[i] = 0; d[i] = 0;   // main.cpp:91-94
}

template <typename T>
__global__ void k_source_box(Grid g, T* __restrict__ d, int x0, int x1, int y0, int y1, int z0, int z1) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x + x0, y = blockIdx.y * blockDim.y + threadIdx.y + y0;
    const int gz = blockIdx.z + z0, lz = gz - g.gz0;
    if (x >= x1 || y >= y1 || gz >= z1 || lz < 0 || lz >= g.nzl) return;
    d[gidx(g, x, y, lz)] = (T)1;   // include/Grid2.h:58
}

template <typename T>
int scene_kernel(fs3d_sim* s, int what, const int* box) {
    for (Part& p : s->parts) {
        dim3 b(64, 4, 1), g(div_up(s->nx, 64), div_up(s->ny, 4), p.g.nzl);
        if (what == 0)
            hipLaunchKernelGGL(k_bottom_influence<T>, g, b, 0, s->stream, p.g, F<T>(p, s->curv, 0), F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), p.cells);
        else if (what == 1)
            hipLaunchKernelGGL(k_clear_interior<T>, g, b, 0, s->stream, p.g, F<T>(p, s->curv, 0), F<T>(p, s->curv, 1), F<T>(p, s->curv, 2), F<T>(p, s->curd, 3));
        else {
            const int ex = box[1] - box[0], ey = box[3] - box[2], ez = box[5] - box[4];
            if (ex <= 0 || ey <= 0 || ez <= 0) continue;
            dim3 gb(div_up(ex, 64), div_up(ey, 4), ez);
            hipLaunchKernelGGL(k_source_box<T>, gb, b, 0, s->stream, p.g, F<T>(p, s->curd, 3), box[0], box[1], box[2], box[3], box[4], box[5]);
        }
    }
    SIM_HIP(hipGetLastError());
    if (what == 0) s->cells_dirty = true;
    return FS_OK;
}

bool config_ok(const fs3d_config* c, std::string* why) {
    if (c->dtype != FS_F32 && c->dtype != FS_F64) { *why = "dtype must be FS_F32 or FS_F64"; return false; }
    if (c->ordering != FS_ORDER_RED_BLACK && c->ordering != FS_ORDER_LEXICOGRAPHIC) { *why = "bad ordering"; return false; }
    if (c->stop_mode < FS_STOP_FIXED || c->stop_mode > FS_STOP_L2_RESIDUAL) { *why = "bad stop_mode"; return false; }
    if (c->step_mode != FS_STEP_REFERENCE && c->step_mode != FS_STEP_REFLECTION) { *why = "bad step_mode"; return false; }
    if (c->max_iters < 1) { *why = "max_iters must be >= 1"; return false; }
    if (c->check_every < FS_CHECK_DEFERRED) { *why = "check_every must be >= 1, FS_CHECK_ADAPTIVE or FS_CHECK_DEFERRED"; return false; }
    if (c->local_slabs < 1) { *why = "local_slabs must be >= 1"; return false; }
    if (c->sweep_kernel < FS_SWEEP_FUSED || c->sweep_kernel > FS_SWEEP_FUSED_LDS) { *why = "bad sweep_kernel"; return false; }
    if (c->advect_kernel < FS_ADVECT_TILED || c->advect_kernel > FS_ADVECT_DIRECT) { *why = "bad advect_kernel"; return false; }
    if (c->solver < FS_SOLVER_SWEEPS || c->solver > FS_SOLVER_MULTIGRID) { *why = "bad solver"; return false; }
    if (c->solver == FS_SOLVER_MULTIGRID && c->ordering != FS_ORDER_RED_BLACK) {
        *why = "the multigrid solver needs red-black ordering";
        return false;
    }
    if (c->solver == FS_SOLVER_MULTIGRID && c->stop_mode == FS_STOP_L1_CHANGE) {
        *why = "the multigrid solver stops on the residual or after a fixed count, not on the change of one sweep";
        return false;
    }
    if (c->advection < FS_ADVECTION_SEMI_LAGRANGE || c->advection > FS_ADVECTION_MACCORMACK) { *why = "bad advection"; return false; }
    if (!(c->omega > 0.0 && c->omega < 2.0)) { *why = "omega must lie in (0, 2)"; return false; }
    return true;
}

int create_common(int nx, int ny, int nz, const fs3d_config* cfg_in, int rank, int nranks, const uint8_t* id,
                  fs3d_sim** out) {
    if (!out) return FS_EINVAL;
    *out = nullptr;
    fs3d_config cfg;
    if (cfg_in) cfg = *cfg_in; else fs3d_default_config(&cfg);
    std::string why;
    if (nx < 3 || ny < 3 || nz < 3 || !config_ok(&cfg, &why) || rank < 0 || rank >= nranks) return FS_EINVAL;
    if (cfg.ordering == FS_ORDER_LEXICOGRAPHIC && (nranks > 1 || cfg.local_slabs > 1)) return FS_EINVAL;
    if (cfg.solver == FS_SOLVER_MULTIGRID && !mg_slabs_ok(nz, nranks * cfg.local_slabs)) return FS_EINVAL;
    fs3d_sim* s = new (std::nothrow) fs3d_sim();
    if (!s) return FS_ENOMEM;
    *out = s;   // handed back even on failure so that the caller can read the error text, then destroy
    s->cfg = cfg;
    s->nx = nx; s->ny = ny; s->nz = nz;
    s->esz = cfg.dtype == FS_F64 ? 8 : 4;
    s->rank = rank; s->nranks = nranks; s->nlocal = cfg.local_slabs; s->nparts_global = nranks * cfg.local_slabs;
    if (cfg.device >= 0) s->device = cfg.device;
    else SIM_HIP(hipGetDevice(&s->device));
    DeviceGuard guard(s->device);
    if (!guard.ok) return set_err(s, FS_EHIP, "hipSetDevice failed");
    SIM_HIP(hipDeviceGetAttribute(&s->num_cus, hipDeviceAttributeMultiprocessorCount, s->device));
    if (s->num_cus < 1) s->num_cus = 256;
    SIM_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (auto& e : s->ev) SIM_HIP(hipEventCreate(&e));
    SIM_HIP(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    SIM_HIP(hipEventCreateWithFlags(&s->ev_bnd, hipEventDisableTiming));
    SIM_HIP(hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming));
    SIM_HIP(hipEventCreateWithFlags(&s->ev_b, hipEventDisableTiming));
    SIM_HIP(hipMalloc((void**)&s->dres, sizeof(double) * (2 * s->nlocal + 4)));
    // (coherent: the host polls this record while a kernel writes it)
    SIM_HIP(hipHostMalloc((void**)&s->hres, sizeof(double) * 4, hipHostMallocCoherent | hipHostMallocMapped));
    // the flag the advection kernels raise sits behind the combined record
    s->doob = reinterpret_cast<int*>(s->dres + 2 * s->nlocal + 3);
    SIM_HIP(hipMemset(s->doob, 0, sizeof(double)));
    s->hres[0] = s->hres[1] = s->hres[2] = s->hres[3] = 0.0;
    if (nranks > 1) {
        s->rccl = Rccl::get(s->err);
        if (!s->rccl) return FS_ECOMM;
        ncclUniqueId uid;
        static_assert(sizeof(uid) == FS_COMM_ID_BYTES, "unique id size");
        memcpy(&uid, id, sizeof uid);
        FS_TRY(nccl_check(s, s->rccl->CommInitRank(&s->comm, nranks, uid, rank), "ncclCommInitRank"));
        s->comm_red = s->comm;
        // One communicator by default: the reductions then queue behind the halo traffic (join_halos), which is the
        // order every rank issues.  FSAMD_SPLIT_COMMUNICATOR=1 gives the reductions a copy of their own, so that
        // they run beside the halo traffic of the other stream; two communicators with work in flight on one
        // device is a pattern that needs a validated multi-GPU run first, hence opt-in.
        if (s->rccl->CommSplit && getenv("FSAMD_SPLIT_COMMUNICATOR") && getenv("FSAMD_SPLIT_COMMUNICATOR")[0] == '1') {
            ncclComm_t copy = nullptr;   // same ranks, same order; every rank takes part (collective call)
            if (s->rccl->CommSplit(s->comm, 0, rank, &copy, nullptr) == ncclSuccess && copy) s->comm_red = copy;
        }
    }
    FS_TRY(alloc_parts(s));
    FS_TRY(zero_fields(s));
    FS_TRY(init_cells(s));
    for (int k = 0; k < 3; ++k) SIM_HIP(hipEventRecord(s->ev[k], s->stream));
    SIM_HIP(hipEventRecord(s->ev[3], s->stream));
    FS_TRY(quiesce(s));
    return FS_OK;
}

// The text of the last failure of a call that takes no handle, per host thread (fs3d_last_error(NULL)).
thread_local std::string g_handleless_err;

}  // namespace

// ===========================================================================
extern "C" {

const char* fsamd_version(void) { return "fluidsim_amd 0.1 (gfx950)"; }

void fs3d_default_config(fs3d_config* c) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->device = -1;
    c->dtype = FS_F32;
    c->ordering = FS_ORDER_RED_BLACK;
    c->stop_mode = FS_STOP_L1_CHANGE;   // main.cpp:108
    c->max_iters = 1000;
    c->check_every = 1;
    c->step_mode = FS_STEP_REFERENCE;
    c->local_slabs = 1;
    c->sweep_kernel = FS_SWEEP_FUSED;
    c->advect_kernel = FS_ADVECT_TILED;
    c->advection = FS_ADVECTION_SEMI_LAGRANGE;
    c->solver = FS_SOLVER_SWEEPS;
    c->tol = 5.0;
    c->omega = 1.0;
    c->buoyancy = 0.0;
}

int fs3d_create(int nx, int ny, int nz, const fs3d_config* cfg, fs3d_sim** out) {
    return create_common(nx, ny, nz, cfg, 0, 1, nullptr, out);
}

int fsamd_comm_unique_id(uint8_t id[FS_COMM_ID_BYTES]) {
    std::string err;
    Rccl* R = Rccl::get(err);
    if (!R || !id) return FS_ECOMM;
    ncclUniqueId uid;
    if (R->GetUniqueId(&uid) != ncclSuccess) return FS_ECOMM;
    memcpy(id, &uid, sizeof uid);
    return FS_OK;
}

// Loopback check of the RCCL binding on one GPU: a one-rank communicator, an all-reduce and a grouped
// send/receive to itself on a private stream.  Returns FS_OK when the collective library could be
// loaded, every call succeeded and the data came back intact.
int fsamd_comm_selftest(int device) {
    std::string err;
    Rccl* R = Rccl::get(err);
    if (!R) return FS_ECOMM;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return FS_EHIP;
    DeviceGuard guard(dev);
    if (!guard.ok) return FS_EHIP;
    ncclUniqueId uid;
    if (R->GetUniqueId(&uid) != ncclSuccess) return FS_ECOMM;
    ncclComm_t comm = nullptr;
    if (R->CommInitRank(&comm, 1, uid, 0) != ncclSuccess) return FS_ECOMM;
    int rc = FS_OK;
    hipStream_t st = nullptr;
    double* d = nullptr;
    const int n = 1 << 16;
    std::vector<double> h(2 * n + 2);
    for (int i = 0; i < n; ++i) h[i] = 0.5 * i;
    h[2 * n] = 3.0; h[2 * n + 1] = 7.0;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&d, h.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = FS_EHIP;
    if (rc == FS_OK) {
        bool ok = R->AllReduce(d + 2 * n, d + 2 * n, 1, ncclDouble, ncclSum, comm, st) == ncclSuccess;
        ok = ok && R->AllReduce(d + 2 * n + 1, d + 2 * n + 1, 1, ncclDouble, ncclMax, comm, st) == ncclSuccess;
        ok = ok && R->GroupStart() == ncclSuccess;
        ok = ok && R->Send(d, n * sizeof(double), ncclUint8, 0, comm, st) == ncclSuccess;
        ok = ok && R->Recv(d + n, n * sizeof(double), ncclUint8, 0, comm, st) == ncclSuccess;
        ok = ok && R->GroupEnd() == ncclSuccess;
        // the copy of the communicator that the scalar reductions use, where the library offers the call: an
        // all-reduce on it, on another stream, while the first communicator is busy
        if (ok && R->CommSplit) {
            ncclComm_t copy = nullptr;
            hipStream_t st2 = nullptr;
            if (R->CommSplit(comm, 0, 0, &copy, nullptr) == ncclSuccess && copy) {
                ok = hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) == hipSuccess;
                ok = ok && R->AllReduce(d + 2 * n, d + 2 * n, 1, ncclDouble, ncclSum, copy, st2) == ncclSuccess;
                ok = ok && hipStreamSynchronize(st2) == hipSuccess;
                if (st2) (void)hipStreamDestroy(st2);
                (void)R->CommDestroy(copy);
            }
        }
        if (!ok) rc = FS_ECOMM;
    }
    if (rc == FS_OK && (hipStreamSynchronize(st) != hipSuccess ||
                        hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) rc = FS_EHIP;
    if (rc == FS_OK) {
        for (int i = 0; i < n; ++i) if (h[n + i] != 0.5 * i) rc = FS_ECOMM;
        if (h[2 * n] != 3.0 || h[2 * n + 1] != 7.0) rc = FS_ECOMM;
    }
    (void)hipFree(d);
    if (st) (void)hipStreamDestroy(st);
    (void)R->CommDestroy(comm);
    return rc;
}

int fs3d_create_slab(int nx, int ny, int nz_global, const fs3d_config* cfg, int rank, int nranks,
                     const uint8_t id[FS_COMM_ID_BYTES], fs3d_sim** out) {
    if (nranks < 1 || (nranks > 1 && !id)) return FS_EINVAL;
    return create_common(nx, ny, nz_global, cfg, rank, nranks, id, out);
}

int fs3d_slab_range(const fs3d_sim* s, int* z0, int* z1, int* halo) {
    if (!s || s->parts.empty()) return FS_EINVAL;
    int a, b;
    owned_range(s, &a, &b);
    if (z0) *z0 = a;
    if (z1) *z1 = b;
    if (halo) *halo = s->parts.front().H;
    return FS_OK;
}

void fs3d_destroy(fs3d_sim* s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->comm_red && s->comm_red != s->comm && s->rccl) (void)s->rccl->CommDestroy(s->comm_red);
    if (s->comm && s->rccl) (void)s->rccl->CommDestroy(s->comm);
    free_parts(s);
    (void)hipFree(s->dres);
    if (s->hres) (void)hipHostFree(s->hres);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto& t : s->timed) { if (t.a) (void)hipEventDestroy(t.a); if (t.b) (void)hipEventDestroy(t.b); }
    if (s->comm_stream) { (void)hipStreamSynchronize(s->comm_stream); (void)hipStreamDestroy(s->comm_stream); }
    if (s->ev_bnd) (void)hipEventDestroy(s->ev_bnd);
    if (s->ev_comm) (void)hipEventDestroy(s->ev_comm);
    if (s->ev_b) (void)hipEventDestroy(s->ev_b);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int fs3d_reset(fs3d_sim* s) {
    SIM_GUARD(s);
    FS_TRY(zero_fields(s));
    FS_TRY(init_cells(s));
    s->curv = 0;
    s->curd = 0;
    s->tilde_gen = -1;
    s->sweep_hint[0] = s->sweep_hint[1] = 1;
    s->sweep_hint_prev[0] = s->sweep_hint_prev[1] = 0;
    for (int k = 0; k < 2; ++k) s->spec_backoff[k] = s->spec_fails[k] = s->mg_backoff[k] = s->mg_fails[k] = 0;
    for (int k = 0; k < 2; ++k) s->fuse_backoff[k] = s->fuse_fails[k] = 0;
    s->mg_backoff[0] = s->mg_backoff[1] = 0;
    s->mg_fails[0] = s->mg_fails[1] = 0;
    SIM_HIP(hipMemsetAsync(s->doob, 0, sizeof(double), s->stream));
    s->hres[1] = 0.0;
    s->stats = fs3d_stats{};
    s->halo_dirty = HALO_ALL;
    FS_TRY(quiesce(s));
    return FS_OK;
}

int fs3d_set_config(fs3d_sim* s, const fs3d_config* c) {
    if (!s || !c) return FS_EINVAL;
    std::string why;
    if (!config_ok(c, &why)) return set_err(s, FS_EINVAL, why);
    if (c->dtype != s->cfg.dtype || c->local_slabs != s->cfg.local_slabs)
        return set_err(s, FS_EINVAL, "dtype and local_slabs are fixed at creation");
    if (c->ordering == FS_ORDER_LEXICOGRAPHIC && s->nparts_global > 1)
        return set_err(s, FS_EINVAL, "lexicographic order needs a single slab");
    if (c->solver == FS_SOLVER_MULTIGRID && !mg_slabs_ok(s->nz, s->nparts_global))
        return set_err(s, FS_EINVAL, "the multigrid solver needs slabs that start on even planes");
    const int dev = s->cfg.device;
    s->cfg = *c;
    s->cfg.device = dev;
    return FS_OK;
}

int fs3d_get_config(const fs3d_sim* s, fs3d_config* c) {
    if (!s || !c) return FS_EINVAL;
    *c = s->cfg;
    c->device = s->device;
    return FS_OK;
}

const char* fs3d_last_error(const fs3d_sim* s) {
    if (s) return s->err.c_str();
    return g_handleless_err.empty() ? "null handle" : g_handleless_err.c_str();
}

int fs3d_set_field(fs3d_sim* s, int which, const void* host, size_t count) {
    SIM_GUARD(s);
    if (!host || which < FS_FIELD_U || which > FS_FIELD_DENSITY) return set_err(s, FS_EINVAL, "bad field or null pointer");
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny;
    if (count != plane * (size_t)(z1 - z0)) return set_err(s, FS_EINVAL, "count does not match the owned planes");
    FS_TRY(quiesce(s));
    for (Part& p : s->parts) {
        char* dst = (char*)field_ptr(s, p, which) + (size_t)p.H * plane * s->esz;
        const char* src = (const char*)host + (size_t)(p.z0 - z0) * plane * s->esz;
        SIM_HIP(hipMemcpy(dst, src, (size_t)p.own_planes() * plane * s->esz, hipMemcpyHostToDevice));
    }
    s->halo_dirty = HALO_ALL;
    return FS_OK;
}

int fs3d_get_field(fs3d_sim* s, int which, void* host, size_t count) {
    SIM_GUARD(s);
    if (!host || which < FS_FIELD_U || which > FS_FIELD_VORTICITY) return set_err(s, FS_EINVAL, "bad field or null pointer");
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny;
    if (count != plane * (size_t)(z1 - z0)) return set_err(s, FS_EINVAL, "count does not match the owned planes");
    if (which == FS_FIELD_VORTICITY) FS_TRY(DISPATCH(s, vorticity, s, true, nullptr));
    if (which == FS_FIELD_DIVERGENCE) {
        FS_TRY(prepare(s));
        for (Part& p : s->parts) SIM_HIP(hipMemsetAsync(p.tmp[0], 0, p.n * s->esz, s->stream));
        FS_TRY(DISPATCH(s, residual, s, nullptr, nullptr, true));
    }
    FS_TRY(quiesce(s));
    for (Part& p : s->parts) {
        const char* src = (const char*)field_ptr(s, p, which) + (size_t)p.H * plane * s->esz;
        char* dst = (char*)host + (size_t)(p.z0 - z0) * plane * s->esz;
        SIM_HIP(hipMemcpy(dst, src, (size_t)p.own_planes() * plane * s->esz, hipMemcpyDeviceToHost));
    }
    return FS_OK;
}

int fs3d_get_field_plane(fs3d_sim* s, int which, int z, void* host, size_t count) {
    SIM_GUARD(s);
    if (!host || which < FS_FIELD_U || which > FS_FIELD_DENSITY) return set_err(s, FS_EINVAL, "bad field or null pointer");
    const size_t plane = (size_t)s->nx * s->ny;
    if (count != plane) return set_err(s, FS_EINVAL, "count must be nx*ny");
    FS_TRY(quiesce(s));
    for (Part& p : s->parts)
        if (z >= p.z0 && z < p.z1) {
            const char* src = (const char*)field_ptr(s, p, which) + (size_t)(z - p.g.gz0) * plane * s->esz;
            SIM_HIP(hipMemcpy(host, src, plane * s->esz, hipMemcpyDeviceToHost));
            return FS_OK;
        }
    return set_err(s, FS_EINVAL, "plane not owned by this handle");
}

int fs3d_set_cell_types(fs3d_sim* s, const uint8_t* host, size_t count) {
    SIM_GUARD(s);
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny;
    if (!host || count != plane * (size_t)(z1 - z0)) return set_err(s, FS_EINVAL, "count does not match the owned planes");
    FS_TRY(quiesce(s));
    for (Part& p : s->parts) {
        std::vector<uint8_t> h((size_t)p.own_planes() * plane);
        for (int z = p.z0; z < p.z1; ++z)
            for (int y = 0; y < s->ny; ++y)
                for (int x = 0; x < s->nx; ++x) {
                    const bool shell = z == 0 || z == s->nz - 1 || y == 0 || y == s->ny - 1 || x == 0 || x == s->nx - 1;
                    uint8_t t = host[(size_t)(z - z0) * plane + (size_t)y * s->nx + x];
                    if (t > FS_CELL_STATIC_FLUID) return set_err(s, FS_EINVAL, "cell type out of range");
                    h[(size_t)(z - p.z0) * plane + (size_t)y * s->nx + x] = shell ? (uint8_t)FS_CELL_OBJECT : t;
                }
        SIM_HIP(hipMemcpy(p.cells + (size_t)p.H * plane, h.data(), h.size(), hipMemcpyHostToDevice));
    }
    s->cells_dirty = true;
    return FS_OK;
}

int fs3d_get_cell_types(fs3d_sim* s, uint8_t* host, size_t count) {
    SIM_GUARD(s);
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny;
    if (!host || count != plane * (size_t)(z1 - z0)) return set_err(s, FS_EINVAL, "count does not match the owned planes");
    FS_TRY(quiesce(s));
    for (Part& p : s->parts)
        SIM_HIP(hipMemcpy(host + (size_t)(p.z0 - z0) * plane, p.cells + (size_t)p.H * plane, (size_t)p.own_planes() * plane,
                          hipMemcpyDeviceToHost));
    return FS_OK;
}

int fs3d_sphere_emitter(fs3d_sim* s, float radius, float cx, float cy, float cz, float strength) {
    SIM_GUARD(s);
    if (!(radius > 0)) return set_err(s, FS_EINVAL, "radius must be positive");
    s->halo_dirty = HALO_ALL;
    return DISPATCH(s, sphere_emitter, s, radius, cx, cy, cz, strength);
}

int fs3d_solid_sphere(fs3d_sim* s, float radius, float cx, float cy, float cz) {
    SIM_GUARD(s);
    if (!(radius > 0)) return set_err(s, FS_EINVAL, "radius must be positive");
    for (Part& p : s->parts) {
        Stamp st;
        sphere_cells(s, p, radius, cx, cy, cz, &st);
        FS_TRY(DISPATCH(s, scatter_to, s, p, st, -1, 0.0f, FS_CELL_OBJECT));
    }
    s->cells_dirty = true;
    return FS_OK;
}

int fs3d_sphere_influence(fs3d_sim* s, float radius) {
    if (!s) return FS_EINVAL;
    return fs3d_sphere_emitter(s, radius, s->nx / 2.0f, radius, s->nz / 2.0f, 20.0f);   // main.cpp:46-55
}

int fs3d_fill_sphere(fs3d_sim* s, float radius, float cx, float cy, float cz, int iter) {
    SIM_GUARD(s);
    if (iter % 100 != 0) return FS_OK;   // main.cpp:69
    s->halo_dirty = HALO_ALL;
    return DISPATCH(s, fill_sphere, s, radius, cx, cy, cz);
}

int fs3d_fill_sphere_influence(fs3d_sim* s, float radius, int iter) {
    if (!s) return FS_EINVAL;
    return fs3d_fill_sphere(s, radius, s->nx / 2.0f, radius, s->nz / 2.0f, iter);
}

int fs3d_bottom_influence(fs3d_sim* s) {
    SIM_GUARD(s);
    s->halo_dirty = HALO_ALL;
    return DISPATCH(s, scene_kernel, s, 0, nullptr);
}

int fs3d_clear(fs3d_sim* s) {
    SIM_GUARD(s);
    s->halo_dirty = HALO_ALL;
    return DISPATCH(s, scene_kernel, s, 1, nullptr);
}

int fs3d_apply_source(fs3d_sim* s, double xmin, double xmax, double ymin, double ymax, double zmin, double zmax) {
    SIM_GUARD(s);
    // include/Grid2.h:56-57: integer truncation of fraction * resolution, half-open ranges
    int box[6] = {(int)(xmin * s->nx), (int)(xmax * s->nx), (int)(ymin * s->ny), (int)(ymax * s->ny),
                  (int)(zmin * s->nz), (int)(zmax * s->nz)};
    if (box[0] < 0 || box[2] < 0 || box[4] < 0 || box[1] > s->nx || box[3] > s->ny || box[5] > s->nz)
        return set_err(s, FS_EINVAL, "source box outside the grid");
    s->halo_dirty = HALO_ALL;
    return DISPATCH(s, scene_kernel, s, 2, box);
}

int fs3d_gauss_seidel_projection(fs3d_sim* s, double* change) {
    SIM_GUARD(s);
    FS_TRY(prepare(s));
    double c = 0;
    FS_TRY(DISPATCH(s, sweep_with_change, s, &c));
    halos_after_sweeps(s);
    if (change) *change = c;
    return FS_OK;
}

int fs3d_sweeps(fs3d_sim* s, int n) {
    SIM_GUARD(s);
    FS_TRY(prepare(s));
    FS_TRY(DISPATCH(s, sweeps_n, s, n));
    halos_after_sweeps(s);
    return FS_OK;
}

int fs3d_time_sweep_pairs(fs3d_sim* s, int npairs, double* ms) {
    SIM_GUARD(s);
    if (!pairs_possible(s)) return set_err(s, FS_ESTATE, "two sweeps per pass need a single domain, red-black order, FS_SWEEP_FUSED and an even nx");
    FS_TRY(prepare(s));
    SIM_HIP(hipEventRecord(s->ev[0], s->stream));
    for (int i = 0; i < npairs; ++i) FS_TRY(DISPATCH(s, sweep_pair, s));
    SIM_HIP(hipEventRecord(s->ev[1], s->stream));
    SIM_HIP(hipEventSynchronize(s->ev[1]));
    float t = 0;
    SIM_HIP(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
    if (ms) *ms = t;
    return FS_OK;
}

// The three velocity fields streamed from one generation to the other: what any pass of a sweep has to do at least.
// One 16-byte element of each field per thread and no loop, so that consecutive workgroups form one compact front
// through each array: of the plain copies tried (grid-stride loops, a contiguous piece per workgroup, tiles marched
// along z in chunks of 1 to 512 planes, 256 to 1024 threads, plain or non-temporal accesses, base addresses skewed
// against each other) this is the fastest at 512^3, 5.6 TB/s.  (The same shape on ONE array runs at 6.3 TB/s, the
// figure usually quoted for a "float4 copy": three arrays read and three written at once cost the memory system an
// eighth of its rate whatever the shape.)
__global__ void __launch_bounds__(1024)
k_copy_fields(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c, uint4* __restrict__ x,
              uint4* __restrict__ y, uint4* __restrict__ z, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) { x[i] = a[i]; y[i] = b[i]; z[i] = c[i]; }
}

int fs3d_time_field_copies(fs3d_sim* s, int n, double* ms) {
    SIM_GUARD(s);
    if (n < 1) return set_err(s, FS_EINVAL, "n must be >= 1");
    FS_TRY(prepare(s));
    FS_TRY(join_halos(s));
    if (s->tilde_gen == (s->curv ^ 1)) tilde_to_tmp(s);   // the other generation is about to be overwritten
    SIM_HIP(hipEventRecord(s->ev[0], s->stream));
    for (int i = 0; i < n; ++i)
        for (Part& p : s->parts) {
            const size_t n16 = p.n * s->esz / 16;   // (the tail of an array that is no multiple of 16 bytes is not part of the measurement)
            hipLaunchKernelGGL(k_copy_fields, dim3((unsigned)((n16 + 1023) / 1024)), dim3(1024), 0, s->stream, (const uint4*)p.f[s->curv][0],
                               (const uint4*)p.f[s->curv][1], (const uint4*)p.f[s->curv][2], (uint4*)p.f[s->curv ^ 1][0],
                               (uint4*)p.f[s->curv ^ 1][1], (uint4*)p.f[s->curv ^ 1][2], n16);
        }
    SIM_HIP(hipGetLastError());
    SIM_HIP(hipEventRecord(s->ev[1], s->stream));
    SIM_HIP(hipEventSynchronize(s->ev[1]));
    float t = 0;
    SIM_HIP(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
    if (ms) *ms = t;
    return FS_OK;
}

int fs3d_time_sweeps(fs3d_sim* s, int n, double* ms) {
    SIM_GUARD(s);
    FS_TRY(prepare(s));
    SIM_HIP(hipEventRecord(s->ev[0], s->stream));
    for (int i = 0; i < n; ++i) FS_TRY(DISPATCH(s, sweep_once, s));
    FS_TRY(join_halos(s));   // the time includes the halo traffic of the last sweep
    SIM_HIP(hipEventRecord(s->ev[1], s->stream));
    SIM_HIP(hipEventSynchronize(s->ev[1]));
    float t = 0;
    SIM_HIP(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
    halos_after_sweeps(s);
    if (ms) *ms = t;
    return FS_OK;
}

int fs3d_time_advections(fs3d_sim* s, int n, double t_step, int reflected, double* ms) {
    SIM_GUARD(s);
    if (n < 1) return set_err(s, FS_EINVAL, "n must be >= 1");
    FS_TRY(prepare(s, HALO_ALL));
    const bool refl = reflected != 0;
    if (refl) {   // any field will do as the unprojected one: the scratch velocity is a copy of the current one
        if (s->cfg.advect_kernel != FS_ADVECT_TILED) return set_err(s, FS_ESTATE, "the reflected advection is the staged kernel's");
        s->tilde_gen = -1;
        for (Part& p : s->parts)
            for (int k = 0; k < 3; ++k)
                SIM_HIP(hipMemcpyAsync(p.tmp[k], p.f[s->curv][k], p.n * s->esz, hipMemcpyDeviceToDevice, s->stream));
        s->halo_dirty &= ~HALO_TMP;
    }
    FS_TRY(join_halos(s));
    SIM_HIP(hipEventRecord(s->ev[0], s->stream));
    for (int i = 0; i < n; ++i) {
        if (refl) FS_TRY(DISPATCH(s, advect, s, t_step, true, 0, false, true, nullptr));
        else FS_TRY(DISPATCH(s, advect, s, t_step, false, 1, false, false, nullptr));
    }
    FS_TRY(join_halos(s));
    SIM_HIP(hipEventRecord(s->ev[1], s->stream));
    SIM_HIP(hipEventSynchronize(s->ev[1]));
    float t = 0;
    SIM_HIP(hipEventElapsedTime(&t, s->ev[0], s->ev[1]));
    if (ms) *ms = t;
    return FS_OK;
}

int fs3d_project(fs3d_sim* s, int* sweeps, double* metric) {
    SIM_GUARD(s);
    s->resid_dt = 1.0;
    return DISPATCH(s, project, s, sweeps, metric);
}

int fs3d_residual(fs3d_sim* s, double* l2, double* max_abs_div) {
    SIM_GUARD(s);
    FS_TRY(prepare(s));
    return DISPATCH(s, residual, s, l2, max_abs_div, false);
}

int fs3d_semi_lagrange(fs3d_sim* s, double t_step) {
    SIM_GUARD(s);
    return DISPATCH(s, advect, s, t_step, false, 1);
}

int fs3d_sweep_advect(fs3d_sim* s, double t_step, double* residual_l2, int* served) {
    SIM_GUARD(s);
    if (!fused_step_possible(s))
        return set_err(s, FS_ESTATE, "one sweep and the advection in one launch need FS_F32, red-black order, FS_SWEEP_FUSED, FS_ADVECT_TILED and a single domain");
    double sum = 0;
    bool bad = false;
    FS_TRY(sweep_advect(s, t_step, 1, &sum, &bad));
    if (bad) {   // back to the state before the call
        s->curv ^= 1;
        s->curd ^= 1;
    }
    if (served) *served = bad ? 0 : 1;
    if (residual_l2) *residual_l2 = sqrt(sum) / ((double)(s->nx - 2) * (s->ny - 2) * (s->nz - 2));
    return FS_OK;
}

int fs3d_update(fs3d_sim* s, double t_step) {
    SIM_GUARD(s);
    s->stats.sweeps_step = 0;
    s->resid_dt = t_step > 0 ? t_step : 1.0;
    FS_TRY(DISPATCH(s, project, s, nullptr, nullptr));
    return DISPATCH(s, advect, s, t_step, false, 1);
}

int fs3d_advance(fs3d_sim* s, double dt) {
    SIM_GUARD(s);
    return DISPATCH(s, step, s, dt);
}

int fs3d_mg_vcycle(fs3d_sim* s) {
    SIM_GUARD(s);
    if (s->cfg.ordering != FS_ORDER_RED_BLACK || !mg_slabs_ok(s->nz, s->nparts_global))
        return set_err(s, FS_ESTATE, "the multigrid cycle needs red-black ordering and slabs that start on even planes");
    FS_TRY(prepare(s));
    return DISPATCH(s, mg_vcycle, s);
}

int fs3d_diagnostics(fs3d_sim* s, fs3d_diag* out) {
    SIM_GUARD(s);
    if (!out) return set_err(s, FS_EINVAL, "null pointer");
    FS_TRY(DISPATCH(s, energy_mass, s, 0, &out->kinetic_energy, &out->max_speed));
    FS_TRY(DISPATCH(s, energy_mass, s, 1, &out->density_sum, nullptr));
    return DISPATCH(s, vorticity, s, false, &out->enstrophy);
}

int fs3d_get_density_downsampled(fs3d_sim* s, int factor, void* host, size_t count) {
    SIM_GUARD(s);
    if (!host || factor < 1) return set_err(s, FS_EINVAL, "null pointer or factor < 1");
    return DISPATCH(s, downsample, s, factor, host, count);
}

// Restart file: a 64-byte header, then the cell types (1 byte per cell) and u, v, w, density of the whole
// grid, x fastest.  Every rank writes (reads) its owned planes at their place in the one file.
namespace {
// The schedule of residual evaluations depends on a few small counters; a continued run repeats the
// uninterrupted one only if they come along.
uint32_t pack_schedule(const fs3d_sim* s) {
    uint32_t v = 0;
    for (int k = 0; k < 2; ++k) {
        const uint32_t f = (uint32_t)(s->spec_backoff[k] & 15) | (uint32_t)(s->spec_fails[k] & 3) << 4 |
                           (uint32_t)(s->mg_backoff[k] & 15) << 10 |
                           (uint32_t)(s->mg_fails[k] & 3) << 14;
        v |= f << (16 * k);
    }
    return v;
}
void unpack_schedule(fs3d_sim* s, uint32_t v) {
    for (int k = 0; k < 2; ++k) {
        const uint32_t f = v >> (16 * k);
        s->spec_backoff[k] = f & 15; s->spec_fails[k] = (f >> 4) & 3;
        s->mg_backoff[k] = (f >> 10) & 15; s->mg_fails[k] = (f >> 14) & 3;
    }
}

struct RestartHeader {
    char magic[8];            // "FSAMD3D"
    int32_t version, dtype, nx, ny, nz, hint0, hint1, prev0;
    int64_t step;
    double time;
    int32_t prev1;
    uint32_t sched;           // the counters of the evaluation schedule, see pack_schedule

};
static_assert(sizeof(RestartHeader) == 64, "restart header is 64 bytes");

int file_io(fs3d_sim* s, int fd, bool write, void* buf, size_t bytes, off_t at) {
    char* c = (char*)buf;
    while (bytes > 0) {
        const ssize_t n = write ? pwrite(fd, c, bytes, at) : pread(fd, c, bytes, at);
        if (n <= 0) return set_err(s, FS_ESTATE, write ? "short write to the restart file" : "short read from the restart file");
        c += n; at += n; bytes -= (size_t)n;
    }
    return FS_OK;
}
}  // namespace

int fs3d_save(fs3d_sim* s, const char* path) {
    SIM_GUARD(s);
    if (!path) return set_err(s, FS_EINVAL, "null path");
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny, cells = plane * s->nz, mine = plane * (size_t)(z1 - z0);
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return set_err(s, FS_ESTATE, "cannot open the restart file for writing");
    int rc = FS_OK;
    // the file has exactly the size of header + payload, whatever was at the path before (every rank sets the
    // same size, so the order in which the ranks arrive does not matter)
    const off_t total = (off_t)(sizeof(RestartHeader) + cells + 4 * cells * s->esz);
    if (ftruncate(fd, total) != 0) rc = set_err(s, FS_ESTATE, "cannot size the restart file");
    if (rc == FS_OK && s->rank == 0) {
        RestartHeader h{};
        memcpy(h.magic, "FSAMD3D", 8);
        h.version = 1; h.dtype = s->cfg.dtype; h.nx = s->nx; h.ny = s->ny; h.nz = s->nz;
        h.hint0 = s->sweep_hint[0]; h.hint1 = s->sweep_hint[1];
        h.prev0 = s->sweep_hint_prev[0]; h.prev1 = s->sweep_hint_prev[1];
        h.sched = pack_schedule(s);
        h.step = s->stats.step; h.time = s->stats.time;
        rc = file_io(s, fd, true, &h, sizeof(h), 0);
    }
    std::vector<char> buf(mine * s->esz);
    if (rc == FS_OK) rc = fs3d_get_cell_types(s, (uint8_t*)buf.data(), mine);
    if (rc == FS_OK) rc = file_io(s, fd, true, buf.data(), mine, (off_t)(sizeof(RestartHeader) + plane * z0));
    for (int k = 0; k < 4 && rc == FS_OK; ++k) {
        rc = fs3d_get_field(s, k, buf.data(), mine);
        if (rc == FS_OK)
            rc = file_io(s, fd, true, buf.data(), mine * s->esz,
                         (off_t)(sizeof(RestartHeader) + cells + ((size_t)k * cells + plane * z0) * s->esz));
    }
    if (close(fd) != 0 && rc == FS_OK) rc = set_err(s, FS_ESTATE, "closing the restart file failed");
    return rc;
}

int fs3d_load(fs3d_sim* s, const char* path) {
    SIM_GUARD(s);
    if (!path) return set_err(s, FS_EINVAL, "null path");
    int z0, z1;
    owned_range(s, &z0, &z1);
    const size_t plane = (size_t)s->nx * s->ny, cells = plane * s->nz, mine = plane * (size_t)(z1 - z0);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_err(s, FS_ESTATE, "cannot open the restart file");
    RestartHeader h{};
    int rc = file_io(s, fd, false, &h, sizeof(h), 0);
    const off_t fsize = lseek(fd, 0, SEEK_END);
    if (rc == FS_OK && (memcmp(h.magic, "FSAMD3D", 8) != 0 || h.version != 1))
        rc = set_err(s, FS_EINVAL, "not a restart file of this library");
    if (rc == FS_OK && (h.nx != s->nx || h.ny != s->ny || h.nz != s->nz || h.dtype != s->cfg.dtype))
        rc = set_err(s, FS_EINVAL, "the restart file has another grid size or element type");
    if (rc == FS_OK && fsize != (off_t)(sizeof(RestartHeader) + cells + 4 * cells * s->esz))
        rc = set_err(s, FS_EINVAL, "the restart file is incomplete or has trailing bytes");
    std::vector<char> buf(mine * s->esz);
    if (rc == FS_OK) rc = file_io(s, fd, false, buf.data(), mine, (off_t)(sizeof(RestartHeader) + plane * z0));
    if (rc == FS_OK) rc = fs3d_set_cell_types(s, (const uint8_t*)buf.data(), mine);
    for (int k = 0; k < 4 && rc == FS_OK; ++k) {
        rc = file_io(s, fd, false, buf.data(), mine * s->esz,
                     (off_t)(sizeof(RestartHeader) + cells + ((size_t)k * cells + plane * z0) * s->esz));
        if (rc == FS_OK) rc = fs3d_set_field(s, k, buf.data(), mine);
    }
    (void)close(fd);
    if (rc != FS_OK) return rc;
    s->sweep_hint[0] = std::max(1, h.hint0); s->sweep_hint[1] = std::max(1, h.hint1);
    s->sweep_hint_prev[0] = h.prev0; s->sweep_hint_prev[1] = h.prev1;
    unpack_schedule(s, h.sched);
    s->stats = fs3d_stats{};
    s->stats.step = h.step; s->stats.time = h.time;
    // a loaded state starts with a clean out-of-halo record, like a reset one
    SIM_HIP(hipMemsetAsync(s->doob, 0, sizeof(double), s->stream));
    s->hres[1] = 0.0;
    FS_TRY(quiesce(s));
    return FS_OK;
}

int fs3d_get_stats(fs3d_sim* s, fs3d_stats* out) {
    SIM_GUARD(s);
    if (!out) return FS_EINVAL;
    FS_TRY(quiesce(s));
    float a = 0, b = 0, c = 0;
    if (s->stats.step > 0) {
        (void)hipEventElapsedTime(&a, s->ev[0], s->ev[1]);
        (void)hipEventElapsedTime(&b, s->ev[1], s->ev[2]);
        (void)hipEventElapsedTime(&c, s->ev[0], s->ev[3]);
    }
    s->stats.ms_project = a; s->stats.ms_advect = b; s->stats.ms_step = c;
    *out = s->stats;
    return FS_OK;
}

int fs3d_comm_timing(fs3d_sim* s, int on) {
    SIM_GUARD(s);
    FS_TRY(quiesce(s));
    for (auto& t : s->timed) timed_resolve(s, t);
    s->comm_timing = on != 0;
    return FS_OK;
}

int fs3d_get_comm_stats(fs3d_sim* s, fs3d_comm_stats* out, int reset) {
    SIM_GUARD(s);
    if (!out) return set_err(s, FS_EINVAL, "null pointer");
    FS_TRY(quiesce(s));
    SIM_HIP(hipStreamSynchronize(s->comm_stream));
    for (auto& t : s->timed) timed_resolve(s, t);
    *out = s->cstats;
    if (reset) s->cstats = fs3d_comm_stats{};
    return FS_OK;
}

int fs3d_synchronize(fs3d_sim* s) {
    SIM_GUARD(s);
    // local: this rank's own out-of-halo flag, and what the last reduction over the ranks reported
    FS_TRY(quiesce(s));
    if (s->nparts_global > 1) {
        int flag = 0;
        SIM_HIP(hipMemcpy(&flag, s->doob, sizeof(int), hipMemcpyDeviceToHost));
        if (flag) s->hres[1] = 1.0;
    }
    return oob_status(s);
}

int fs3d_synchronize_all(fs3d_sim* s) {
    SIM_GUARD(s);
    // the out-of-halo flag is added up over the ranks, so that all of them return the same status
    if (s->nparts_global > 1) FS_TRY(queue_reduction(s, false));
    FS_TRY(quiesce(s));
    return oob_status(s);
}


}  // extern "C"

// ===========================================================================
// fs3d_residual_noalloc: the residual of fs3d_residual for callers that cannot allocate (code under HIP
// graph capture, tight serving loops).  The caller passes device arrays; everything else the kernels need
// beyond inputs and outputs (the flag bytes derived from the cell types, the per-workgroup partial sums
// and the combined pair) lives in a file-scope device array of fixed size.  Every GPU holds its own copy
// of that array at its own address: the address is looked up once per device, with that device current,
// and only the looked-up pointer is ever handed to a kernel (the array's name is a host-side stand-in,
// not a device address).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race.
namespace {

constexpr size_t kNoallocMaxCells = (size_t)FS_NOALLOC_MAX_CELLS;
constexpr size_t kNoallocPartials = 2 * (size_t)kReduceBlocks * sizeof(double);
constexpr size_t kNoallocBytes = kNoallocMaxCells + kNoallocPartials + 256;   // flags | partials | pair

static __device__ __align__(256) unsigned char scratch[kNoallocBytes];

constexpr int kNoallocMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kNoallocMaxDevices] = {nullptr};

// The device address of `scratch` on `device`, which must be the current device.
int scratch_on_device(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            g_handleless_err = hip_err("hipGetSymbolAddress(scratch)", e, __FILE__, __LINE__);
            return FS_EHIP;
        }
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return FS_OK;
}

// out[0] = sqrt(sum) / #interior (3_fluid/FluidSim.cpp:40-43), out[1] = max |div|
__global__ void k_residual_finish(const double* __restrict__ pair, double ninterior, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = sqrt(pair[0]) / ninterior;
        out[1] = pair[1];
    }
}

template <typename T>
int residual_noalloc(int nx, int ny, int nz, const void* u, const void* v, const void* w, const uint8_t* cells,
                     unsigned char* sc, double* out, hipStream_t st) {
    Grid g{};
    g.nx = nx; g.ny = ny; g.nzl = nz; g.gz0 = 0; g.gnz = nz;
    g.sy = nx; g.sz = (int64_t)nx * ny;
    uint8_t* flags = sc;
    double* partials = reinterpret_cast<double*>(sc + kNoallocMaxCells);
    double* pair = reinterpret_cast<double*>(sc + kNoallocMaxCells + kNoallocPartials);
    // every kernel below writes all of what the next one reads (flags of every cell, one partial pair per
    // workgroup, the combined pair), so nothing has to start at zero
    dim3 b(64, 4, 1), gf(div_up(nx, 64), div_up(ny, 4), nz);
    hipLaunchKernelGGL(k_build_flags, gf, b, 0, st, g, cells, flags, (uint8_t*)nullptr);
    hipLaunchKernelGGL(k_div_sumsq<T>, dim3(kReduceBlocks), dim3(256), 0, st, g, (const T*)u, (const T*)v, (const T*)w,
                       (const uint8_t*)flags, 1, nz - 1, partials, (T*)nullptr);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, (const double*)partials, kReduceBlocks, pair);
    hipLaunchKernelGGL(k_residual_finish, dim3(1), dim3(64), 0, st, (const double*)pair,
                       (double)(nx - 2) * (ny - 2) * (nz - 2), out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_handleless_err = hip_err("kernel launch", e, __FILE__, __LINE__);
        return FS_EHIP;
    }
    return FS_OK;
}

}  // namespace

extern "C" int fs3d_residual_noalloc(int device, int nx, int ny, int nz, int dtype, const void* d_u, const void* d_v,
                                     const void* d_w, const uint8_t* d_cells, double* d_out, void* stream) {
    g_handleless_err.clear();
    if (nx < 3 || ny < 3 || nz < 3 || (dtype != FS_F32 && dtype != FS_F64) || !d_u || !d_v || !d_w || !d_cells || !d_out) {
        g_handleless_err = "fs3d_residual_noalloc: bad size, element type or null pointer";
        return FS_EINVAL;
    }
    if ((size_t)nx * (size_t)ny * (size_t)nz > kNoallocMaxCells) {
        char buf[160];
        snprintf(buf, sizeof buf, "fs3d_residual_noalloc: %d x %d x %d cells exceed the fixed scratch of %zu cells "
                 "(FS_NOALLOC_MAX_CELLS); use fs3d_residual on a handle", nx, ny, nz, kNoallocMaxCells);
        g_handleless_err = buf;
        return FS_EINVAL;
    }
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { g_handleless_err = "hipGetDevice failed"; return FS_EHIP; }
    if (dev >= kNoallocMaxDevices) { g_handleless_err = "device ordinal beyond the table of scratch addresses"; return FS_EINVAL; }
    DeviceGuard guard(dev);
    if (!guard.ok) { g_handleless_err = "hipSetDevice failed"; return FS_EHIP; }
    unsigned char* sc = nullptr;
    FS_TRY(scratch_on_device(dev, &sc));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == FS_F64) return residual_noalloc<double>(nx, ny, nz, d_u, d_v, d_w, d_cells, sc, d_out, st);
    return residual_noalloc<float>(nx, ny, nz, d_u, d_v, d_w, d_cells, sc, d_out, st);
}
