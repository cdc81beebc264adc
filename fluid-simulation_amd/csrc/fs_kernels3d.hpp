// Device kernels of the 3D MAC-grid fluid step for gfx950 (wave64).
//
// Layout: every field is a dense array of nx*ny*nzl scalars, x fastest
// (i = (z*ny + y)*nx + x, the reference's index(), advection-reflection/
// main.cpp:319-325).  nzl counts the planes held by this slab, halo planes
// included; gz0 is the global z coordinate of local plane 0 (0 on a single
// GPU).  u[i], v[i], w[i] live on the low-x/low-y/low-z face of cell i.
//
// All kernels are HBM-bandwidth bound: x runs along the 64 lanes of a wave so
// that every wave-instruction touches whole 256-byte segments.
//
// Bit parity with the CPU restatement needs IEEE division and no fused
// multiply-add: build with -ffp-contract=off and without fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsamd {

// Per-cell flag byte derived from the cell types (main.cpp:209-215, :227-232):
// bits 0..5 = the x-,x+,y-,y+,z-,z+ neighbour is Fluid; bit 6 = the cell is
// Fluid; bit 7 = the cell is Object.  popcount(bits 0..5) is the reference's s.
enum : uint32_t {
    FL_XM = 1u, FL_XP = 2u, FL_YM = 4u, FL_YP = 8u, FL_ZM = 16u, FL_ZP = 32u,
    FL_NBR = 63u, FL_SELF_FLUID = 64u, FL_SELF_OBJECT = 128u
};

struct Grid {
    int nx, ny, nzl;   // local extents (nzl includes halo planes)
    int gz0;           // global z of local plane 0
    int gnz;           // global number of planes
    int64_t sy, sz;    // strides: sy = nx, sz = nx*ny
};

__device__ __forceinline__ int64_t gidx(const Grid& g, int x, int y, int z) {
    return (int64_t)z * g.sz + (int64_t)y * g.sy + x;
}

// ---------------------------------------------------------------------------
// cell types -> flags.  Planes outside the global domain and neighbours outside
// the local array count as Object.
//
// vflags (optional) receives the "visit byte" of each cell, which is all a Gauss-Seidel visit
// (main.cpp:198-233) needs of the mask: bits 0..5 (FL_XM..FL_ZP) = the faces the visit moves - none of an
// Object cell or of a cell without Fluid neighbours (:199, :221), the high faces towards Fluid neighbours,
// the low ones only if the cell is Fluid itself (:227-232) - and bits 6..7 = the number of Fluid neighbours
// that count in the divisor (:222) without their face being moved (the low-side neighbours of a cell that
// is not Fluid itself).
__device__ __forceinline__ int visit_divisor(uint32_t vb) { return __popc(vb & FL_NBR) + (int)(vb >> 6); }

// (the kernels that are no templates are static: this header is part of more than one translation unit)
static __global__ void k_build_flags(Grid g, const uint8_t* __restrict__ cells, uint8_t* __restrict__ flags,
                              uint8_t* __restrict__ vflags) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= g.nx || y >= g.ny) return;
    const int64_t i = gidx(g, x, y, z);
    auto type_at = [&](int xx, int yy, int zz) -> uint32_t {
        if (xx < 0 || yy < 0 || zz < 0 || xx >= g.nx || yy >= g.ny || zz >= g.nzl) return 1u;
        return cells[gidx(g, xx, yy, zz)];
    };
    const uint32_t self = cells[i];
    uint32_t f = 0;
    if (type_at(x - 1, y, z) == 0u) f |= FL_XM;
    if (type_at(x + 1, y, z) == 0u) f |= FL_XP;
    if (type_at(x, y - 1, z) == 0u) f |= FL_YM;
    if (type_at(x, y + 1, z) == 0u) f |= FL_YP;
    if (type_at(x, y, z - 1) == 0u) f |= FL_ZM;
    if (type_at(x, y, z + 1) == 0u) f |= FL_ZP;
    if (vflags) {
        constexpr uint32_t low = FL_XM | FL_YM | FL_ZM;
        uint32_t vb = 0;
        if (self != 1u) vb = self == 0u ? f : ((f & ~low) | ((uint32_t)__popc(f & low) << 6));
        vflags[i] = (uint8_t)vb;
    }
    if (self == 0u) f |= FL_SELF_FLUID;
    if (self == 1u) f |= FL_SELF_OBJECT;
    flags[i] = (uint8_t)f;
}

// ---------------------------------------------------------------------------
// Projection, red-black order.
//
// The reference visits a cell, computes dd = div/s and pushes +dd onto the
// cell's low faces and -dd onto its high faces under the mask predicates of
// main.cpp:227-232.  Every face touches exactly one cell of each colour, so
// within one colour the six faces of a cell are read and written by that
// cell's visit alone: one colour is one race-free in-place parallel pass, and
// its result does not depend on the visiting order inside the colour.
template <typename T>
__device__ __forceinline__ T cell_dd(const Grid& g, const T* __restrict__ u, const T* __restrict__ v,
                                     const T* __restrict__ w, int64_t c, uint32_t fl, T omega) {
    // main.cpp:217-225
    const T div = (u[c + 1] - u[c]) + (v[c + g.sy] - v[c]) + (w[c + g.sz] - w[c]);
    T dd = div / (T)__popc(fl & FL_NBR);
    if (omega != (T)1) dd = omega * dd;
    return dd;
}

__device__ __forceinline__ bool relaxes(uint32_t fl) {   // main.cpp:199, :221
    return !(fl & FL_SELF_OBJECT) && (fl & FL_NBR);
}

// One thread per cell of the active colour: x = 1 + 2*t + (row offset), y in
// [1,ny-2], z in [zb, ze).  A slab's top cells write their high-z face into the
// slab's ghost copy of the interface plane.
template <typename T>
__global__ void __launch_bounds__(256)
k_rb_halfsweep(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w,
               const uint8_t* __restrict__ flags, int colour, T omega, int zb, int ze) {
    const int y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    const int z = blockIdx.z + zb;
    if (y > g.ny - 2) return;
    // first interior x of the active colour in this row: (x + y + z + gz0) & 1 == colour
    const int x = 1 + 2 * (blockIdx.x * blockDim.x + threadIdx.x) + ((1 + y + z + g.gz0 + colour) & 1);
    if (x > g.nx - 2) return;
    const int64_t i = gidx(g, x, y, z);
    const uint32_t fl = flags[i];
    if (!relaxes(fl)) return;
    const T dd = cell_dd(g, u, v, w, i, fl, omega);
    const bool self = fl & FL_SELF_FLUID;
    if ((fl & FL_XM) && self) u[i] += dd;      // main.cpp:227-232
    if (fl & FL_XP) u[i + 1] -= dd;
    if ((fl & FL_YM) && self) v[i] += dd;
    if (fl & FL_YP) v[i + g.sy] -= dd;
    if ((fl & FL_ZM) && self) w[i] += dd;
    if (fl & FL_ZP) w[i + g.sz] -= dd;
}

// After a half-sweep the two slabs either side of an interface each hold the
// fresh value of w on a complementary checkerboard half of the interface
// plane.  take_parity selects the half that the other slab updated.
template <typename T>
__global__ void k_merge_w_plane(Grid g, T* __restrict__ w, const T* __restrict__ recv, int z, int take_parity) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.nx || y >= g.ny) return;
    if (((x + y + z + g.gz0) & 1) == take_parity) w[gidx(g, x, y, z)] = recv[(int64_t)y * g.nx + x];
}

// ---------------------------------------------------------------------------
// Projection, the reference's lexicographic order (main.cpp:235), bit for bit.
//
// A visit reads and writes only the visited cell's own six faces, so two visits
// commute unless the cells share a face, and the serial sweep is fixed by one
// rule: of two face neighbours the one with the smaller index is visited first.
// Every schedule that keeps that rule gives the serial result.  Here the
// interior is cut into blocks of LB^3 cells.  Blocks on a diagonal
// bi + bj + bk = K share no face (face neighbours differ by one in exactly one
// block index), their low neighbours lie on diagonal K - 1 and their high
// neighbours on K + 1: one launch per diagonal, K ascending.  Inside a block
// the same argument holds for cells: one workgroup keeps the block's faces in
// LDS and walks the local diagonals lx + ly + lz = k with a barrier between
// them.  A sweep is nbx + nby + nbz - 2 launches instead of one per cell
// diagonal of the whole grid.
template <typename T>
struct LexiBlock {
    static constexpr int LB = 16;              // cells per block edge; LB * LB threads
    static constexpr int PU = LB + 2;          // row pitch of the u tile (LB + 1 faces per row; even pitch - 1 = odd stride)
    static constexpr int NU = PU * LB * LB;    // u[lz][ly][0..LB]
    static constexpr int NV = LB * (LB + 1) * LB;   // v[lz][0..LB][lx]
    static constexpr int NW = LB * LB * (LB + 1);   // w[0..LB][ly][lx]
    static constexpr int NF = LB * LB * LB;         // flag bytes of the block's cells
    static constexpr size_t lds_bytes() { return (size_t)(NU + NV + NW) * sizeof(T) + NF; }
};

template <typename T>
__global__ void __launch_bounds__(256)
k_lexi_blocks(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w, const uint8_t* __restrict__ flags, int K,
              int nbx, T omega) {
    using C = LexiBlock<T>;
    constexpr int LB = C::LB, PU = C::PU;
    const int bj = blockIdx.x, bk = blockIdx.y, bi = K - bj - bk;
    if (bi < 0 || bi >= nbx) return;   // uniform over the workgroup
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* tu = reinterpret_cast<T*>(lds_raw);
    T* tv = tu + C::NU;
    T* tw = tv + C::NV;
    uint8_t* tf = reinterpret_cast<uint8_t*>(tw + C::NW);
    const int x0 = 1 + bi * LB, y0 = 1 + bj * LB, z0 = 1 + bk * LB;
    // cells of this block: the part of [x0, x0+LB) x ... that is interior
    const int ex = min(LB, g.nx - 1 - x0), ey = min(LB, g.ny - 1 - y0), ez = min(LB, g.nzl - 1 - z0);
    const int tid = threadIdx.x;
    // Faces of the block's cells: u[0..ex] x ey x ez, v: ex x [0..ey] x ez, w: ex x ey x [0..ez]; all inside the grid
    // because the cells are interior.  A thread moves the column (cx, cy, 0..LB-1) of every array plus one element
    // of each array's extra layer (the high faces of the last cells); all its loads are issued before the first
    // is used, so the tile arrives in about one memory round trip.
    const int cx = tid % LB, cy = tid / LB;
    {
        T ru[LB], rv[LB], rw[LB], xu = 0, xv = 0, xw = 0;
        uint8_t rf[LB];
        const bool col = cx < ex && cy < ey;
#pragma unroll
        for (int lz = 0; lz < LB; ++lz) {
            const bool in = col && lz < ez;
            const int64_t i = in ? gidx(g, x0 + cx, y0 + cy, z0 + lz) : 0;
            ru[lz] = in ? u[i] : (T)0; rv[lz] = in ? v[i] : (T)0; rw[lz] = in ? w[i] : (T)0;
            rf[lz] = in ? flags[i] : (uint8_t)FL_SELF_OBJECT;
        }
        // extra layers: u at lx = ex for (ly, lz) = (cx, cy); v at ly = ey for (lx, lz) = (cx, cy); w at lz = ez for (lx, ly)
        if (cx < ey && cy < ez) xu = u[gidx(g, x0 + ex, y0 + cx, z0 + cy)];
        if (cx < ex && cy < ez) xv = v[gidx(g, x0 + cx, y0 + ey, z0 + cy)];
        if (cx < ex && cy < ey) xw = w[gidx(g, x0 + cx, y0 + cy, z0 + ez)];
        // A partial block keeps its extra layers inside the tile (column ex of tu, row ey of tv), where the fill
        // below would put the zeros of cells outside the block: those slots belong to the extra-layer stores that
        // follow, which other threads make, so the fill leaves them alone.  (The extra layer of tw, plane ez, is
        // written twice by the same thread, in program order.)
#pragma unroll
        for (int lz = 0; lz < LB; ++lz) {
            if (cx != ex) tu[(lz * LB + cy) * PU + cx] = ru[lz];
            if (cy != ey) tv[(lz * (LB + 1) + cy) * LB + cx] = rv[lz];
            tw[(lz * LB + cy) * LB + cx] = rw[lz];
            tf[(lz * LB + cy) * LB + cx] = rf[lz];
        }
        if (cx < ey && cy < ez) tu[(cy * LB + cx) * PU + ex] = xu;
        if (cx < ex && cy < ez) tv[(cy * (LB + 1) + ey) * LB + cx] = xv;
        if (cx < ex && cy < ey) tw[(ez * LB + cy) * LB + cx] = xw;
    }
    const int ly = tid % LB, lz = tid / LB;
    const bool row = ly < ey && lz < ez;
    __syncthreads();
    for (int k = 0; k <= ex + ey + ez - 3; ++k) {
        const int lx = k - ly - lz;
        if (row && lx >= 0 && lx < ex) {
            const uint32_t fl = tf[(lz * LB + ly) * LB + lx];
            if (relaxes(fl)) {
                T* pu = tu + (lz * LB + ly) * PU + lx;
                T* pv = tv + (lz * (LB + 1) + ly) * LB + lx;
                T* pw = tw + (lz * LB + ly) * LB + lx;
                // main.cpp:217-225
                const T div = (pu[1] - pu[0]) + (pv[LB] - pv[0]) + (pw[LB * LB] - pw[0]);
                T dd = div / (T)__popc(fl & FL_NBR);
                if (omega != (T)1) dd = omega * dd;
                const bool self = fl & FL_SELF_FLUID;
                if ((fl & FL_XM) && self) pu[0] += dd;      // main.cpp:227-232
                if (fl & FL_XP) pu[1] -= dd;
                if ((fl & FL_YM) && self) pv[0] += dd;
                if (fl & FL_YP) pv[LB] -= dd;
                if ((fl & FL_ZM) && self) pw[0] += dd;
                if (fl & FL_ZP) pw[LB * LB] -= dd;
            }
        }
        __syncthreads();
    }
    {
        const bool col = cx < ex && cy < ey;
#pragma unroll
        for (int lz2 = 0; lz2 < LB; ++lz2) {
            if (col && lz2 < ez) {
                const int64_t i = gidx(g, x0 + cx, y0 + cy, z0 + lz2);
                u[i] = tu[(lz2 * LB + cy) * PU + cx];
                v[i] = tv[(lz2 * (LB + 1) + cy) * LB + cx];
                w[i] = tw[(lz2 * LB + cy) * LB + cx];
            }
        }
        if (cx < ey && cy < ez) u[gidx(g, x0 + ex, y0 + cx, z0 + cy)] = tu[(cy * LB + cx) * PU + ex];
        if (cx < ex && cy < ez) v[gidx(g, x0 + cx, y0 + ey, z0 + cy)] = tv[(cy * (LB + 1) + ey) * LB + cx];
        if (cx < ex && cy < ey) w[gidx(g, x0 + cx, y0 + cy, z0 + ez)] = tw[(ez * LB + cy) * LB + cx];
    }
}

// The reference sums the change in one scalar of the working precision, in
// ascending cell order (main.cpp:243-249).  One wave reproduces that order:
// the lanes fetch 64 consecutive cells of an interior row, lane order is cell
// order, and every lane adds the 192 values of the chunk in that order (lane
// values read one by one into scalar registers), so the running sum is a chain
// of dependent additions as in the reference.
template <typename T>
__device__ __forceinline__ T lane_value(T v, int l) {
    if constexpr (sizeof(T) == 4) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    } else {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
}

template <typename T>
__global__ void k_change_serial(Grid g, const T* __restrict__ au, const T* __restrict__ av,
                                const T* __restrict__ aw, const T* __restrict__ bu, const T* __restrict__ bv,
                                const T* __restrict__ bw, double* __restrict__ out) {
    const int lane = threadIdx.x;
    T acc = 0;
    // interior cells in ascending index order: planes, rows, then x; cells outside the interior add nothing in the
    // reference, and adding +0 here for the lanes beyond the end of a row changes no bit (the sum is never -0)
    for (int z = 1; z < g.nzl - 1; ++z)
        for (int y = 1; y < g.ny - 1; ++y) {
            const int64_t row = gidx(g, 0, y, z);
            for (int xb = 1; xb < g.nx - 1; xb += 64) {
                const int x = xb + lane;
                T a = 0, b = 0, c = 0;
                if (x < g.nx - 1) {
                    const int64_t i = row + x;
                    a = fabs(au[i] - bu[i]); b = fabs(av[i] - bv[i]); c = fabs(aw[i] - bw[i]);
                }
#pragma unroll
                for (int l = 0; l < 64; ++l) {
                    acc += lane_value(a, l); acc += lane_value(b, l); acc += lane_value(c, l);
                }
            }
        }
    if (lane == 0) out[0] = (double)acc;
}

// ---------------------------------------------------------------------------
// Reductions: per-wave shuffle tree, per-block LDS, one partial per block; a
// second one-block kernel adds the partials in a fixed order so that results
// are reproducible from run to run.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// blockDim.x must be 256.  Returns the block total in thread 0.
__device__ __forceinline__ void block_sum_max(double& s, double& m) {
    __shared__ double ls[4], lm[4];
    s = wave_sum(s); m = wave_max(m);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { ls[wid] = s; lm[wid] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = ls[0] + ls[1] + ls[2] + ls[3];
        m = fmax(fmax(lm[0], lm[1]), fmax(lm[2], lm[3]));
    }
}

// L1 norm of (a - b) over the interior locations of planes [zb, ze).
// part[2*block] = sum, part[2*block+1] = unused max slot.
template <typename T>
__global__ void __launch_bounds__(256)
k_change_l1(Grid g, const T* __restrict__ au, const T* __restrict__ av, const T* __restrict__ aw,
            const T* __restrict__ bu, const T* __restrict__ bv, const T* __restrict__ bw, int zb, int ze,
            double* __restrict__ part) {
    const int64_t rows = (int64_t)(g.ny - 2) * (ze - zb);
    double s = 0, m = 0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int y = (int)(r % (g.ny - 2)) + 1, z = (int)(r / (g.ny - 2)) + zb;
        const int64_t row = gidx(g, 0, y, z);
        for (int x = threadIdx.x + 1; x < g.nx - 1; x += 256) {
            const int64_t i = row + x;
            s += (double)fabs(au[i] - bu[i]);
            s += (double)fabs(av[i] - bv[i]);
            s += (double)fabs(aw[i] - bw[i]);
        }
    }
    block_sum_max(s, m);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = 0; }
}

template <typename T, int N> struct alignas(N * sizeof(T)) Vec { T e[N]; };
// One N-wide store that the caches need not keep: nothing reads a swept plane again before the pass ends.
template <typename T, int N>
__device__ __forceinline__ void stream_store(T* p, const Vec<T, N>& v) {
    typedef T Native __attribute__((ext_vector_type(N)));
    Native n;
#pragma unroll
    for (int k = 0; k < N; ++k) n[k] = v.e[k];
    __builtin_nontemporal_store(n, reinterpret_cast<Native*>(p));
}

// Sum of squared divergences and max |div| over the cells a visit can relax
// (not Object, s > 0, and at least one face the visit may move: a StaticFluid
// cell moves only its high faces, main.cpp:227-232), planes [zb, ze).
// Optionally stores the divergence field.
template <typename T>
__global__ void __launch_bounds__(256)
k_div_sumsq(Grid g, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ w,
            const uint8_t* __restrict__ flags, int zb, int ze, double* __restrict__ part, T* __restrict__ div_out) {
    // A job is a strip of kRows rows marched through kPlanes planes by one workgroup: the next row of v
    // and the next plane of w that a cell reads are then read again by the same workgroup a moment later
    // (round-robin rows made every v row and w plane come from HBM twice, 21 instead of 13 B per cell).
    constexpr int kRows = 4, kPlanes = 16;
    constexpr int kVec = 16 / (int)sizeof(T);
    const bool wide = g.nx % kVec == 0;   // rows then start on 16-byte boundaries
    const int ytiles = (g.ny - 2 + kRows - 1) / kRows, zchunks = (ze - zb + kPlanes - 1) / kPlanes;
    double s = 0, m = 0;
    for (int64_t job = blockIdx.x; job < (int64_t)ytiles * zchunks; job += gridDim.x) {
        const int y0 = 1 + (int)(job % ytiles) * kRows, z0 = zb + (int)(job / ytiles) * kPlanes;
        const int yend = min(y0 + kRows, g.ny - 1);
        for (int z = z0; z < min(z0 + kPlanes, ze); ++z) {
            if (wide) {
                // kVec cells per lane with 16-byte loads, the rows of the strip side by side; the high-x face
                // of a lane's last cell is the next lane's first u, or a scalar load at the end of a wave
                const int nvx = g.nx / kVec;
                for (int idx = threadIdx.x; idx < nvx * (yend - y0); idx += 256) {
                    const int ry = idx / nvx, x = (idx - ry * nvx) * kVec;
                    const int64_t i = gidx(g, x, y0 + ry, z);
                    const Vec<T, kVec> uu = *reinterpret_cast<const Vec<T, kVec>*>(u + i);
                    const Vec<T, kVec> v0 = *reinterpret_cast<const Vec<T, kVec>*>(v + i);
                    const Vec<T, kVec> v1 = *reinterpret_cast<const Vec<T, kVec>*>(v + i + g.sy);
                    const Vec<T, kVec> w0 = *reinterpret_cast<const Vec<T, kVec>*>(w + i);
                    const Vec<T, kVec> w1 = *reinterpret_cast<const Vec<T, kVec>*>(w + i + g.sz);
                    const Vec<uint8_t, kVec> ff = *reinterpret_cast<const Vec<uint8_t, kVec>*>(flags + i);
                    T un = __shfl_down(uu.e[0], 1, 64);
                    if ((threadIdx.x & 63) == 63 && x + kVec < g.nx) un = u[i + kVec];
#pragma unroll
                    for (int k = 0; k < kVec; ++k) {
                        const T u1 = k + 1 < kVec ? uu.e[k + 1 < kVec ? k + 1 : k] : un;
                        const T div = (u1 - uu.e[k]) + (v1.e[k] - v0.e[k]) + (w1.e[k] - w0.e[k]);
                        const bool interior = x + k >= 1 && x + k < g.nx - 1;
                        if (div_out && interior) div_out[i + k] = div;
                        const uint32_t fl = ff.e[k];
                        const uint32_t movable = (fl & FL_SELF_FLUID) ? (fl & FL_NBR) : (fl & (FL_XP | FL_YP | FL_ZP));
                        if (interior && relaxes(fl) && movable) {
                            const double dv = (double)div;
                            s += dv * dv;
                            m = fmax(m, fabs(dv));
                        }
                    }
                }
                continue;
            }
            for (int y = y0; y < yend; ++y) {
                const int64_t row = gidx(g, 0, y, z);
                for (int x = threadIdx.x + 1; x < g.nx - 1; x += 256) {
                    const int64_t i = row + x;
                    const uint32_t fl = flags[i];
                    const T div = (u[i + 1] - u[i]) + (v[i + g.sy] - v[i]) + (w[i + g.sz] - w[i]);
                    if (div_out) div_out[i] = div;
                    const uint32_t movable = (fl & FL_SELF_FLUID) ? (fl & FL_NBR) : (fl & (FL_XP | FL_YP | FL_ZP));
                    if (relaxes(fl) && movable) {
                        const double dv = (double)div;
                        s += dv * dv;
                        m = fmax(m, fabs(dv));
                    }
                }
            }
        }
    }
    block_sum_max(s, m);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = m; }
}

// out[0] = sum of part[2b], out[1] = max of part[2b+1], b ascending.
static __global__ void __launch_bounds__(256) k_reduce_final(const double* __restrict__ part, int nblocks,
                                                      double* __restrict__ out) {
    double s = 0, m = 0;
    // fixed order: thread t adds blocks t, t+256, ...; then the fixed tree of block_sum_max
    for (int b = threadIdx.x; b < nblocks; b += 256) { s += part[2 * b]; m = fmax(m, part[2 * b + 1]); }
    block_sum_max(s, m);
    if (threadIdx.x == 0) { out[0] = s; out[1] = m; }
}

// ---------------------------------------------------------------------------
// Semi-Lagrangian advection (main.cpp:256-291, sampler :380-435).
//
// sample() of main.cpp:380-409 with safeSample (main.cpp:411-435) folded in:
// same backtrace, same weight grouping, same order of the eight terms; a tap
// reads zero outside the grid, on the boundary shell and for non-finite values.
// Coordinates are global; a tap the reference would read but whose plane this
// slab does not hold raises the out-of-halo flag.
//
// The work is split into prepare (backtrace, weights, validity bits, base
// address), gather (eight unconditional loads from the clamped base address
// plus fixed offsets) and combine, so that a thread has the gathers of all its
// samples in flight together instead of one dependent round trip per tap.
// Clamping the base into [0, n-2] never moves a tap the reference reads (such
// a tap has both itself and its base in range).
template <typename T>
struct Trace {
    T fu, fv, fw;
    int64_t base;
    int64_t dz1;    // elements from the plane of tap 0 to the plane of tap 3: one plane, or 0 where a slab holds
                    // only one of the two planes (the other is then a tap the reference does not read)
    uint32_t ok;    // bit k: tap k (order of main.cpp:400-407) is read by the reference
    bool oob;
};

template <typename T>
__device__ __forceinline__ Trace<T> trace_prepare(const Grid& g, int x, int y, int gz, T dx, T dy, T dz, T t_step) {
    Trace<T> tr;
    const T px = (T)x - dx * t_step;
    const T py = (T)y - dy * t_step;
    const T pz = (T)gz - dz * t_step;
    const T bx = floor(px), by = floor(py), bz = floor(pz);
    // (n - b) / (t - b) with t = b + 1: t - b is exactly 1 unless |b| >= 2^24 (2^53 in double), where
    // the reference divides by 0 or 2; dividing by 1 is exact, so the divisions run only in that case
    tr.fu = px - bx; tr.fv = py - by; tr.fw = pz - bz;
    const T ex = (bx + (T)1) - bx, ey = (by + (T)1) - by, ez = (bz + (T)1) - bz;
    if (__builtin_expect(ex != (T)1 || ey != (T)1 || ez != (T)1, 0)) {
        tr.fu = tr.fu / ex; tr.fv = tr.fv / ey; tr.fw = tr.fw / ez;
    }
    const int ibx = (int)bx, iby = (int)by, ibz = (int)bz;
    const bool x0 = ibx >= 1 && ibx < g.nx - 1, x1 = ibx >= 0 && ibx < g.nx - 2;
    const bool y0 = iby >= 1 && iby < g.ny - 1, y1 = iby >= 0 && iby < g.ny - 2;
    bool z0 = ibz >= 1 && ibz < g.gnz - 1, z1 = ibz >= 0 && ibz < g.gnz - 2;
    const int lzb = ibz - g.gz0;
    const bool h0 = lzb >= 0 && lzb < g.nzl, h1 = lzb >= -1 && lzb < g.nzl - 1;
    tr.oob = ((z0 && !h0) || (z1 && !h1)) && (x0 || x1) && (y0 || y1);
    z0 = z0 && h0; z1 = z1 && h1;
    tr.ok = (uint32_t)(x0 && y0 && z0) | ((uint32_t)(x1 && y0 && z0) << 1) | ((uint32_t)(x0 && y1 && z0) << 2) |
            ((uint32_t)(x0 && y0 && z1) << 3) | ((uint32_t)(x1 && y1 && z0) << 4) | ((uint32_t)(x1 && y0 && z1) << 5) |
            ((uint32_t)(x0 && y1 && z1) << 6) | ((uint32_t)(x1 && y1 && z1) << 7);
    // x, y: clamping the base into [0, n-2] never moves a tap the reference reads (such a tap has both itself
    // and its base in range).  z: the first and the last plane a slab holds can be read taps whose partner plane
    // lies outside the slab (and is then the shell of the grid, which is not read), so the two tap planes are
    // clamped one by one
    const int cx = min(max(ibx, 0), g.nx - 2), cy = min(max(iby, 0), g.ny - 2);
    const int cz0 = min(max(lzb, 0), g.nzl - 1), cz1 = min(max(lzb + 1, 0), g.nzl - 1);
    tr.base = gidx(g, cx, cy, cz0);
    tr.dz1 = (int64_t)(cz1 - cz0) * g.sz;
    return tr;
}

template <typename T> struct Taps { T s[8]; };

template <typename T>
__device__ __forceinline__ Taps<T> trace_gather(const Grid& g, const T* __restrict__ q, const Trace<T>& tr) {
    const T* b = q + tr.base;
    Taps<T> t;
    const T* b1 = b + tr.dz1;
    t.s[0] = b[0]; t.s[1] = b[1]; t.s[2] = b[g.sy]; t.s[3] = b1[0];
    t.s[4] = b[g.sy + 1]; t.s[5] = b1[1]; t.s[6] = b1[g.sy]; t.s[7] = b1[g.sy + 1];
    return t;
}

template <typename T>
__device__ __forceinline__ T trace_combine(const Trace<T>& tr, const Taps<T>& t) {
    T s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = (((tr.ok >> k) & 1u) && isfinite(t.s[k])) ? t.s[k] : (T)0;
    const T one = (T)1, fu = tr.fu, fv = tr.fv, fw = tr.fw;
    T r = 0;
    r += s[0] * ((one - fu) * (one - fv) * (one - fw));
    r += s[1] * ((fu) * (one - fv) * (one - fw));
    r += s[2] * ((one - fu) * (fv) * (one - fw));
    r += s[3] * ((one - fu) * (one - fv) * (fw));
    r += s[4] * ((fu) * (fv) * (one - fw));
    r += s[5] * ((fu) * (one - fv) * (fw));
    r += s[6] * ((one - fu) * (fv) * (fw));
    r += s[7] * ((fu) * (fv) * (fw));
    return r;
}

// ---- the same sampler, arranged for the vector ALU ---------------------------
// The advection kernel is bound by vector-ALU issue, not by memory, so the
// common case is kept short.  Instead of masking each of the eight taps, the
// three per-axis weight pairs are zeroed where the reference would not read the
// tap: a zero factor makes the tap's term an exact zero, which leaves the sum
// unchanged.  The test for non-finite taps is made once, on the result: a
// non-finite tap, wanted or not, always makes the raw sum non-finite, and only
// then are the taps cleaned one by one and combined again.  A backtrace that
// ends beyond 2^23 cells (where the reference's (n-b)/(t-b) of main.cpp:391-393
// stops being n-b) reads no tap at all; it keeps the reference's weights, NaN
// included, and takes the same second combine with every tap zero.  The weights
// are the same products in the same order, so the result has the bits of the
// masked form in every case.
//
// OFF is the type of a byte offset into a field: uint32_t when the arrays are
// smaller than 4 GiB (one address add per tap row instead of 64-bit arithmetic).
template <typename T, typename OFF>
struct FastTrace {
    T x0, x1, y0, y1, z0, z1;   // (1-fu), fu, (1-fv), fv, (1-fw), fw, or 0 where the tap is not read
    int cx, cy, cz;             // tap 0, clamped into the array (cz is a local plane)
    OFF dz1;                    // bytes from the plane of tap 0 to the plane of tap 3 (see Trace::dz1)
    bool oob, far;
    bool vx0, vx1, vy0, vy1, vz0, vz1;   // which taps the reference reads, per axis (for callers that look at the taps themselves)
};

template <typename T, typename OFF>
__device__ __forceinline__ OFF trace_base(const Grid& g, const FastTrace<T, OFF>& tr) {   // byte offset of tap 0
    return (((OFF)tr.cz * (OFF)g.ny + (OFF)tr.cy) * (OFF)g.nx + (OFF)tr.cx) * (OFF)sizeof(T);
}

template <typename T, typename OFF>
__device__ __forceinline__ const T* at_byte(const T* __restrict__ q, OFF off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(q) + off);
}

template <typename T, typename OFF, bool SLAB>
__device__ __forceinline__ FastTrace<T, OFF> fast_prepare(const Grid& g, int x, int y, int gz, T dx, T dy, T dz, T t_step) {
    FastTrace<T, OFF> tr;
    const T px = (T)x - dx * t_step;
    const T py = (T)y - dy * t_step;
    const T pz = (T)gz - dz * t_step;
    const T bx = floor(px), by = floor(py), bz = floor(pz);
    T fu = px - bx, fv = py - by, fw = pz - bz;
    const int ibx = (int)bx, iby = (int)by, ibz = (int)bz;
    const uint32_t mx = (uint32_t)(g.nx - 2), my = (uint32_t)(g.ny - 2), mz = (uint32_t)(g.gnz - 2);
    // tap b is read when 1 <= b <= n-2, tap b+1 when 0 <= b <= n-3
    const bool x0 = (uint32_t)ibx - 1u < mx, x1 = (uint32_t)ibx < mx;
    const bool y0 = (uint32_t)iby - 1u < my, y1 = (uint32_t)iby < my;
    bool z0 = (uint32_t)ibz - 1u < mz, z1 = (uint32_t)ibz < mz;
    const int lzb = ibz - g.gz0;
    tr.oob = false;
    if (SLAB) {   // planes of the global grid this slab does not hold
        const bool h0 = (uint32_t)lzb < (uint32_t)g.nzl, h1 = (uint32_t)lzb + 1u < (uint32_t)g.nzl;
        tr.oob = ((z0 && !h0) || (z1 && !h1)) && (x0 || x1) && (y0 || y1);
        z0 = z0 && h0; z1 = z1 && h1;
    }
    tr.vx0 = x0; tr.vx1 = x1; tr.vy0 = y0; tr.vy1 = y1; tr.vz0 = z0; tr.vz1 = z1;
    const T one = (T)1, zero = (T)0;
    tr.x0 = x0 ? one - fu : zero; tr.x1 = x1 ? fu : zero;
    tr.y0 = y0 ? one - fv : zero; tr.y1 = y1 ? fv : zero;
    tr.z0 = z0 ? one - fw : zero; tr.z1 = z1 ? fw : zero;
    tr.far = !(fmax(fmax(fabs(bx), fabs(by)), fabs(bz)) < (T)8388608);
    if (__builtin_expect(tr.far, 0)) {
        fu = fu / ((bx + one) - bx); fv = fv / ((by + one) - by); fw = fw / ((bz + one) - bz);
        tr.x0 = one - fu; tr.x1 = fu; tr.y0 = one - fv; tr.y1 = fv; tr.z0 = one - fw; tr.z1 = fw;
    }
    tr.cx = min(max(ibx, 0), g.nx - 2); tr.cy = min(max(iby, 0), g.ny - 2);
    const OFF plane = (OFF)g.nx * (OFF)g.ny * (OFF)sizeof(T);
    if (SLAB) {   // the two tap planes clamped one by one, see trace_prepare
        tr.cz = min(max(lzb, 0), g.nzl - 1);
        tr.dz1 = (OFF)(min(max(lzb + 1, 0), g.nzl - 1) - tr.cz) * plane;
    } else {
        tr.cz = min(max(lzb, 0), g.nzl - 2);
        tr.dz1 = plane;
    }
    return tr;
}

typedef float float2v __attribute__((ext_vector_type(2)));

// The eight terms of main.cpp:400-407 in their order, each weight grouped as (x*y)*z.  For 4-byte elements
// the products are formed two at a time (packed multiplies: the same IEEE products, half the instructions);
// the taps pair up the way the two-element LDS reads deliver them.
template <typename T, typename OFF>
__device__ __forceinline__ T fast_combine(const FastTrace<T, OFF>& tr, const Taps<T>& t) {
    if constexpr (sizeof(T) == 4) {
        const float2v xs = {tr.x0, tr.x1};
        const float2v a0 = xs * float2v{tr.y0, tr.y0};   // a00, a10
        const float2v a1 = xs * float2v{tr.y1, tr.y1};   // a01, a11
        const float2v z0 = {tr.z0, tr.z0}, z1 = {tr.z1, tr.z1};
        const float2v p01 = float2v{t.s[0], t.s[1]} * (a0 * z0);
        const float2v p24 = float2v{t.s[2], t.s[4]} * (a1 * z0);
        const float2v p35 = float2v{t.s[3], t.s[5]} * (a0 * z1);
        const float2v p67 = float2v{t.s[6], t.s[7]} * (a1 * z1);
        T r = 0;
        r += p01.x; r += p01.y; r += p24.x; r += p35.x; r += p24.y; r += p35.y; r += p67.x; r += p67.y;
        return r;
    } else {
        const T a00 = tr.x0 * tr.y0, a10 = tr.x1 * tr.y0, a01 = tr.x0 * tr.y1, a11 = tr.x1 * tr.y1;
        T r = 0;
        r += t.s[0] * (a00 * tr.z0);
        r += t.s[1] * (a10 * tr.z0);
        r += t.s[2] * (a01 * tr.z0);
        r += t.s[3] * (a00 * tr.z1);
        r += t.s[4] * (a11 * tr.z0);
        r += t.s[5] * (a10 * tr.z1);
        r += t.s[6] * (a01 * tr.z1);
        r += t.s[7] * (a11 * tr.z1);
        return r;
    }
}

template <typename T, typename OFF>
__device__ __forceinline__ Taps<T> fast_gather(const T* __restrict__ q, OFF base, OFF sy, OFF sz) {
    const T* r00 = at_byte(q, base);
    const T* r10 = at_byte(q, base + sy);
    const T* r01 = at_byte(q, base + sz);
    const T* r11 = at_byte(q, base + sz + sy);
    Taps<T> t;
    t.s[0] = r00[0]; t.s[1] = r00[1]; t.s[2] = r10[0]; t.s[3] = r01[0];
    t.s[4] = r10[1]; t.s[5] = r01[1]; t.s[6] = r11[0]; t.s[7] = r11[1];
    return t;
}

// combine, with the second combine for the rare cases
template <typename T, typename OFF>
__device__ __forceinline__ T fast_finish(const FastTrace<T, OFF>& tr, Taps<T>& t) {
    T r = fast_combine(tr, t);
    if (__builtin_expect(tr.far || !isfinite(r), 0)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t.s[k] = (tr.far || !isfinite(t.s[k])) ? (T)0 : t.s[k];
        r = fast_combine(tr, t);
    }
    return r;
}

template <typename T, typename OFF>
__device__ __forceinline__ T fast_sample(const Grid& g, const T* __restrict__ q, const FastTrace<T, OFF>& tr, OFF sy, OFF sz) {
    (void)sz;
    Taps<T> t = fast_gather<T, OFF>(q, trace_base(g, tr), sy, tr.dz1);
    return fast_finish(tr, t);
}

// One thread per location of planes [zb, ze): carries (qu,qv,qw,qd) along the
// velocity (U,V,W) into (ou,ov,ow,od).  Locations the reference does not
// advect (non-Fluid faces, Object densities, the shell) copy q.  density_mode 1
// is the reference's cell-centre velocity: the cell's own, already advected
// low faces averaged with the not yet advected high faces (main.cpp:285-287 in
// serial order); density_mode 0 averages the carrying velocity's faces.
// (cu,cv,cw), when given, receive a second copy of the new velocity.
// SLAB says that the arrays hold only some planes of the grid.
template <typename T, typename OFF, bool SLAB>
__global__ void __launch_bounds__(256)
k_advect(Grid g, const T* __restrict__ qu, const T* __restrict__ qv, const T* __restrict__ qw,
         const T* __restrict__ qd, const T* __restrict__ U, const T* __restrict__ V, const T* __restrict__ W,
         T* __restrict__ ou, T* __restrict__ ov, T* __restrict__ ow, T* __restrict__ od, T* __restrict__ cu,
         T* __restrict__ cv, T* __restrict__ cw, const uint8_t* __restrict__ flags, T t_step, int density_mode, int zb,
         int* __restrict__ oob_flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z + zb;
    if (x >= g.nx || y >= g.ny) return;
    const OFF sy = (OFF)g.nx * (OFF)sizeof(T), sz = (OFF)g.nx * (OFF)g.ny * (OFF)sizeof(T);
    const OFF e = ((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x;   // element index
    const OFF i = e * (OFF)sizeof(T);                                   // its byte offset
    const int gz = z + g.gz0;
    T nu = *at_byte(qu, i), nv = *at_byte(qv, i), nw = *at_byte(qw, i), nd = *at_byte(qd, i);
    const bool interior = x > 0 && y > 0 && gz > 0 && x < g.nx - 1 && y < g.ny - 1 && gz < g.gnz - 1;
    if (interior) {
        const uint32_t fl = flags[e];
        const T four = (T)4, two = (T)2;
        const bool fluid = fl & FL_SELF_FLUID;
        // the 13-point neighbourhood of the carrying velocity, by rows
        const T* Uc = at_byte(U, i); const T* Uym = at_byte(U, i - sy); const T* Uzm = at_byte(U, i - sz);
        const T* Vc = at_byte(V, i); const T* Vyp = at_byte(V, i + sy); const T* Vzm = at_byte(V, i - sz);
        const T* Vypzm = at_byte(V, i + sy - sz);
        const T* Wc = at_byte(W, i); const T* Wzp = at_byte(W, i + sz); const T* Wym = at_byte(W, i - sy);
        const T* Wymzp = at_byte(W, i - sy + sz);
        const T u_c = Uc[0], u_xp = Uc[1], u_ym = Uym[0], u_xp_ym = Uym[1], u_zm = Uzm[0], u_xp_zm = Uzm[1];
        const T v_c = Vc[0], v_xm = Vc[-1], v_yp = Vyp[0], v_xm_yp = Vyp[-1], v_zm = Vzm[0], v_yp_zm = Vypzm[0];
        const T w_c = Wc[0], w_xm = Wc[-1], w_zp = Wzp[0], w_xm_zp = Wzp[-1], w_ym = Wym[0], w_ym_zp = Wymzp[0];
        // face-collocated velocities: main.cpp:261-265, :269-273, :276-280
        const T uy = (v_c + v_xm + v_yp + v_xm_yp) / four, uz = (w_c + w_xm + w_zp + w_xm_zp) / four;
        const T vx = (u_c + u_ym + u_xp + u_xp_ym) / four, vz = (w_c + w_ym + w_zp + w_ym_zp) / four;
        const T wx = (u_c + u_zm + u_xp + u_xp_zm) / four, wy = (v_c + v_zm + v_yp + v_yp_zm) / four;
        const FastTrace<T, OFF> tu = fast_prepare<T, OFF, SLAB>(g, x, y, gz, u_c, uy, uz, t_step);
        const FastTrace<T, OFF> tv = fast_prepare<T, OFF, SLAB>(g, x, y, gz, vx, v_c, vz, t_step);
        const FastTrace<T, OFF> tw = fast_prepare<T, OFF, SLAB>(g, x, y, gz, wx, wy, w_c, t_step);
        bool oob = fluid && (tu.oob || tv.oob || tw.oob);
        const T su = fast_sample<T, OFF>(g, qu, tu, sy, sz), sv = fast_sample<T, OFF>(g, qv, tv, sy, sz),
                sw = fast_sample<T, OFF>(g, qw, tw, sy, sz);
        if (fluid) { nu = su; nv = sv; nw = sw; }   // main.cpp:260
        // main.cpp:284-289
        T uc, vc, wc;
        if (density_mode == 1) {
            uc = (nu + u_xp) / two; vc = (nv + v_yp) / two; wc = (nw + w_zp) / two;
        } else {
            uc = (u_c + u_xp) / two; vc = (v_c + v_yp) / two; wc = (w_c + w_zp) / two;
        }
        const FastTrace<T, OFF> td = fast_prepare<T, OFF, SLAB>(g, x, y, gz, uc, vc, wc, t_step);
        const T sd = fast_sample<T, OFF>(g, qd, td, sy, sz);
        if (!(fl & FL_SELF_OBJECT)) { nd = sd; oob = oob || td.oob; }
        if (SLAB && oob) *oob_flag = 1;
    }
    T* pu = const_cast<T*>(at_byte<T, OFF>(ou, i)); T* pv = const_cast<T*>(at_byte<T, OFF>(ov, i));
    T* pw = const_cast<T*>(at_byte<T, OFF>(ow, i)); T* pd = const_cast<T*>(at_byte<T, OFF>(od, i));
    *pu = nu; *pv = nv; *pw = nw; *pd = nd;
    if (cu) {
        *const_cast<T*>(at_byte<T, OFF>(cu, i)) = nu; *const_cast<T*>(at_byte<T, OFF>(cv, i)) = nv;
        *const_cast<T*>(at_byte<T, OFF>(cw, i)) = nw;
    }
}

// MacCormack correction and limiter, the scheme of 3_fluid/FluidSim.cpp:217-389 on the 3D sampler
// (not in the 3D reference).  q = the field before the forward advection, (U,V,W) the carrying
// velocity, f the forward result.  One location: b = f traced backwards over -t_step, m = f + (q - b)/2,
// and m falls back to f where it leaves the range of the eight taps of q the forward trace read (taps
// that read as the default count as 0).  Locations the advection leaves alone keep f.
template <typename T>
__device__ __forceinline__ T maccormack_one(const Grid& g, const T* __restrict__ q, const T* __restrict__ f, int x, int y,
                                            int gz, T dx, T dy, T dz, T t_step, T q_own, T f_own, bool& oob) {
    // (the backward sample in the short form of k_advect: weights of taps the reference does not read are zero)
    const FastTrace<T, int64_t> tb = fast_prepare<T, int64_t, true>(g, x, y, gz, dx, dy, dz, -t_step);
    const T b = fast_sample<T, int64_t>(g, f, tb, (int64_t)g.nx * (int64_t)sizeof(T), (int64_t)0);
    T m = f_own + (T)0.5 * (q_own - b);
    const FastTrace<T, int64_t> tf = fast_prepare<T, int64_t, true>(g, x, y, gz, dx, dy, dz, t_step);
    const Taps<T> a = fast_gather<T, int64_t>(q, trace_base(g, tf), (int64_t)g.nx * (int64_t)sizeof(T), tf.dz1);
    // taps in the order of main.cpp:400-407: 000, 100, 010, 001, 110, 101, 011, 111
    const bool ok[8] = {tf.vx0 && tf.vy0 && tf.vz0, tf.vx1 && tf.vy0 && tf.vz0, tf.vx0 && tf.vy1 && tf.vz0, tf.vx0 && tf.vy0 && tf.vz1,
                        tf.vx1 && tf.vy1 && tf.vz0, tf.vx1 && tf.vy0 && tf.vz1, tf.vx0 && tf.vy1 && tf.vz1, tf.vx1 && tf.vy1 && tf.vz1};
    T lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T s = (ok[k] && isfinite(a.s[k])) ? a.s[k] : (T)0;
        if (k == 0) { lo = s; hi = s; }
        lo = s < lo ? s : lo;
        hi = s > hi ? s : hi;
    }
    if (m < lo || m > hi) m = f_own;
    oob = oob || tb.oob || tf.oob;
    return m;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_maccormack(Grid g, const T* __restrict__ qu, const T* __restrict__ qv, const T* __restrict__ qw,
             const T* __restrict__ qd, const T* __restrict__ U, const T* __restrict__ V, const T* __restrict__ W,
             const T* __restrict__ fu, const T* __restrict__ fv, const T* __restrict__ fw, const T* __restrict__ fd,
             T* __restrict__ mu, T* __restrict__ mv, T* __restrict__ mw, T* __restrict__ md, T* __restrict__ cu,
             T* __restrict__ cv, T* __restrict__ cw, const uint8_t* __restrict__ flags, T t_step, int zb,
             int* __restrict__ oob_flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z + zb;
    if (x >= g.nx || y >= g.ny) return;
    const int64_t i = gidx(g, x, y, z);
    const int gz = z + g.gz0;
    T nu = fu[i], nv = fv[i], nw = fw[i], nd = fd[i];
    const bool interior = x > 0 && y > 0 && gz > 0 && x < g.nx - 1 && y < g.ny - 1 && gz < g.gnz - 1;
    if (interior) {
        const uint32_t fl = flags[i];
        const T four = (T)4, two = (T)2;
        bool oob = false;
        const T u_c = U[i], u_xp = U[i + 1], v_c = V[i], v_yp = V[i + g.sy], w_c = W[i], w_zp = W[i + g.sz];
        if (fl & FL_SELF_FLUID) {
            const T u_ym = U[i - g.sy], u_xp_ym = U[i + 1 - g.sy], u_zm = U[i - g.sz], u_xp_zm = U[i + 1 - g.sz];
            const T v_xm = V[i - 1], v_xm_yp = V[i - 1 + g.sy], v_zm = V[i - g.sz], v_yp_zm = V[i + g.sy - g.sz];
            const T w_xm = W[i - 1], w_xm_zp = W[i - 1 + g.sz], w_ym = W[i - g.sy], w_ym_zp = W[i - g.sy + g.sz];
            // face-collocated velocities: main.cpp:261-265, :269-273, :276-280
            nu = maccormack_one(g, qu, fu, x, y, gz, u_c, (v_c + v_xm + v_yp + v_xm_yp) / four,
                                (w_c + w_xm + w_zp + w_xm_zp) / four, t_step, qu[i], nu, oob);
            nv = maccormack_one(g, qv, fv, x, y, gz, (u_c + u_ym + u_xp + u_xp_ym) / four, v_c,
                                (w_c + w_ym + w_zp + w_ym_zp) / four, t_step, qv[i], nv, oob);
            nw = maccormack_one(g, qw, fw, x, y, gz, (u_c + u_zm + u_xp + u_xp_zm) / four,
                                (v_c + v_zm + v_yp + v_yp_zm) / four, w_c, t_step, qw[i], nw, oob);
        }
        if (!(fl & FL_SELF_OBJECT))
            nd = maccormack_one(g, qd, fd, x, y, gz, (u_c + u_xp) / two, (v_c + v_yp) / two, (w_c + w_zp) / two, t_step,
                                qd[i], nd, oob);
        if (oob) *oob_flag = 1;
    }
    // (non-temporal stores: nothing reads the results before the launch ends, and the caches are busy with the taps)
    __builtin_nontemporal_store(nu, mu + i); __builtin_nontemporal_store(nv, mv + i);
    __builtin_nontemporal_store(nw, mw + i); __builtin_nontemporal_store(nd, md + i);
    if (cu) { __builtin_nontemporal_store(nu, cu + i); __builtin_nontemporal_store(nv, cv + i); __builtin_nontemporal_store(nw, cw + i); }
}

// Reflection about the projected field (Zehnder et al. 2018, Alg. 1):
// out = 2*proj - tilde for the three velocity components.
template <typename T>
__global__ void __launch_bounds__(256)
k_reflect(int64_t n, const T* __restrict__ pu, const T* __restrict__ pv, const T* __restrict__ pw,
          const T* __restrict__ tu, const T* __restrict__ tv, const T* __restrict__ tw, T* __restrict__ ou,
          T* __restrict__ ov, T* __restrict__ ow) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        ou[i] = (T)2 * pu[i] - tu[i];
        ov[i] = (T)2 * pv[i] - tv[i];
        ow[i] = (T)2 * pw[i] - tw[i];
    }
}

// Buoyancy on low-y faces between two Fluid cells, the 3D form of
// 3_fluid/FluidSim.h:109-118 and :136-148: v += dt * (coef*(d[y-1]+d[y])/2*(64/nx)).
template <typename T>
__global__ void __launch_bounds__(256)
k_buoyancy(Grid g, T* __restrict__ v, const T* __restrict__ d, const uint8_t* __restrict__ flags, T dt, T coef,
           int zb, int ze) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    const int z = blockIdx.z + zb;
    if (x > g.nx - 2 || y > g.ny - 2) return;
    const int64_t i = gidx(g, x, y, z);
    const uint32_t fl = flags[i];
    if (!(fl & FL_SELF_FLUID) || !(fl & FL_YM)) return;
    const T scaling = (T)64 / (T)g.nx;
    const T f = coef * (d[i - g.sy] + d[i]) / (T)2 * scaling;
    v[i] += dt * f;
}

// Sparse stamps used by the emitters: field[idx[k]] = val[k]; cells[idx[k]] = type.
template <typename T>
__global__ void k_scatter(int n, const int64_t* __restrict__ idx, const float* __restrict__ val, float scale,
                          T* __restrict__ field) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) field[idx[k]] = (T)(scale * val[k]);
}
static __global__ void k_scatter_cells(int n, const int64_t* __restrict__ idx, uint8_t type, uint8_t* __restrict__ cells) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) cells[idx[k]] = type;
}

}  // namespace fsamd
