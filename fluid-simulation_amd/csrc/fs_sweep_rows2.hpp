// TWO full red-black sweeps in one pass over HBM: the register-resident one-pass sweep
// (fs_sweep_rows.hpp) carried one level further.
//
// The one-pass sweep is bound by memory (its vector ALUs are about a third busy), and a projection that
// needs K sweeps reads and writes the velocity K times.  Here a workgroup applies, per step t of its march
// along z, the first colour of sweep 1 to plane t, the second colour of sweep 1 to plane t-1, the first
// colour of sweep 2 to plane t-2 and the second colour of sweep 2 to plane t-3, and stores plane t-4: every
// face is read once and written once per TWO sweeps (12.5 B per cell and sweep).
//
// Why the arithmetic is that of four separate half-sweeps: a cell visit (advection-reflection/
// main.cpp:198-233) touches only the cell's own six faces, and a half-sweep of one colour needs the
// previous half-sweep finished on the six neighbours of every cell it visits.  The neighbours in the
// same plane and in the plane below were visited in earlier steps (there is a barrier between steps);
// the neighbour above is the cell directly above in the same column, which the same lane has just
// visited in the same step (w never leaves the lane).  The four phases of a step work on four different
// planes, so they do not meet in LDS either.
//
// Each half-sweep lets one more ring of a tile go stale (its cells lack the visits of cells outside the
// tile): tiles overlap by eight cells in x and y, z-chunks by four planes either side, and the stored
// core is exact.  Output goes to the other generation of the arrays.  A launch stores the planes [zs0, zs1) of
// the arrays it is given and reads four planes more on either side: the whole grid of a single domain, or
// the owned planes of a z-slab whose arrays carry four current halo planes on either side.
#pragma once
#include <type_traits>
#include "fs_kernels3d.hpp"
#include "fs_sweep_rows.hpp"

namespace fsamd {

// LPR lanes of a wave share a row, so a wave holds 64 / LPR rows and a workgroup of 16 waves a tile of
// (LPR * 2 * CP * NSEG) columns x (16 * 64 / LPR) rows: the same number of cells whatever LPR is, but a squarer
// tile loses a smaller share of them to the stale rings.  The rows of a wave lie two apart, so they have one colour
// and the wave takes one branch of a visit (the number of waves is even).
template <typename T, int NSEG, int CP, int LPR = 64>
struct Rows2Cfg {
    static constexpr int WAVES = 16;        // waves of a workgroup
    static constexpr int RPW = 64 / LPR;    // rows of a wave
    static constexpr int RY = WAVES * RPW;  // rows of a tile
    static constexpr int SEGW = 2 * LPR * CP;
    static constexpr int W = SEGW * NSEG;
    static constexpr int NQ = NSEG * CP;
    static constexpr int NPV = 8;           // v planes in LDS: one per phase of a step and the plane being stored make five; eight, so that
                                            // the slot of a plane is a mask and not a division (scalar instructions are not free here)
    static constexpr int S = 4;             // rings of a tile that go stale (one per half-sweep)
    static constexpr size_t v_bytes() { return (size_t)NPV * WAVES * 2 * NQ * 64 * sizeof(T); }
    static constexpr size_t lds_bytes() { return v_bytes() + kRcpBytes; }   // + the reciprocals of visit_quotient()
};

template <typename T, int NSEG, int CP, int LPR>
__global__ void __launch_bounds__(1024, 4)
k_rb_rows2(Grid g, const T* __restrict__ ui, const T* __restrict__ vi, const T* __restrict__ wi,
           T* __restrict__ uo, T* __restrict__ vo, T* __restrict__ wo, const uint8_t* __restrict__ vflags, T omega,
           int zb, int ze, int zs0, int zs1, int zchunk, int zlen, int ntx, int nty, int rows, int xbase, int xedges) {
    using C = Rows2Cfg<T, NSEG, CP, LPR>;
    constexpr int NQ = C::NQ, S = C::S;
    using VecT = Vec<T, 2 * CP>;
    using FlagWord = typename std::conditional<CP == 1, uint16_t, uint32_t>::type;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* sv = reinterpret_cast<T*>(lds_raw);   // [NPV][waves][2 * NQ][64]: a wave's lanes side by side, whatever rows they hold

    double* rcp = reinterpret_cast<double*>(lds_raw + C::v_bytes());
    fill_reciprocals(rcp, threadIdx.y * 64 + threadIdx.x, blockDim.y * 64);   // (the barrier of the first step comes before its first use)

    const int lane = threadIdx.x, ry = threadIdx.y, waves = blockDim.y;
    const int lx = lane & (LPR - 1), sub = lane / LPR;
    // row of the tile this lane works on: a pair of waves shares 2 * RPW adjacent rows, the even wave takes the
    // even ones
    const int r = (ry >> 1) * (2 * C::RPW) + 2 * sub + (ry & 1);
    // workgroups sharing an XCD are y-neighbours of one z-chunk (see k_rb_rows)
    const int total = gridDim.x, lin = blockIdx.x;
    const int xcd = lin & 7, qq = total >> 3, rem = total & 7;
    const int logical = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (lin >> 3);
    const int by = logical % nty, bx = (logical / nty) % ntx, bz = logical / (nty * ntx);
    // A launch may cover only some of the tiles of a row (the rest being left to a launch of another tile width):
    // xbase is the origin of its first tile, xedges says whether its first (bit 0) and last (bit 1) tiles touch
    // the domain boundary.
    const int xr0 = xbase + bx * (C::W - 2 * S), yr0 = by * (rows - 2 * S);
    const int y = yr0 + r;
    const bool row_in = y < g.ny;
    const int cx_lo = (bx == 0 && (xedges & 1)) ? 0 : xr0 + S, cx_hi = (bx == ntx - 1 && (xedges & 2)) ? g.nx : xr0 + C::W - S;
    const int cy_lo = by == 0 ? 0 : yr0 + S, cy_hi = by == nty - 1 ? g.ny : yr0 + rows - S;
    const bool row_core = y >= cy_lo && y < cy_hi;

    // planes [c0, c1) are stored.  The last half-sweep visits [c0 - 1, c1) (the low-z face of plane c0 is the
    // high-z face of the cells below it), every earlier one a plane more on either side, all inside the planes
    // the sweep relaxes, [zb, ze)
    // (a workgroup's planes: zlen of them every zchunk planes - zlen = zchunk, except where one launch serves the planes
    // next to a slab's lower and upper interface)
    const int c0 = zs0 + bz * zchunk, c1 = min(c0 + zlen, zs1);
    const int r4b = max(zb, c0 - 1), r4e = min(ze, c1);
    const int r3b = max(zb, c0 - 2), r3e = min(ze, c1 + 1);
    const int r2b = max(zb, c0 - 3), r2e = min(ze, c1 + 2);
    const int r1b = max(zb, c0 - 4), r1e = min(ze, c1 + 3);
    const int lb = max(0, c0 - 4), le = min(g.nzl, c1 + 4);

    // Rows (waves) a half-sweep has to visit.  The k-th half-sweep is wrong on row k-2 from a tile edge that is not
    // a domain edge (it reads a face the half-sweep before left stale there) and nobody needs its rows further out;
    // the last row of a tile never visits (the v of the row above it is not held).
    const int lo2 = by == 0 ? 0 : 1, lo3 = by == 0 ? 0 : 2, lo4 = by == 0 ? 0 : 3;
    const bool top = by == nty - 1;
    const int hi1 = rows - 1, hi2 = top ? rows - 1 : rows - 2, hi3 = top ? rows - 1 : rows - 3, hi4 = top ? rows - 1 : rows - 4;

    const int64_t rowoff = (int64_t)y * g.sy + xr0 + 2 * CP * lx;
    bool seg_in[NSEG], pair_core[NQ];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int x = xr0 + sg * C::SEGW + 2 * CP * lx;
        seg_in[sg] = row_in && x < g.nx;
#pragma unroll
        for (int j = 0; j < CP; ++j) pair_core[sg * CP + j] = row_core && x + 2 * j >= cx_lo && x + 2 * j < cx_hi;
    }

    // cells of this lane that are ever visited: not those of the last row of the tile (the v of the row above is not
    // held) nor the last column of the tile (the u of the column after it is not held)
    uint32_t keep[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        keep[sg] = r < rows - 1 ? 0xffffffffu : 0u;
        if (sg == NSEG - 1 && lx == LPR - 1) keep[sg] &= CP == 1 ? 0x00ffu : 0x00ffffffu;
    }

    RowPlane<T, NQ, NSEG> A, B, Cc, D, E, F;   // planes t-3, t-2, t-1, t, t+1 (u, flags; w since a step earlier), t+2 (w)
    VecT vst[NSEG];

    auto load_uf = [&](RowPlane<T, NQ, NSEG>& X, int p) {
        const bool pin = p >= lb && p < le;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const bool in = pin && seg_in[sg];
            const int64_t i = in ? (int64_t)p * g.sz + rowoff + sg * C::SEGW : 0;
            VecT pu, pv;
            uint32_t fw = 0;   // columns outside the grid are not visited
#pragma unroll
            for (int k = 0; k < 2 * CP; ++k) { pu.e[k] = 0; pv.e[k] = 0; }
            if (in) {
                pu = *reinterpret_cast<const VecT*>(ui + i);
                pv = *reinterpret_cast<const VecT*>(vi + i);
                fw = *reinterpret_cast<const FlagWord*>(vflags + i);
            }
            X.f[sg] = fw & keep[sg];
#pragma unroll
            for (int j = 0; j < CP; ++j) { X.u[sg * CP + j][0] = pu.e[2 * j]; X.u[sg * CP + j][1] = pu.e[2 * j + 1]; }
            vst[sg] = pv;
        }
    };
    auto load_w = [&](RowPlane<T, NQ, NSEG>& X, int p) {
        const bool pin = p >= lb && p < le;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const bool in = pin && seg_in[sg];
            const int64_t i = in ? (int64_t)p * g.sz + rowoff + sg * C::SEGW : 0;
            VecT pw;
#pragma unroll
            for (int k = 0; k < 2 * CP; ++k) pw.e[k] = 0;
            if (in) pw = *reinterpret_cast<const VecT*>(wi + i);
#pragma unroll
            for (int j = 0; j < CP; ++j) { X.w[sg * CP + j][0] = pw.e[2 * j]; X.w[sg * CP + j][1] = pw.e[2 * j + 1]; }
        }
    };
    auto vrow = [&](int p, int wv) { return sv + ((size_t)((unsigned)(p + 2 * C::NPV) % (unsigned)C::NPV) * waves + wv) * (2 * NQ * 64); };
    auto vpos = [&](int q, int c) { return (2 * q + c) * 64 + lane; };
    auto stage_v = [&](int p) {
        T* row = vrow(p, ry);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
            for (int j = 0; j < CP; ++j) {
                row[vpos(sg * CP + j, 0)] = vst[sg].e[2 * j];
                row[vpos(sg * CP + j, 1)] = vst[sg].e[2 * j + 1];
            }
    };
    // One colour on plane p (X = plane p, Y = plane p+1): the visit of k_rb_rows.
    auto relax = [&](RowPlane<T, NQ, NSEG>& X, RowPlane<T, NQ, NSEG>& Y, int p, int colour) {
        const int c = (colour + xr0 + y + p + g.gz0) & 1;
        T* v0row = vrow(p, ry);
        // the row above: the same lanes of the next wave (from an even wave); from an odd wave the next row of the
        // wave before, LPR lanes further on, or, from its last row, the first row of the wave after
        const bool row_ok = r < rows - 1;
        T* v1row = !row_ok ? v0row
                   : (ry & 1) == 0 ? vrow(p, ry + 1)
                   : sub < C::RPW - 1 ? vrow(p, ry - 1) + LPR
                                      : vrow(p, ry + 1) - (C::RPW - 1) * LPR;
        T send[NQ];
        auto visit = [&](auto colour_of_pair) {
            constexpr int cc = decltype(colour_of_pair)::value;
            static_for<NQ>([&](auto column_pair) {
                constexpr int q = decltype(column_pair)::value, sg = q / CP, j = q % CP, sh = 16 * j + 8 * cc;
                const uint32_t vw = X.f[sg];
                const T u0 = X.u[q][cc];
                T u1;
                if constexpr (cc == 0) u1 = X.u[q][1];
                else if constexpr (j + 1 < CP) u1 = X.u[q + 1 < NQ ? q + 1 : q][0];
                else {
                    u1 = __shfl_down(X.u[sg * CP][0], 1, 64);   // (the last column of a tile gets a foreign value: it is never updated)
                    if constexpr (sg + 1 < NSEG) {
                        const T wrap = __shfl(X.u[(sg + 1 < NSEG ? sg + 1 : sg) * CP][0], lane & ~(LPR - 1), 64);
                        if (lx == LPR - 1) u1 = wrap;
                    }
                }
                const int a = vpos(q, cc);
                const T v0 = v0row[a], v1 = v1row[a];
                const T w0 = X.w[q][cc], w1 = Y.w[q][cc];
                const T div = (u1 - u0) + (v1 - v0) + (w1 - w0);
                const T dd = omega * visit_quotient(div, vw >> sh, rcp);   // (x * 1 is x: omega == 1 needs no test)
                if constexpr (sizeof(T) == 4) {
                    v0row[a] = pick(bit_mask<sh + 2>(vw), v0 + dd, v0);
                    v1row[a] = pick(bit_mask<sh + 3>(vw), v1 - dd, v1);
                    X.u[q][cc] = pick(bit_mask<sh>(vw), u0 + dd, u0);
                    if constexpr (cc == 0) X.u[q][1] = pick(bit_mask<sh + 1>(vw), u1 - dd, u1);
                    else send[q] = __uint_as_float(__float_as_uint(dd) & bit_mask<sh + 1>(vw));
                    X.w[q][cc] = pick(bit_mask<sh + 4>(vw), w0 + dd, w0);
                    Y.w[q][cc] = pick(bit_mask<sh + 5>(vw), w1 - dd, w1);
                } else {
                    const uint32_t vb = vw >> sh;
                    if (vb & FL_XM) X.u[q][cc] = u0 + dd;
                    if constexpr (cc == 0) { if (vb & FL_XP) X.u[q][1] = u1 - dd; }
                    else send[q] = (vb & FL_XP) ? dd : (T)0;
                    if (vb & FL_YM) v0row[a] = v0 + dd;
                    if (vb & FL_YP) v1row[a] = v1 - dd;
                    if (vb & FL_ZM) X.w[q][cc] = w0 + dd;
                    if (vb & FL_ZP) Y.w[q][cc] = w1 - dd;
                }
            });
        };
        if (c == 0) visit(std::integral_constant<int, 0>{});
        else {
            visit(std::integral_constant<int, 1>{});
            // what the second colour's visits took from the u face they share with the column after them
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int sg = q / CP, j = q % CP;
                T got;
                bool any = true;
                if (j > 0) {
                    got = send[q > 0 ? q - 1 : 0];
                } else {
                    got = __shfl_up(send[sg * CP + CP - 1], 1, 64);
                    any = lx > 0;
                    if (sg > 0) {
                        const T wrap = __shfl(send[(sg > 0 ? sg : 1) * CP - 1], lane | (LPR - 1), 64);
                        if (lx == 0) { got = wrap; any = true; }
                    }
                }
                if (any && got != (T)0) X.u[q][0] = X.u[q][0] - got;
            }
        }
    };
    auto store_plane = [&](RowPlane<T, NQ, NSEG>& X, int p) {
        const T* row = vrow(p, ry);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const int64_t i = (int64_t)p * g.sz + rowoff + sg * C::SEGW;
            bool all = true, none = true;
#pragma unroll
            for (int j = 0; j < CP; ++j) { all = all && pair_core[sg * CP + j]; none = none && !pair_core[sg * CP + j]; }
            if (none) continue;
            VecT pu, pv, pw;
#pragma unroll
            for (int j = 0; j < CP; ++j) {
                const int q = sg * CP + j;
                pu.e[2 * j] = X.u[q][0]; pu.e[2 * j + 1] = X.u[q][1];
                pv.e[2 * j] = row[vpos(q, 0)]; pv.e[2 * j + 1] = row[vpos(q, 1)];
                pw.e[2 * j] = X.w[q][0]; pw.e[2 * j + 1] = X.w[q][1];
            }
            if (all) {
                stream_store(uo + i, pu);
                stream_store(vo + i, pv);
                stream_store(wo + i, pw);
            } else {
#pragma unroll
                for (int j = 0; j < CP; ++j)
                    if (pair_core[sg * CP + j]) {
                        using Pr = Vec<T, 2>;
                        *reinterpret_cast<Pr*>(uo + i + 2 * j) = Pr{{pu.e[2 * j], pu.e[2 * j + 1]}};
                        *reinterpret_cast<Pr*>(vo + i + 2 * j) = Pr{{pv.e[2 * j], pv.e[2 * j + 1]}};
                        *reinterpret_cast<Pr*>(wo + i + 2 * j) = Pr{{pw.e[2 * j], pw.e[2 * j + 1]}};
                    }
            }
        }
    };

    // Step t: Z = plane t-4, A = t-3, B = t-2, Cc = t-1, D = t, E = t+1, F = t+2.  u, flags and v of a plane are
    // fetched one step before its first visit, w two steps before (it is the high-z face of the plane below).  The
    // planes move down one name per step (register moves: the vector ALUs have the time); what a step fetches is
    // moved at the start of the NEXT step, so that a fetch has a whole step to arrive.  A plane is stored one step
    // after its last visit: the barrier that opens a step is then the only one (it also separates that visit, which
    // other waves made to this wave's v row, from the store).
    RowPlane<T, NQ, NSEG> Z;
    load_uf(Z, -1); load_uf(A, -1); load_uf(B, -1); load_uf(Cc, -1); load_uf(D, -1); load_uf(E, -1); load_uf(F, -1);
    load_w(Z, -1); load_w(A, -1); load_w(B, -1); load_w(Cc, -1); load_w(D, -1); load_w(E, -1);
    load_w(F, lb);   // the step before the first one "fetched" the w of plane lb
    for (int t = lb - 1; t <= c1 + 3; ++t) {
        if (t >= lb && t < le) stage_v(t);            // the v fetched with plane t in the step before
        D = E;                                         // plane t: u and flags fetched in the step before, w the step before that
#pragma unroll
        for (int q = 0; q < NQ; ++q) { E.w[q][0] = F.w[q][0]; E.w[q][1] = F.w[q][1]; }
        load_uf(E, t + 1);
        load_w(F, t + 2);
        __syncthreads();
        if (t - 4 >= c0 && t - 4 < c1) store_plane(Z, t - 4);
        // (the row conditions are per lane; the lanes of one row, which shuffle with each other, agree on them)
        if (t >= r1b && t < r1e && r < hi1) relax(D, E, t, 0);
        if (t - 1 >= r2b && t - 1 < r2e && r >= lo2 && r < hi2) relax(Cc, D, t - 1, 1);
        if (t - 2 >= r3b && t - 2 < r3e && r >= lo3 && r < hi3) relax(B, Cc, t - 2, 0);
        if (t - 3 >= r4b && t - 3 < r4e && r >= lo4 && r < hi4) relax(A, B, t - 3, 1);
        Z = A; A = B; B = Cc; Cc = D;
    }
}

// The kernel as hipLaunchKernel takes it.  Defined in fs_rows2.hip, the translation unit that instantiates the
// two-sweep kernels: this is the one kernel that runs faster (3-4 %) with the compiler's SLP vectorizer on, which
// the rest of the library is built without.
template <typename T, int NSEG, int CP, int LPR>
const void* rows2_kernel();

}  // namespace fsamd
