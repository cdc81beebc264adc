// One full red-black sweep in one pass over HBM, register-resident variant
// for even nx (the arrays are then aligned per column pair).
//
// A workgroup is up to 16 waves; wave r owns row y0 + r of a tile that is
// W = 128*CP*NSEG columns wide.  A lane owns CP adjacent column pairs in each of
// the NSEG segments of the row: columns sg*128*CP + 2*CP*lane + 2*j + c
// (pair j, column c of the pair), so that one load or store of a lane moves
// 2*CP elements (16 bytes: CP = 2 for 4-byte elements when nx is a multiple
// of 4, CP = 1 for 8-byte elements).  The wave marches along z.  For its
// columns a lane keeps u and w of the four planes in flight in registers: the
// two cells a w face touches lie in the same column, so w never leaves the
// lane, and the u faces a cell shares with its x neighbours live in the next
// pair of the lane or in the neighbouring lane and move by wave shuffles.
// Only v, which couples rows and therefore waves, is staged in LDS (two
// planes).  Every face is read once and written once per sweep (25 B per cell
// instead of the 50 B of two half-sweep launches).
//
// Per step t the workgroup applies colour 0 to plane t and then colour 1 to
// plane t-1; a cell visit (advection-reflection/main.cpp:198-233) touches only
// the cell's own six faces and black on plane p needs red finished on p-1, p,
// p+1, which the one-plane lag provides (see the step below), so the result is
// that of two separate half-sweeps exactly.
// Tiles overlap their neighbours by two cells in x and y (and chunks by two
// planes in z): the outer two rings of a tile go stale, the stored core is
// exact.  Output goes to the other generation of the arrays, so workgroups
// only read old data and need no ordering among themselves.
#pragma once
#include <type_traits>
#include "fs_kernels3d.hpp"
#include "fs_sweep_fused.hpp"

namespace fsamd {

template <typename T, int NQ, int NSEG>
struct RowPlane {            // what a lane holds of one z-plane: NQ = NSEG*CP column pairs
    T u[NQ][2];
    T w[NQ][2];
    uint32_t f[NSEG];        // visit bytes (k_build_flags) of the lane's 2*CP columns of a segment, column k in bits 8k..8k+7
};

// ---- pieces of a cell visit shared by the register-resident sweep kernels -----------------------------------
// All ones where bit BIT of w is set: one signed bit-field extract.  (Written as shifts the compiler turns it back
// into a test, a compare and a conditional move, which is what this is here to avoid.)
template <int BIT>
__device__ __forceinline__ uint32_t bit_mask(uint32_t w) {
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(BIT));
    return r;
}
// m ? a : b, bit by bit: one bit-field insert
__device__ __forceinline__ float pick(uint32_t m, float a, float b) {
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// 1 / divisor for each of the 256 visit bytes, in double: the table of visit_quotient(), filled by the workgroup.
constexpr size_t kRcpBytes = 256 * sizeof(double);
__device__ __forceinline__ void fill_reciprocals(double* rcp, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) rcp[i] = 1.0 / (double)max(1, visit_divisor((uint32_t)i));
}
// div / (number of Fluid neighbours), the division of main.cpp:222, correctly rounded.  In single precision it is
// formed as float(double(div) * rcp): the double product is within 2^-52 of the quotient, and a quotient of a float by
// 1..6 is either a float itself or at least 2^-27 (relative) away from the nearest point where rounding to float
// changes, so the rounding to float sees the same side as the exact quotient does; zeros, infinities, NaNs and
// subnormal results come out as IEEE division gives them (all 2^32 dividends x 6 divisors compared on the host).
// Five instructions instead of the fifteen of the division sequence.  vb = the cell's visit byte (0: the cell is not
// relaxed and the result is unused).
template <typename T>
__device__ __forceinline__ T visit_quotient(T div, uint32_t vb, const double* rcp) {
    if constexpr (sizeof(T) == 4) return (T)((double)div * rcp[vb & 0xffu]);
    else return div / (T)max(1, visit_divisor(vb & 0xffu));
}

template <typename T, int NSEG, int CP>
struct RowsCfg {
    static constexpr int RY = 16;           // most rows (waves) a workgroup can have
    static constexpr int SEGW = 128 * CP;   // columns of one segment
    static constexpr int W = SEGW * NSEG;
    static constexpr int NQ = NSEG * CP;
    static constexpr int NPV = 2;           // v planes in LDS: the planes of the two colours of a step
    static constexpr size_t v_bytes() { return (size_t)NPV * RY * W * sizeof(T); }
    static constexpr size_t lds_bytes() { return v_bytes() + kRcpBytes; }
};

// Variants whose lanes hold at most 8 bytes of a row per plane fit 64 registers - two workgroups per CU - with the
// compare-and-move form of the visit and the division instruction sequence (LEAN); the wider ones have 128 registers
// anyway and take the shorter select form.
template <typename T, int NSEG, int CP>
__global__ void __launch_bounds__(1024, (NSEG * CP * sizeof(T) <= 8 ? 8 : 4))
k_rb_rows(Grid g, const T* __restrict__ ui, const T* __restrict__ vi, const T* __restrict__ wi,
          T* __restrict__ uo, T* __restrict__ vo, T* __restrict__ wo, const uint8_t* __restrict__ flags,
          const uint8_t* __restrict__ vflags, T omega, FusedRanges r, int ntx, int nty, int rows, int xbase, int xedges) {
    using C = RowsCfg<T, NSEG, CP>;
    constexpr int NQ = C::NQ;
    constexpr bool LEAN = NSEG * CP * sizeof(T) <= 8;
    using VecT = Vec<T, 2 * CP>;
    using FlagWord = typename std::conditional<CP == 1, uint16_t, uint32_t>::type;   // the 2*CP flag bytes of a lane
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // [NPV][rows][W]; a row holds, per pair index q, the even columns of all lanes, then the odd ones:
    // the active cells of a row all have the same x parity, so a wave reads consecutive words
    T* sv = reinterpret_cast<T*>(lds_raw);
    double* rcp = reinterpret_cast<double*>(lds_raw + C::v_bytes());
    if constexpr (!LEAN) fill_reciprocals(rcp, threadIdx.y * 64 + threadIdx.x, blockDim.y * 64);   // (the barrier of the first step comes before its first use)

    const int lane = threadIdx.x, ry = threadIdx.y;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Renumber them so that the
    // workgroups sharing an XCD are y-neighbours of one z-chunk: the two halo rows a tile shares with each
    // neighbour are then fetched from HBM once and served from that L2 the second time.  (Placement is a
    // speed matter only; any mapping gives the same result.)
    const int total = gridDim.x, lin = blockIdx.x;
    const int xcd = lin & 7, qq = total >> 3, rem = total & 7;
    const int logical = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (lin >> 3);
    const int by = logical % nty, bx = (logical / nty) % ntx, bz = logical / (nty * ntx);
    // tile origin; tiles overlap by 4.  A launch may cover only some of the tiles of a row (the others being
    // left to a launch of another tile width): xbase is the origin of its first tile, and xedges says whether
    // its first (bit 0) and last (bit 1) tiles touch the domain boundary.
    const int xr0 = xbase + bx * (C::W - 4), yr0 = by * (rows - 4);
    const int y = yr0 + ry;
    const bool row_in = y < g.ny;
    // the part of the tile this workgroup stores: everything but the two stale rings, except at the
    // domain boundary where there is no neighbour tile
    const int cx_lo = (bx == 0 && (xedges & 1)) ? 0 : xr0 + 2, cx_hi = (bx == ntx - 1 && (xedges & 2)) ? g.nx : xr0 + C::W - 2;
    const int cy_lo = by == 0 ? 0 : yr0 + 2, cy_hi = by == nty - 1 ? g.ny : yr0 + rows - 2;
    const bool row_core = y >= cy_lo && y < cy_hi;

    const int c0 = r.z0 + bz * r.zchunk, c1 = min(c0 + r.zlen, r.z1);
    const int s1b = max(r.s1b, c0 - 1), s1e = min(r.s1e, c1);
    const int s0b = max(r.s0b, c0 - 2), s0e = min(r.s0e, c1 + 1);
    const int lb = max(0, c0 - 2), le = min(g.nzl, c1 + 2);

    const int64_t rowoff = (int64_t)y * g.sy + xr0 + 2 * CP * lane;
    bool seg_in[NSEG], pair_core[NQ];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const int x = xr0 + sg * C::SEGW + 2 * CP * lane;
        seg_in[sg] = row_in && x < g.nx;   // nx is a multiple of 2*CP: a lane's columns are in or out as a whole
#pragma unroll
        for (int j = 0; j < CP; ++j)       // the bounds are even: no pair straddles them
            pair_core[sg * CP + j] = row_core && x + 2 * j >= cx_lo && x + 2 * j < cx_hi;
    }

    // cells of this lane that are ever visited: not those of the last row of the tile (the v of the row above is not
    // held) nor the last column of the tile (the u of the column after it is not held)
    uint32_t keep[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        keep[sg] = ry < rows - 1 ? 0xffffffffu : 0u;
        if (sg == NSEG - 1 && lane == 63) keep[sg] &= CP == 1 ? 0x00ffu : 0x00ffffffu;
    }

    RowPlane<T, NQ, NSEG> P0, P1, P2, P3;
    VecT vst[NSEG];

    auto load_uf = [&](RowPlane<T, NQ, NSEG>& X, int p) {   // u, flags and the v staging registers of plane p
        const bool pin = p >= lb && p < le;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const bool in = pin && seg_in[sg];
            const int64_t i = in ? (int64_t)p * g.sz + rowoff + sg * C::SEGW : 0;
            VecT pu, pv;
            // columns outside the grid are not visited: FL_SELF_OBJECT in every flag byte, or visit bytes of zero
            uint32_t fw = LEAN ? (CP == 1 ? 0x8080u : 0x80808080u) : 0u;
            static_assert(FL_SELF_OBJECT == 0x80u, "filler for columns outside the grid");
#pragma unroll
            for (int k = 0; k < 2 * CP; ++k) { pu.e[k] = 0; pv.e[k] = 0; }
            if (in) {
                pu = *reinterpret_cast<const VecT*>(ui + i);
                pv = *reinterpret_cast<const VecT*>(vi + i);
                fw = *reinterpret_cast<const FlagWord*>((LEAN ? flags : vflags) + i);
            }
            X.f[sg] = LEAN ? fw : fw & keep[sg];
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
    auto vrow = [&](int p, int row) { return sv + ((size_t)((p + 3 * C::NPV) % C::NPV) * rows + row) * C::W; };
    auto vpos = [&](int q, int c) { return (2 * q + c) * 64 + lane; };   // place of (pair q, column c) in a row
    auto stage_v = [&](int p) {   // v staging registers -> LDS, own row
        T* row = vrow(p, ry);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
            for (int j = 0; j < CP; ++j) {
                row[vpos(sg * CP + j, 0)] = vst[sg].e[2 * j];
                row[vpos(sg * CP + j, 1)] = vst[sg].e[2 * j + 1];
            }
    };
    // One colour on plane p.  X holds plane p, Y plane p+1 (for the high-z faces).  A cell's visit byte says which
    // of its faces move; every face is rewritten, with its old value where the byte says so: in single precision a
    // bit-field extract and a bitwise select per face, no tests, compares or conditional moves.
    auto relax = [&](RowPlane<T, NQ, NSEG>& X, RowPlane<T, NQ, NSEG>& Y, int p, int colour) {
        const int c = (colour + xr0 + y + p + g.gz0) & 1;   // the active column of every pair of this row
        T* v0row = vrow(p, ry);
        T* v1row = vrow(p, ry < rows - 1 ? ry + 1 : ry);
        T send[NQ];
        auto visit = [&](auto colour_of_pair) {
            constexpr int cc = decltype(colour_of_pair)::value;
            static_for<NQ>([&](auto column_pair) {
                constexpr int q = decltype(column_pair)::value, sg = q / CP, j = q % CP, sh = 16 * j + 8 * cc;
                const uint32_t vw = X.f[sg];
                const T u0 = X.u[q][cc];
                // the high-x face of the pair's second column is the low-x face of the next pair: the lane's own
                // next pair, or the first pair of the segment in the next lane (the last column of a tile gets a
                // foreign value: it is never updated)
                T u1;
                if constexpr (cc == 0) u1 = X.u[q][1];
                else if constexpr (j + 1 < CP) u1 = X.u[q + 1 < NQ ? q + 1 : q][0];
                else {
                    u1 = __shfl_down(X.u[sg * CP][0], 1, 64);
                    if constexpr (sg + 1 < NSEG) {
                        const T wrap = __shfl(X.u[(sg + 1 < NSEG ? sg + 1 : sg) * CP][0], 0, 64);
                        if (lane == 63) u1 = wrap;
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
#pragma unroll
            for (int q = 0; q < NQ; ++q) {   // deliver -dd to the low-x face of the pair on the right
                const int sg = q / CP, j = q % CP;
                T got;
                bool any = true;
                if (j > 0) {
                    got = send[q > 0 ? q - 1 : 0];
                } else {
                    got = __shfl_up(send[sg * CP + CP - 1], 1, 64);
                    any = lane > 0;
                    if (sg > 0) {
                        const T wrap = __shfl(send[(sg > 0 ? sg : 1) * CP - 1], 63, 64);
                        if (lane == 0) { got = wrap; any = true; }
                    }
                }
                if (any && got != (T)0) X.u[q][0] = X.u[q][0] - got;
            }
        }
    };
    // The same visit on the flag bytes themselves, as compares and conditional moves with the division instruction
    // sequence: the form that fits 64 registers (LEAN variants).
    auto relax_lean = [&](RowPlane<T, NQ, NSEG>& X, RowPlane<T, NQ, NSEG>& Y, int p, int colour) {
        const int c = (colour + xr0 + y + p + g.gz0) & 1;   // the active column of every pair of this row
        T* v0row = vrow(p, ry);
        T* v1row = vrow(p, ry < rows - 1 ? ry + 1 : ry);
        const bool row_ok = ry < rows - 1;
        T send[NQ];
        if (c == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t fl = (X.f[q / CP] >> (16 * (q % CP))) & 0xffu;
                const bool ok = row_ok && relaxes(fl);
                const int a = vpos(q, 0);
                const T u0 = X.u[q][0], u1 = X.u[q][1];
                const T v0 = v0row[a], v1 = v1row[a];
                const T w0 = X.w[q][0], w1 = Y.w[q][0];
                const T div = (u1 - u0) + (v1 - v0) + (w1 - w0);
                T dd = div / (T)(ok ? __popc(fl & FL_NBR) : 1);
                if (omega != (T)1) dd = omega * dd;
                const bool self = fl & FL_SELF_FLUID;
                if (ok && (fl & FL_XM) && self) X.u[q][0] = u0 + dd;
                if (ok && (fl & FL_XP)) X.u[q][1] = u1 - dd;
                if (ok && (fl & FL_YM) && self) v0row[a] = v0 + dd;
                if (ok && (fl & FL_YP)) v1row[a] = v1 - dd;
                if (ok && (fl & FL_ZM) && self) X.w[q][0] = w0 + dd;
                if (ok && (fl & FL_ZP)) Y.w[q][0] = w1 - dd;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int sg = q / CP, j = q % CP;
                const uint32_t fl = (X.f[sg] >> (16 * j + 8)) & 0xffu;
                // the high-x face of the pair's second column is the low-x face of the next pair: the lane's
                // own next pair, or the first pair of the segment in the next lane
                T u1;
                bool have_hi = true;
                if (j + 1 < CP) {
                    u1 = X.u[q + 1 < NQ ? q + 1 : q][0];
                } else {
                    u1 = __shfl_down(X.u[sg * CP][0], 1, 64);
                    if (sg + 1 < NSEG) {
                        const T wrap = __shfl(X.u[(sg + 1 < NSEG ? sg + 1 : sg) * CP][0], 0, 64);
                        if (lane == 63) u1 = wrap;
                    } else {
                        have_hi = lane < 63;
                    }
                }
                const bool ok = row_ok && have_hi && relaxes(fl);
                const int a = vpos(q, 1);
                const T u0 = X.u[q][1];
                const T v0 = v0row[a], v1 = v1row[a];
                const T w0 = X.w[q][1], w1 = Y.w[q][1];
                const T div = (u1 - u0) + (v1 - v0) + (w1 - w0);
                T dd = div / (T)(ok ? __popc(fl & FL_NBR) : 1);
                if (omega != (T)1) dd = omega * dd;
                const bool self = fl & FL_SELF_FLUID;
                if (ok && (fl & FL_XM) && self) X.u[q][1] = u0 + dd;
                send[q] = (ok && (fl & FL_XP)) ? dd : (T)0;
                if (ok && (fl & FL_YM) && self) v0row[a] = v0 + dd;
                if (ok && (fl & FL_YP)) v1row[a] = v1 - dd;
                if (ok && (fl & FL_ZM) && self) X.w[q][1] = w0 + dd;
                if (ok && (fl & FL_ZP)) Y.w[q][1] = w1 - dd;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {   // deliver -dd to the low-x face of the pair on the right
                const int sg = q / CP, j = q % CP;
                T got;
                bool any = true;
                if (j > 0) {
                    got = send[q > 0 ? q - 1 : 0];
                } else {
                    got = __shfl_up(send[sg * CP + CP - 1], 1, 64);
                    any = lane > 0;
                    if (sg > 0) {
                        const T wrap = __shfl(send[(sg > 0 ? sg : 1) * CP - 1], 63, 64);
                        if (lane == 0) { got = wrap; any = true; }
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
            } else {   // the lane straddles the edge of the stored core: pair by pair
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

    // Step t: M = plane t-1, Cc = t, N = t+1, F = t+2.  The first colour runs on plane t and then, in the
    // same thread, the second colour on plane t-1.  A second-colour cell of plane t-1 needs the first colour
    // finished on its six neighbours: those in plane t-1 and the one in plane t-2 ran in earlier steps
    // (barriers since), and the one in plane t is the cell directly above it, in the same column, i.e. the
    // same lane a moment earlier (w never leaves the lane).  u/flags/v of a plane are fetched one step before
    // its first colour, w two steps before (it is the high-z face of the plane below).
#define FS_ROWS_STEP(M, Cc, N, F)                                               \
    {                                                                           \
        if (t >= lb && t < le) stage_v(t);                                      \
        load_uf(N, t + 1);                                                      \
        load_w(F, t + 2);                                                       \
        __syncthreads();                                                        \
        if (t >= s0b && t < s0e) {                                              \
            if constexpr (LEAN) relax_lean(Cc, N, t, 0); else relax(Cc, N, t, 0); \
        }                                                                       \
        if (t - 1 >= s1b && t - 1 < s1e) {                                      \
            if constexpr (LEAN) relax_lean(M, Cc, t - 1, 1); else relax(M, Cc, t - 1, 1); \
        }                                                                       \
        __syncthreads();                                                        \
        if (t - 1 >= c0 && t - 1 < c1) store_plane(M, t - 1);                   \
        ++t;                                                                    \
        if (t > tend) break;                                                    \
    }

    int t = lb - 1;
    const int tend = c1;
    load_w(P2, lb);   // at the first step (t = lb-1) plane lb is "N"
    // keep the compiler from treating never-loaded planes as undefined
    load_w(P0, -1); load_w(P1, -1); load_w(P3, -1);
    load_uf(P0, -1); load_uf(P1, -1); load_uf(P2, -1); load_uf(P3, -1);
    for (;;) {
        FS_ROWS_STEP(P0, P1, P2, P3)
        FS_ROWS_STEP(P1, P2, P3, P0)
        FS_ROWS_STEP(P2, P3, P0, P1)
        FS_ROWS_STEP(P3, P0, P1, P2)
    }
#undef FS_ROWS_STEP
}

}  // namespace fsamd
