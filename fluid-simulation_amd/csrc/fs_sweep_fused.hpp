// One full red-black sweep (colour 0 then colour 1) in ONE pass over HBM.
//
// Two half-sweep launches read and write every face twice (50 B per cell per
// sweep).  Here a workgroup marches a TX x TY column of cells along z with a
// window of six planes of u, v, w and the flag bytes in LDS, applies the red
// half-sweep to plane t and the black half-sweep to plane t-2 in the same
// step, and writes a plane out once both colours have passed it: every face
// is read once and written once (25 B per cell per sweep).  The result is
// written to a second set of arrays, so workgroups only ever read the old
// generation and need no ordering among themselves.
//
// Why this is the same arithmetic as two separate half-sweeps: a cell visit
// (advection-reflection/main.cpp:198-233) reads and writes only the cell's
// own six faces; black on plane p needs red finished on planes p-1, p, p+1,
// which the two-plane lag guarantees; the visits inside one colour are order
// independent.  The tile is loaded with a two-cell apron in x and y: after
// the red pass the outer ring of the window is stale, after the black pass
// the second ring, and the TX x TY core that is stored is exact.
//
// LDS rows are stored de-interleaved (even x first, then odd x): the active
// cells of a row all have the same x parity, so every access of a wave is to
// consecutive words and free of bank conflicts.
#pragma once
#include "fs_kernels3d.hpp"

namespace fsamd {

template <typename T, int TX_, int TY_>
struct FusedTile {
    static constexpr int TX = TX_, TY = TY_;
    static constexpr int RX = TX + 4, RY = TY + 4;   // window with a two-cell apron
    static constexpr int HX = RX / 2;
    static constexpr int PLANE = RX * RY;
    static constexpr int NP = 6;                     // planes in flight
    static constexpr int NLOAD = (PLANE + 255) / 256;
    static constexpr int NSTORE = (TX * TY + 255) / 256;
    static constexpr int NWORK = (RY - 1) * HX;      // candidate cells of one colour on one plane
    static constexpr size_t lds_bytes() { return (size_t)NP * PLANE * (3 * sizeof(T) + 1); }
    static_assert(RX % 2 == 0, "window width must be even");
};

struct FusedRanges {
    int s0b, s0e;   // local planes the first colour visits
    int s1b, s1e;   // local planes the second colour visits
    int z0, z1;     // local planes this launch stores (the other planes of the output are left alone)
    int zchunk;     // distance along z between the first planes of consecutive workgroups
    int zlen;       // planes stored per workgroup (= zchunk, except where one launch serves two separate
                    // ranges: the planes next to a slab's lower and upper interface)
};

template <typename T, int TX, int TY>
__global__ void __launch_bounds__(256)
k_rb_fused(Grid g, const T* __restrict__ ui, const T* __restrict__ vi, const T* __restrict__ wi,
           T* __restrict__ uo, T* __restrict__ vo, T* __restrict__ wo, const uint8_t* __restrict__ flags, T omega,
           FusedRanges r) {
    using C = FusedTile<T, TX, TY>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* su = reinterpret_cast<T*>(lds_raw);
    T* sv = su + C::NP * C::PLANE;
    T* sw = sv + C::NP * C::PLANE;
    uint8_t* sf = reinterpret_cast<uint8_t*>(sw + C::NP * C::PLANE);

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int c0 = r.z0 + blockIdx.z * r.zchunk, c1 = min(c0 + r.zlen, r.z1);   // planes this workgroup stores
    // a stored plane needs the second colour on itself and the plane below, and the first colour one
    // plane beyond that on either side
    const int s1b = max(r.s1b, c0 - 1), s1e = min(r.s1e, c1);
    const int s0b = max(r.s0b, c0 - 2), s0e = min(r.s0e, c1 + 1);
    const int lb = max(0, c0 - 2), le = min(g.nzl, c1 + 2);                  // planes loaded

    auto slot = [](int p) { return (p + 2 * C::NP) % C::NP; };
    auto pos = [](int rx, int ry) { return ry * C::RX + (rx & 1) * C::HX + (rx >> 1); };

    T ru[C::NLOAD], rv[C::NLOAD], rw[C::NLOAD];
    uint32_t rf[C::NLOAD];

    auto fetch = [&](int p) {   // global -> registers
#pragma unroll
        for (int k = 0; k < C::NLOAD; ++k) {
            const int e = tid + k * 256;
            const int ry = e / C::RX, rx = e - ry * C::RX;
            const int x = x0 - 2 + rx, y = y0 - 2 + ry;
            const bool in = e < C::PLANE && x >= 0 && y >= 0 && x < g.nx && y < g.ny;
            const int64_t i = in ? gidx(g, x, y, p) : 0;
            ru[k] = in ? ui[i] : (T)0;
            rv[k] = in ? vi[i] : (T)0;
            rw[k] = in ? wi[i] : (T)0;
            rf[k] = in ? (uint32_t)flags[i] : (uint32_t)FL_SELF_OBJECT;
        }
    };
    auto stage_in = [&](int p) {   // registers -> LDS
        const int sp = slot(p) * C::PLANE;
#pragma unroll
        for (int k = 0; k < C::NLOAD; ++k) {
            const int e = tid + k * 256;
            if (e < C::PLANE) {
                const int ry = e / C::RX, rx = e - ry * C::RX;
                const int a = sp + pos(rx, ry);
                su[a] = ru[k]; sv[a] = rv[k]; sw[a] = rw[k]; sf[a] = (uint8_t)rf[k];
            }
        }
    };
    auto relax_plane = [&](int p, int colour) {   // one colour on one plane, in place in LDS
        const int sp = slot(p) * C::PLANE, sp1 = slot(p + 1) * C::PLANE;
        for (int idx = tid; idx < C::NWORK; idx += 256) {
            const int ry = idx / C::HX, k = idx - ry * C::HX;
            const int y = y0 - 2 + ry;
            const int rx = 2 * k + ((colour + x0 + y + p + g.gz0) & 1);
            if (rx > C::RX - 2) continue;
            const int a = pos(rx, ry);
            const uint32_t fl = sf[sp + a];
            if (!relaxes(fl)) continue;
            const int ax = pos(rx + 1, ry), ay = pos(rx, ry + 1);
            const T u0 = su[sp + a], u1 = su[sp + ax];
            const T v0 = sv[sp + a], v1 = sv[sp + ay];
            const T w0 = sw[sp + a], w1 = sw[sp1 + a];
            const T div = (u1 - u0) + (v1 - v0) + (w1 - w0);            // main.cpp:217-219
            T dd = div / (T)__popc(fl & FL_NBR);                        // main.cpp:225
            if (omega != (T)1) dd = omega * dd;
            const bool self = fl & FL_SELF_FLUID;
            if ((fl & FL_XM) && self) su[sp + a] = u0 + dd;             // main.cpp:227-232
            if (fl & FL_XP) su[sp + ax] = u1 - dd;
            if ((fl & FL_YM) && self) sv[sp + a] = v0 + dd;
            if (fl & FL_YP) sv[sp + ay] = v1 - dd;
            if ((fl & FL_ZM) && self) sw[sp + a] = w0 + dd;
            if (fl & FL_ZP) sw[sp1 + a] = w1 - dd;
        }
    };
    auto store_out = [&](int p) {   // LDS core -> global (new generation)
        const int sp = slot(p) * C::PLANE;
#pragma unroll
        for (int k = 0; k < C::NSTORE; ++k) {
            const int e = tid + k * 256;
            const int cy = e / TX, cx = e - cy * TX;
            const int x = x0 + cx, y = y0 + cy;
            if (e < TX * TY && x < g.nx && y < g.ny) {
                const int a = sp + pos(cx + 2, cy + 2);
                const int64_t i = gidx(g, x, y, p);
                uo[i] = su[a]; vo[i] = sv[a]; wo[i] = sw[a];
            }
        }
    };

    // plane p: fetched at step p-3, staged into LDS at p-2, first colour at p, second colour at p+2,
    // stored at p+3; its LDS slot is reused at p+4
    if (lb < le) fetch(lb);
    for (int t = lb - 2; t <= c1 + 2; ++t) {
        const int pl = t + 2;
        if (pl >= lb && pl < le) stage_in(pl);
        if (pl + 1 >= lb && pl + 1 < le) fetch(pl + 1);
        if (t >= s0b && t < s0e) relax_plane(t, 0);
        if (t - 2 >= s1b && t - 2 < s1e) relax_plane(t - 2, 1);
        if (t - 3 >= c0 && t - 3 < c1) store_out(t - 3);
        __syncthreads();
    }
}

}  // namespace fsamd
