// One red-black sweep folded into the advection that follows it.
//
// Where a projection is a single sweep (advection-reflection/main.cpp:105-111 leaves the loop after one pass of
// gauss_seidel_projection(), :189-254) the step is bound by the bytes its launches move: the sweep reads and
// writes the velocity (25 B per cell), and semi_lagrange() (:256-291) reads it again (33 B).  Here one launch
// does both: a workgroup marches a tile of 64 x 16 cells along z with a ring of eight planes of u, v, w and
// the flag bytes in LDS, each plane widened by three cells either way.  The planes arrive raw; in step t the
// workgroup applies the first colour of the sweep to plane t and the second colour to plane t-2 IN the ring (a
// cell visit touches only its own six faces, :217-232, so within one colour the visits are independent, and the
// second colour of plane t-2 finds the first colour finished on planes t-3, t-2 and t-1 by the steps before),
// and advects plane t-4 from the planes t-5, t-4, t-3, which the steps before made final, exactly as
// k_advect_tiled does from its staged planes.  The three jobs of a step touch different planes, so a step has
// one barrier and a lane's visits overlap its advection.  The density's taps come straight from memory (eight
// loads per cell, mostly the lines its neighbours load: a ring for it would not fit beside eight planes of the
// velocity).  The swept velocity is never written to memory: the launch moves 17 B per cell in and 16 B out.
//
// Why the result is that of a sweep launch followed by an advection launch, bit for bit: the visits are those
// of k_rb_halfsweep (same expression, IEEE division); a visit near the edge of the window lacks the visits of
// cells outside the window, which makes the outer two rings of the window stale after the two colours - the
// advection reads the window less those two rings (its tile and a one-cell apron), and the planes before the
// first and after the last plane of a z-chunk are treated the same way.  The advection step is the in-window
// step of k_advect_tiled.  A cell whose sample leaves the window (|v| dt of a cell or more) or is not finite
// cannot be served here - the general path of k_advect_tiled reads the swept field from memory, where this
// launch never puts it - so such a cell raises *bad_flag, and the caller drops the result and runs the sweep
// and the advection as two launches.
//
// 4-byte elements only (the rings of 8-byte elements do not fit the 160 KiB of LDS); single domain.
#pragma once
#include "fs_advect_tiled.hpp"

namespace fsamd {

struct SweepAdvTile {
    static constexpr int TX = 64, TY = 16, R = 3, NPL = 8, NF = 3;
    static constexpr int WX = TX + 2 * R, WY = TY + 2 * R, PS = WX * WY;
    static constexpr int NT = TX * TY;
    static constexpr int HALF = WX / 2;                    // cells of one colour in a window row
    static constexpr int NVISIT = HALF * WY;               // threads that visit a cell in a half-sweep
    static constexpr size_t field_bytes() { return (size_t)NF * NPL * PS * sizeof(float); }
    static constexpr size_t lds_bytes() { return field_bytes() + (size_t)NPL * PS; }   // + the flag bytes
};

template <typename OFF>
__global__ void __launch_bounds__(1024, 4)
k_sweep_advect(Grid g, const float* __restrict__ ui, const float* __restrict__ vi, const float* __restrict__ wi,
               const float* __restrict__ qd, float* __restrict__ ou, float* __restrict__ ov, float* __restrict__ ow,
               float* __restrict__ od, const uint8_t* __restrict__ flags, float omega, float t_step, int density_mode,
               int zb, int ze, int zchunk, int* __restrict__ bad_flag, double* __restrict__ res_part) {
    using T = float;
    using C = SweepAdvTile;
    constexpr int R = C::R, WX = C::WX, PS = C::PS, NPL = C::NPL;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* L = reinterpret_cast<T*>(lds_raw);                  // [field][ring slot][WY][WX]
    uint8_t* LF = lds_raw + C::field_bytes();              // [ring slot][WY][WX]

    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * C::TX + tx;
    // tiles numbered so that an XCD works on a run of neighbouring tiles (see k_advect_tiled)
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total = gridDim.x * gridDim.y * gridDim.z;
    const int xcd = lin & 7, qq = total >> 3, rem = total & 7;
    const int tile = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (lin >> 3);
    const int bx = tile % gridDim.x, by = (tile / gridDim.x) % gridDim.y, bz = tile / (gridDim.x * gridDim.y);
    const int x0 = bx * C::TX, y0 = by * C::TY;
    const int c0 = zb + bz * zchunk, c1 = min(c0 + zchunk, ze);   // planes this workgroup advects
    const OFF sz = (OFF)g.nx * (OFF)g.ny;                         // elements of a plane
    const T* src[3] = {ui, vi, wi};

    // the window elements this thread stages (the same for every plane); elements outside the grid come from a
    // clamped address: their flag byte is then that of a shell cell, which is never visited, and they are never a
    // tap of a sample
    static_assert((C::PS + C::NT - 1) / C::NT == 2, "a thread stages two elements per plane and field");
    OFF soff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = min(tid + k * C::NT, PS - 1);
        const int ry = e / WX, rx = e - ry * WX;
        const int gx = min(max(x0 - R + rx, 0), g.nx - 1), gy = min(max(y0 - R + ry, 0), g.ny - 1);
        soff[k] = (OFF)gy * (OFF)g.nx + (OFF)gx;
    }
    const bool second = tid + C::NT < PS;
    T stage[3][2];
    uint8_t fstage[2];
    auto fetch = [&](int p) {    // global -> registers
        const OFF po = (OFF)min(max(p, 0), g.nzl - 1) * sz;
#pragma unroll
        for (int f = 0; f < 3; ++f) stage[f][0] = src[f][po + soff[0]];
        fstage[0] = flags[po + soff[0]];
        if (second) {
#pragma unroll
            for (int f = 0; f < 3; ++f) stage[f][1] = src[f][po + soff[1]];
            fstage[1] = flags[po + soff[1]];
        }
    };
    auto slot = [&](int p) { return p & (NPL - 1); };
    auto commit = [&](int p) {   // registers -> LDS
        const int sl = slot(p);
        T* dst = L + sl * PS + tid;
#pragma unroll
        for (int f = 0; f < 3; ++f) dst[f * NPL * PS] = stage[f][0];
        LF[sl * PS + tid] = fstage[0];
        if (second) {
#pragma unroll
            for (int f = 0; f < 3; ++f) dst[f * NPL * PS + C::NT] = stage[f][1];
            LF[sl * PS + tid + C::NT] = fstage[1];
        }
    };
    auto plane = [&](int f, int p) -> T* { return L + (f * NPL + slot(p)) * PS; };

    // ---- the sweep: one colour on plane p, in the ring -----------------------------------------------------------
    // thread -> the cell it visits: row vry of the window, the vk-th cell of the colour in that row
    const int vry = tid / C::HALF, vk = tid - vry * C::HALF;
    const bool visits = tid < C::NVISIT && vry < C::WY - 1;   // (the last row of the window lacks the v above it)
    auto relax = [&](int p, int colour) {
        if (!visits) return;
        // (x + y + z) & 1 == colour, in global coordinates (k_rb_halfsweep)
        const int wx = 2 * vk + ((colour + x0 - R + y0 - R + vry + p + g.gz0) & 1);
        if (wx >= WX - 1) return;                              // (the last column lacks the u after it)
        const int c = vry * WX + wx;
        const uint32_t fl = LF[slot(p) * PS + c];
        if (!relaxes(fl)) return;
        T* U = plane(0, p) + c;
        T* V = plane(1, p) + c;
        T* W0 = plane(2, p) + c;
        T* W1 = plane(2, p + 1) + c;
        // main.cpp:217-225
        const T div = (U[1] - U[0]) + (V[WX] - V[0]) + (W1[0] - W0[0]);
        T dd = div / (T)__popc(fl & FL_NBR);
        if (omega != (T)1) dd = omega * dd;
        const bool self = fl & FL_SELF_FLUID;
        if ((fl & FL_XM) && self) U[0] += dd;                  // main.cpp:227-232
        if (fl & FL_XP) U[1] -= dd;
        if ((fl & FL_YM) && self) V[0] += dd;
        if (fl & FL_YP) V[WX] -= dd;
        if ((fl & FL_ZM) && self) W0[0] += dd;
        if (fl & FL_ZP) W1[0] -= dd;
    };

    // ---- the advection of the own cell (the in-window step of k_advect_tiled) ------------------------------------
    const int x = x0 + tx, y = y0 + ty;
    const bool mine = x < g.nx && y < g.ny;
    const int wl = (ty + R) * WX + tx + R;   // own cell inside the window
    const bool safe_x = x0 >= 2 && x0 + C::TX <= g.nx - 2, safe_y = y0 >= 2 && y0 + C::TY <= g.ny - 2;
    const T xf = (T)x, yf = (T)y, tv = t_step;
    constexpr uint32_t ESZ = sizeof(T), FS = NPL * PS * ESZ;
    const uint32_t wl4 = (uint32_t)wl * ESZ;
    const T lim = max(max(g.nx, g.ny), g.gnz) <= 8192 ? (T)0.998046875 : (T)0;
    auto sign_mask = [](T v) -> uint32_t { return (uint32_t)(__float_as_int(v) >> 31); };
    const uint32_t edge = (x == 1 ? 1u : 0u) | (x == g.nx - 2 ? 2u : 0u) | (y == 1 ? 4u : 0u) | (y == g.ny - 2 ? 8u : 0u);
    auto bit_mask = [](uint32_t v, int b) -> uint32_t { return (uint32_t)((int32_t)(v << (31 - b)) >> 31); };
    auto keep_unless = [](T w, uint32_t kill) -> T { return __uint_as_float(__float_as_uint(w) & ~kill); };

    // A sample inside the window: weights, the signs of floor(p) - own coordinate per axis (each -1 or 0) and whether the
    // backtrace is short enough for them to be all there is (see k_advect_tiled)
    struct Prep { FastTrace<T, OFF> tr; uint32_t mx, my, mz; };
    auto prepare_sample = [&](auto safe_tag, T dx, T dy, T dz, T zf, uint32_t zedge, bool& bad) -> Prep {
        using S = decltype(safe_tag);
        const T one = (T)1;
        const T dxt = dx * tv, dyt = dy * tv, dzt = dz * tv;
        const T px = xf - dxt, py = yf - dyt, pz = zf - dzt;
        const T bx_ = floor(px), by_ = floor(py), bz_ = floor(pz);
        const T fu = px - bx_, fv = py - by_, fw = pz - bz_;
        Prep q;
        q.tr.x0 = one - fu; q.tr.x1 = fu; q.tr.y0 = one - fv; q.tr.y1 = fv; q.tr.z0 = one - fw; q.tr.z1 = fw;
        const bool in = fmax(fmax(fabs(dxt), fabs(dyt)), fabs(dzt)) < lim;   // (false for a NaN)
        q.mx = sign_mask(bx_ - xf); q.my = sign_mask(by_ - yf); q.mz = sign_mask(bz_ - zf);
        if constexpr (!S::x) { q.tr.x0 = keep_unless(q.tr.x0, bit_mask(edge, 0) & q.mx); q.tr.x1 = keep_unless(q.tr.x1, bit_mask(edge, 1) & ~q.mx); }
        if constexpr (!S::y) { q.tr.y0 = keep_unless(q.tr.y0, bit_mask(edge, 2) & q.my); q.tr.y1 = keep_unless(q.tr.y1, bit_mask(edge, 3) & ~q.my); }
        if constexpr (!S::z) { q.tr.z0 = keep_unless(q.tr.z0, (0u - (zedge & 1u)) & q.mz); q.tr.z1 = keep_unless(q.tr.z1, (0u - (zedge >> 1)) & ~q.mz); }
        bad = bad || !in;
        return q;
    };
    // the taps of a velocity component from the ring
    auto lds_sample = [&](auto safe_tag, int f, T dx, T dy, T dz, T zf, uint32_t zedge, uint32_t sB, uint32_t sC, uint32_t dAB,
                          uint32_t dBC, bool& bad) -> T {
        const Prep q = prepare_sample(safe_tag, dx, dy, dz, zf, zedge, bad);
        const uint32_t a4 = wl4 + (q.mx & (0u - ESZ)) + (q.my & (0u - (uint32_t)WX * ESZ));
        uint32_t kB = sB + (uint32_t)f * FS, kC = sC + (uint32_t)f * FS;
        asm volatile("" : "+s"(kB), "+s"(kC));
        const uint32_t o0 = a4 + (q.mz & dAB) + kB;
        const uint32_t o1 = a4 + (q.mz & dBC) + kC;
        const T* p0 = reinterpret_cast<const T*>(lds_raw + o0);
        const T* p1 = reinterpret_cast<const T*>(lds_raw + o1);
        Taps<T> t;
        t.s[0] = p0[0]; t.s[1] = p0[1]; t.s[2] = p0[WX]; t.s[3] = p1[0];
        t.s[4] = p0[WX + 1]; t.s[5] = p1[1]; t.s[6] = p1[WX]; t.s[7] = p1[WX + 1];
        T r = fast_combine(q.tr, t);
        asm volatile("" : "+v"(r));
        return r;
    };
    // the taps of the density from memory: own element `own`, the lower tap one element, row and plane before it where
    // the mask of the axis says so (all of them inside the grid for an interior cell)
    auto mem_sample = [&](auto safe_tag, OFF own, T dx, T dy, T dz, T zf, uint32_t zedge, bool& bad) -> T {
        const Prep q = prepare_sample(safe_tag, dx, dy, dz, zf, zedge, bad);
        const OFF row = (OFF)g.nx;
        const OFF base = own - (OFF)(q.mx & 1u) - ((OFF)(q.my & 1u)) * row - ((OFF)(q.mz & 1u)) * sz;
        const T* p0 = qd + base;
        const T* p1 = p0 + sz;
        Taps<T> t;
        t.s[0] = p0[0]; t.s[1] = p0[1]; t.s[2] = p0[row]; t.s[3] = p1[0];
        t.s[4] = p0[row + 1]; t.s[5] = p1[1]; t.s[6] = p1[row]; t.s[7] = p1[row + 1];
        T r = fast_combine(q.tr, t);
        asm volatile("" : "+v"(r));
        return r;
    };

    double racc = 0;
    auto compute = [&](auto safe_tag, int z, uint32_t fl, T (&res)[4]) -> bool {
        const int gz = z + g.gz0;
        const OFF own = ((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x;
        T nu = plane(0, z)[wl], nv = plane(1, z)[wl], nw = plane(2, z)[wl], nd = qd[own];
        const bool interior = x > 0 && y > 0 && gz > 0 && x < g.nx - 1 && y < g.ny - 1 && gz < g.gnz - 1;
        bool bad = false;
        if (interior) {
            const T zf = (T)gz;
            const uint32_t sA = (uint32_t)slot(z - 1) * PS * ESZ, sB = (uint32_t)slot(z) * PS * ESZ, sC = (uint32_t)slot(z + 1) * PS * ESZ;
            const uint32_t dAB = sA - sB, dBC = sB - sC;
            const uint32_t zedge = (gz == 1 ? 1u : 0u) | (gz == g.gnz - 2 ? 2u : 0u);
            const T four = (T)4, two = (T)2;
            const bool fluid = fl & FL_SELF_FLUID;
            const T* Uc = plane(0, z) + wl; const T* Uzm = plane(0, z - 1) + wl;
            const T* Vc = plane(1, z) + wl; const T* Vzm = plane(1, z - 1) + wl;
            const T* Wc = plane(2, z) + wl; const T* Wzp = plane(2, z + 1) + wl;
            const T u_c = Uc[0], u_xp = Uc[1], u_ym = Uc[-WX], u_xp_ym = Uc[1 - WX], u_zm = Uzm[0], u_xp_zm = Uzm[1];
            const T v_c = Vc[0], v_xm = Vc[-1], v_yp = Vc[WX], v_xm_yp = Vc[WX - 1], v_zm = Vzm[0], v_yp_zm = Vzm[WX];
            const T w_c = Wc[0], w_xm = Wc[-1], w_zp = Wzp[0], w_xm_zp = Wzp[-1], w_ym = Wc[-WX], w_ym_zp = Wzp[-WX];
            if (res_part) {   // the cells and the value of k_div_sumsq, on the swept field
                const uint32_t movable = (fl & FL_SELF_FLUID) ? (fl & FL_NBR) : (fl & (FL_XP | FL_YP | FL_ZP));
                if (relaxes(fl) && movable) {
                    const double dv = (double)((u_xp - u_c) + (v_yp - v_c) + (w_zp - w_c));
                    racc += dv * dv;
                }
            }
            // face-collocated velocities: main.cpp:261-265, :269-273, :276-280
            const T uy = (v_c + v_xm + v_yp + v_xm_yp) / four, uz = (w_c + w_xm + w_zp + w_xm_zp) / four;
            const T vx = (u_c + u_ym + u_xp + u_xp_ym) / four, vz = (w_c + w_ym + w_zp + w_ym_zp) / four;
            const T wx_ = (u_c + u_zm + u_xp + u_xp_zm) / four, wy = (v_c + v_zm + v_yp + v_yp_zm) / four;
            auto smp = [&](int f, T dx, T dy, T dz) { return lds_sample(safe_tag, f, dx, dy, dz, zf, zedge, sB, sC, dAB, dBC, bad); };
            const T su = smp(0, u_c, uy, uz), sv = smp(1, vx, v_c, vz), sw = smp(2, wx_, wy, w_c);
            if (fluid) { nu = su; nv = sv; nw = sw; }   // main.cpp:260
            // main.cpp:284-289
            T uc, vc, wc;
            if (density_mode == 1) {
                uc = (nu + u_xp) / two; vc = (nv + v_yp) / two; wc = (nw + w_zp) / two;
            } else {
                uc = (u_c + u_xp) / two; vc = (v_c + v_yp) / two; wc = (w_c + w_zp) / two;
            }
            const T sd = mem_sample(safe_tag, own, uc, vc, wc, zf, zedge, bad);
            if (!(fl & FL_SELF_OBJECT)) nd = sd;
            bad = bad || !isfinite((su + sv) + (sw + sd));
        }
        res[0] = nu; res[1] = nv; res[2] = nw; res[3] = nd;
        return bad;
    };
    auto store = [&](int z, const T (&res)[4]) {
        const OFF i = ((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x;
        __builtin_nontemporal_store(res[0], ou + i); __builtin_nontemporal_store(res[1], ov + i);
        __builtin_nontemporal_store(res[2], ow + i); __builtin_nontemporal_store(res[3], od + i);
    };

    // Step t: commit plane t+2 (fetched in the step before), fetch plane t+3, first colour on plane t, second colour on
    // plane t-2, advect plane t-4.  The half-sweeps visit the planes the advected ones need and the interior of the
    // grid only: the second colour [c0-2, c1], the first [c0-3, c1+1].  What a step writes - ring slot t+2 (= t-6), the
    // faces of planes t and t+1 (w), t-2 and t-1 (w) - and what it reads beyond that - planes t-5 to t-3 - are
    // different planes, so the one barrier at its top is all the ordering it needs.
    const int zlo = 1, zhi = g.nzl - 1;                     // interior planes [zlo, zhi) of a single domain
    const int t0 = c0 - 3;
    fetch(t0); commit(t0);
    fetch(t0 + 1); commit(t0 + 1);
    fetch(t0 + 2);
    T res[4];
    bool have = false;
    int zres = 0;
    for (int t = t0; t <= c1 + 3; ++t) {
        __syncthreads();
        commit(t + 2);
        asm volatile("" ::: "memory");
        if (have) { store(zres, res); have = false; }      // (after the commit: its wait for the staged loads does not wait for these stores)
        fetch(t + 3);
        if (t >= zlo && t < zhi && t <= c1 + 1) relax(t, 0);
        if (t - 2 >= zlo && t - 2 < zhi && t - 2 >= c0 - 2 && t - 2 <= c1) relax(t - 2, 1);
        const int z = t - 4;
        if (z >= c0 && z < c1 && mine) {
            const uint32_t fl = LF[slot(z) * PS + wl];
            const int gz = z + g.gz0;
            const bool safe_z = gz >= 2 && gz <= g.gnz - 3;
            bool bad;
            if (safe_x && safe_y && safe_z) bad = compute(AdvSafe<true, true, true>{}, z, fl, res);
            else if (safe_y && safe_z) bad = compute(AdvSafe<false, true, true>{}, z, fl, res);
            else if (safe_x && safe_z) bad = compute(AdvSafe<true, false, true>{}, z, fl, res);
            else if (safe_x && safe_y) bad = compute(AdvSafe<true, true, false>{}, z, fl, res);
            else bad = compute(AdvSafe<false, false, false>{}, z, fl, res);
            if (__builtin_expect(bad, 0)) *bad_flag = 1;
            have = true; zres = z;
        }
    }
    if (have) store(zres, res);
    if (res_part) {   // one partial per workgroup, added in a fixed order
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds_raw);
        racc = wave_sum(racc);
        if ((tid & 63) == 0) red[tid >> 6] = racc;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < C::NT / 64; ++k) t += red[k];
            res_part[2 * tile] = t;
            res_part[2 * tile + 1] = 0;
        }
    }
}

}  // namespace fsamd
