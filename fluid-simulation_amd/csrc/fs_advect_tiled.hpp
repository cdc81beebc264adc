// Semi-Lagrangian advection with the fields staged through LDS.
//
// The direct kernel (k_advect) issues about 32 gather loads per cell and keeps
// the texture-address unit of every CU busy ~90 % of the time; it is bound by
// load issue, not by HBM.  Here a workgroup of 64 x 8 threads marches its
// column of cells along z with a ring of four planes of every field in LDS
// (the plane being computed, one plane either side, one plane in flight), each
// plane widened by a one-cell apron in x and y.  Every global element is then
// loaded once per workgroup with coalesced row loads (1.3 loads per cell and
// field), and the 13-point velocity neighbourhood and the eight taps of each
// of the four samples (advection-reflection/main.cpp:256-291, :380-435) are
// LDS reads.
//
// A backtrace that ends inside the window (|v| dt < 1 cell, the usual case) is
// served from LDS by a step without a branch; a cell with any other sample is
// computed again, all four samples, by the general path of the direct kernel
// (global loads), so the result does not depend on the window size.  The
// arithmetic is that of k_advect, bit for bit.
//
// What bounds it: the in-window step is about 250 vector instructions per wave
// and plane whose cost depends on their kind (see the note above lds_sample),
// beside a staging skeleton - loads, LDS writes, barrier, stores - that alone
// takes about as long; the two overlap.  Stores leave after the LDS commit so
// that the wait for the staged loads does not wait for them as well.
#pragma once
#include "fs_kernels3d.hpp"
#include <type_traits>

namespace fsamd {

template <bool SX, bool SY, bool SZ>
struct AdvSafe { static constexpr bool x = SX, y = SY, z = SZ; };

template <typename T, bool SEP>
struct AdvTile {
    static constexpr int TX = 64, TY = 8, R = 1, NPL = 4;   // NPL = 2R+2 planes, a power of two
    static constexpr int WX = TX + 2 * R, WY = TY + 2 * R, PS = WX * WY;
    static constexpr int NF = SEP ? 7 : 4;                  // u,v,w,d carried (+ the carrying velocity if it is another field)
    static constexpr int kMaxChunk = 128;                   // most planes a workgroup may be given (a bit per plane notes the cells to redo)
    static constexpr int NT = TX * TY;
    static constexpr int NLD = (PS + NT - 1) / NT;
    static constexpr size_t lds_bytes() { return (size_t)NF * NPL * PS * sizeof(T); }
};

// SEP: the carrying velocity (U,V,W) is not (qu,qv,qw).  SLAB: the arrays hold only some planes of the grid.
// REFL (needs SEP): the carried velocity is the reflection 2*(U,V,W) - (qu,qv,qw) of the given field about
// the carrying one (Zehnder et al. 2018, Alg. 1), formed while staging instead of by a pass of its own.
// res_part (optional): the carrying velocity is the field a projection has just left, and its six faces per
// cell are in registers here anyway; the kernel then also adds up the squared divergences that k_div_sumsq
// would (same cells, same per-cell value), one partial sum per workgroup, so that the projection's stopping
// rule can be verified without a pass of its own over the velocity.
// (second launch bound: waves per SIMD the register budget must allow - six for the four-field variant of
// 4-byte elements, three workgroups per CU; the seven-field variants are held to two workgroups by their LDS)
template <typename T, typename OFF, bool SEP, bool SLAB, bool REFL = false>
__global__ void __launch_bounds__(512, sizeof(T) == 4 ? (SEP ? 4 : 6) : 1)
k_advect_tiled(Grid g, const T* __restrict__ qu, const T* __restrict__ qv, const T* __restrict__ qw,
               const T* __restrict__ qd, const T* __restrict__ U, const T* __restrict__ V, const T* __restrict__ W,
               T* __restrict__ ou, T* __restrict__ ov, T* __restrict__ ow, T* __restrict__ od, T* __restrict__ cu,
               T* __restrict__ cv, T* __restrict__ cw, const uint8_t* __restrict__ flags, T t_step, int density_mode,
               int zb, int ze, int zchunk, int zlen, int* __restrict__ oob_flag, double* __restrict__ res_part) {
    using C = AdvTile<T, SEP>;
    constexpr int R = C::R, WX = C::WX, PS = C::PS, NPL = C::NPL;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* L = reinterpret_cast<T*>(lds_raw);   // [field][ring slot][WY][WX]

    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * C::TX + tx;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  They are renumbered so that an XCD
    // works on a run of consecutive tiles - whole rows of x-tiles, neighbouring rows, of one z-chunk: a staged row
    // is 66 elements that start one element before a 256-byte boundary, so its first and last 128-byte line are
    // lines the x-neighbour needs in full, and with the neighbour on the same XCD they come out of that L2 instead
    // of crossing the fabric a second time (the apron rows of the y-neighbours likewise).  Placement is a speed
    // matter only; the tile a workgroup works on decides everything else.
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total = gridDim.x * gridDim.y * gridDim.z;
    const int xcd = lin & 7, qq = total >> 3, rem = total & 7;
    const int tile = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (lin >> 3);
    const int bx = tile % gridDim.x, by = (tile / gridDim.x) % gridDim.y, bz = tile / (gridDim.x * gridDim.y);
    const int x0 = bx * C::TX, y0 = by * C::TY;
    // a workgroup's planes: zlen of them every zchunk planes (zlen = zchunk, except where one launch serves the
    // planes next to a slab's lower and upper interface)
    const int c0 = zb + bz * zchunk, c1 = min(c0 + zlen, ze);
    const OFF sy = (OFF)g.nx * (OFF)sizeof(T), sz = (OFF)g.nx * (OFF)g.ny * (OFF)sizeof(T);
    const T* src[7] = {qu, qv, qw, qd, U, V, W};

    // the window elements this thread stages (the same for every plane).  Elements outside the grid are
    // never the tap of a sample (tap addresses are clamped into the grid): they are loaded from a clamped,
    // valid address so that no load needs a branch.
    static_assert(C::NLD == 2, "a thread stages two elements per plane and field");
    OFF soff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = min(tid + k * C::NT, PS - 1);
        const int ry = e / WX, rx = e - ry * WX;
        const int gx = min(max(x0 - R + rx, 0), g.nx - 1), gy = min(max(y0 - R + ry, 0), g.ny - 1);
        soff[k] = ((OFF)gy * (OFF)g.nx + (OFF)gx) * (OFF)sizeof(T);
    }
    const bool second = tid + C::NT < PS;   // this thread has a second element
    T stage[C::NF][2];
    auto fetch = [&](int p) {    // global -> registers
        const OFF po = (OFF)min(max(p, 0), g.nzl - 1) * sz;
#pragma unroll
        for (int f = 0; f < C::NF; ++f) stage[f][0] = *at_byte(src[f], po + soff[0]);
        if (second) {
#pragma unroll
            for (int f = 0; f < C::NF; ++f) stage[f][1] = *at_byte(src[f], po + soff[1]);
        }
    };
    auto commit = [&](int p) {   // registers -> LDS
        if (REFL) {
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                stage[f][0] = (T)2 * stage[4 + f][0] - stage[f][0];
                stage[f][1] = (T)2 * stage[4 + f][1] - stage[f][1];
            }
        }
        T* dst = L + (p & (NPL - 1)) * PS + tid;
#pragma unroll
        for (int f = 0; f < C::NF; ++f) dst[f * NPL * PS] = stage[f][0];
        if (second) {
#pragma unroll
            for (int f = 0; f < C::NF; ++f) dst[f * NPL * PS + C::NT] = stage[f][1];
        }
    };
    auto plane = [&](int f, int p) -> const T* { return L + (f * NPL + (p & (NPL - 1))) * PS; };

    const int x = x0 + tx, y = y0 + ty;
    const bool mine = x < g.nx && y < g.ny;
    const int wl = (ty + R) * WX + tx + R;   // own cell inside the window
    constexpr int VF = SEP ? 4 : 0;          // first field of the carrying velocity

    // Tiles and planes at least two cells away from the shell along an axis: every tap of a backtrace that ends inside
    // the staged window is then one the reference reads, and the zeroing of weights is skipped for that axis.  All
    // three are uniform over the workgroup; the step is compiled per combination (AdvSafe), so that a safe axis
    // leaves no trace of the tests.
    const bool safe_x = x0 >= 2 && x0 + C::TX <= g.nx - 2, safe_y = y0 >= 2 && y0 + C::TY <= g.ny - 2;

    // ---- samples -----------------------------------------------------------------------------------------
    // The usual sample ends inside the staged window (|v| dt < 1 cell) and has a finite result.  The step is
    // written for that case without a branch: every sample of a cell reads its eight taps from LDS at an
    // address that lies inside the window whatever the lane holds, and notes in `bad` whether it was entitled
    // to; a cell with a bad sample (displacement of a cell or more, non-finite tap, position or result) is then
    // computed again, all four samples, by the general path of the direct kernel (validity of every tap, global
    // loads).  Both paths form the same products in the same order from the same values, so the result does not
    // depend on which one ran.
    //
    // What the vector ALU charges (measured, eight waves per SIMD): fp32 add/sub/mul, moves, and/or and 32-bit
    // adds on registers or literals issue in about 1.05-1.2 ns per wave and SIMD; anything that reads a scalar
    // register, and conversions, floor, compares, selects, min/max, shifts, three-operand integer adds and
    // 24-bit multiplies take about 1.85 ns.  The in-window sample is therefore written on register operands,
    // with the tap address formed from the signs of floor(p) - own coordinate (each -1 or 0 inside the window)
    // instead of converted integer coordinates.
    const T xf = (T)x, yf = (T)y, tv = t_step;
    constexpr uint32_t ESZ = sizeof(T), FS = NPL * PS * ESZ;   // bytes of an element, of one field's ring
    const uint32_t wl4 = (uint32_t)wl * ESZ;
    // |v| dt below this bound puts floor(p) at the own coordinate or one below it, whatever the rounding of
    // own coordinate - v dt (coordinates below 8192 for 4-byte elements; else no sample counts as inside)
    const T lim = sizeof(T) == 4 ? (max(max(g.nx, g.ny), g.gnz) <= 8192 ? (T)0.998046875 : (T)0) : (T)0.99999904632568359375;
    auto sign_mask = [](T v) -> uint32_t {   // all ones for a negative value, zero otherwise
        if constexpr (sizeof(T) == 4) return (uint32_t)(__float_as_int(v) >> 31);
        else return (uint32_t)(__double_as_longlong(v) >> 63);
    };
    // Next to the shell some taps are ones the reference does not read: tap b is read when 1 <= b <= n-2, tap b+1 when
    // 0 <= b <= n-3 (main.cpp:411-435).  Inside the window that is the low tap of a cell at coordinate 1 whose
    // backtrace ends below it, and the high tap of a cell at n-2 whose backtrace does not; such a tap gets a zero
    // weight.  `edge` holds the four tests of x and y as bits; a step is compiled for the axes that need them
    // (Safe<SX, SY, SZ>: the axis is at least two cells away from the shell for the whole tile, or plane).
    const uint32_t edge = (x == 1 ? 1u : 0u) | (x == g.nx - 2 ? 2u : 0u) | (y == 1 ? 4u : 0u) | (y == g.ny - 2 ? 8u : 0u);
    auto bit_mask = [](uint32_t v, int b) -> uint32_t { return (uint32_t)((int32_t)(v << (31 - b)) >> 31); };
    auto keep_unless = [](T w, uint32_t kill) -> T {   // w, or +0 where kill is all ones
        if constexpr (sizeof(T) == 4) return __uint_as_float(__float_as_uint(w) & ~kill);
        else return __longlong_as_double(__double_as_longlong(w) & ~(long long)(int32_t)kill);
    };

    // sB, sC = byte offsets of the ring slots of planes z, z+1; dAB, dBC = slot(z-1) - slot(z), slot(z) - slot(z+1)
    // (uniform over the workgroup); zedge: bit 0 = the plane is gz == 1, bit 1 = it is gz == gnz-2
    auto lds_sample = [&](auto safe_tag, int f, T dx, T dy, T dz, T zf, uint32_t zedge, uint32_t sB, uint32_t sC, uint32_t dAB,
                          uint32_t dBC, T& reach) -> T {
        using S = decltype(safe_tag);
        const T one = (T)1;
        const T dxt = dx * tv, dyt = dy * tv, dzt = dz * tv;
        const T px = xf - dxt, py = yf - dyt, pz = zf - dzt;
        const T bx = floor(px), by = floor(py), bz = floor(pz);
        const T fu = px - bx, fv = py - by, fw = pz - bz;
        FastTrace<T, OFF> tr;
        tr.x0 = one - fu; tr.x1 = fu; tr.y0 = one - fv; tr.y1 = fv; tr.z0 = one - fw; tr.z1 = fw;
        // the largest displacement of the cell's samples so far; the caller compares it with the bound once.  (A
        // maximum skips a NaN: a sample with one has a NaN result, which the caller's test of the results finds.)
        reach = fmax(fmax(reach, fabs(dxt)), fmax(fabs(dyt), fabs(dzt)));
        // floor(p) - own coordinate is -1 or 0 per axis inside the window; the masks are all ones or zero whatever
        // the value, so the address stays inside the window for every lane
        const uint32_t mx = sign_mask(bx - xf), my = sign_mask(by - yf), mz = sign_mask(bz - zf);
        if constexpr (!S::x) { tr.x0 = keep_unless(tr.x0, bit_mask(edge, 0) & mx); tr.x1 = keep_unless(tr.x1, bit_mask(edge, 1) & ~mx); }
        if constexpr (!S::y) { tr.y0 = keep_unless(tr.y0, bit_mask(edge, 2) & my); tr.y1 = keep_unless(tr.y1, bit_mask(edge, 3) & ~my); }
        if constexpr (!S::z) { tr.z0 = keep_unless(tr.z0, (0u - (zedge & 1u)) & mz); tr.z1 = keep_unless(tr.z1, (0u - (zedge >> 1)) & ~mz); }
        const uint32_t a4 = wl4 + (mx & (0u - ESZ)) + (my & (0u - (uint32_t)WX * ESZ));
        // the lower plane of the taps is z-1 (mz all ones) or z: the slot offsets follow by masking the differences
        // of the slots.  (The slot and field offsets as one scalar each: split off as a constant, the field offset
        // would not fit the offset field of a two-element LDS read and cost an addition per tap row.)
        uint32_t kB = sB + (uint32_t)f * FS, kC = sC + (uint32_t)f * FS;
        asm volatile("" : "+s"(kB), "+s"(kC));
        const uint32_t o0 = a4 + (mz & dAB) + kB;
        const uint32_t o1 = a4 + (mz & dBC) + kC;
        const T* p0 = reinterpret_cast<const T*>(lds_raw + o0);
        const T* p1 = reinterpret_cast<const T*>(lds_raw + o1);
        Taps<T> t;
        t.s[0] = p0[0]; t.s[1] = p0[1]; t.s[2] = p0[WX]; t.s[3] = p1[0];
        t.s[4] = p0[WX + 1]; t.s[5] = p1[1]; t.s[6] = p1[WX]; t.s[7] = p1[WX + 1];
        T r = fast_combine(tr, t);
        asm volatile("" : "+v"(r));   // (keeps the sums of two samples from being zipped into packed additions, which costs moves and saves nothing)
        return r;
    };
    // the general path: the sample of the direct kernel
    auto slow_sample = [&](int f, T dx, T dy, T dz, int gz, bool& oob) -> T {
        FastTrace<T, OFF> tr = fast_prepare<T, OFF, SLAB>(g, x, y, gz, dx, dy, dz, t_step);
        oob = oob || tr.oob;
        Taps<T> t;
        const T* r00 = at_byte(src[f], trace_base(g, tr));
        const T* r10 = at_byte(src[f], trace_base(g, tr) + sy);
        const T* r01 = at_byte(src[f], trace_base(g, tr) + tr.dz1);
        const T* r11 = at_byte(src[f], trace_base(g, tr) + tr.dz1 + sy);
        t.s[0] = r00[0]; t.s[1] = r00[1]; t.s[2] = r10[0]; t.s[3] = r01[0];
        t.s[4] = r10[1]; t.s[5] = r01[1]; t.s[6] = r11[0]; t.s[7] = r11[1];
        if (REFL && f < 3) {
            const T* c00 = at_byte(src[4 + f], trace_base(g, tr));
            const T* c10 = at_byte(src[4 + f], trace_base(g, tr) + sy);
            const T* c01 = at_byte(src[4 + f], trace_base(g, tr) + tr.dz1);
            const T* c11 = at_byte(src[4 + f], trace_base(g, tr) + tr.dz1 + sy);
            t.s[0] = (T)2 * c00[0] - t.s[0]; t.s[1] = (T)2 * c00[1] - t.s[1]; t.s[2] = (T)2 * c10[0] - t.s[2];
            t.s[3] = (T)2 * c01[0] - t.s[3]; t.s[4] = (T)2 * c10[1] - t.s[4]; t.s[5] = (T)2 * c01[1] - t.s[5];
            t.s[6] = (T)2 * c11[0] - t.s[6]; t.s[7] = (T)2 * c11[1] - t.s[7];
        }
        return fast_finish(tr, t);
    };

    double racc = 0;
    // The 13-point neighbourhood of the carrying velocity around a cell: from the staged planes (the step), or from
    // memory (the general path, which runs after the march, when the planes are gone).
    struct Nbh {
        T u_c, u_xp, u_ym, u_xp_ym, u_zm, u_xp_zm, v_c, v_xm, v_yp, v_xm_yp, v_zm, v_yp_zm, w_c, w_xm, w_zp, w_xm_zp, w_ym, w_ym_zp;
    };
    auto nbh_staged = [&](int z) -> Nbh {
        const T* Uc = plane(VF + 0, z) + wl; const T* Uzm = plane(VF + 0, z - 1) + wl;
        const T* Vc = plane(VF + 1, z) + wl; const T* Vzm = plane(VF + 1, z - 1) + wl;
        const T* Wc = plane(VF + 2, z) + wl; const T* Wzp = plane(VF + 2, z + 1) + wl;
        return Nbh{Uc[0], Uc[1], Uc[-WX], Uc[1 - WX], Uzm[0], Uzm[1], Vc[0], Vc[-1], Vc[WX], Vc[WX - 1], Vzm[0], Vzm[WX],
                   Wc[0], Wc[-1], Wzp[0], Wzp[-1], Wc[-WX], Wzp[-WX]};
    };
    auto nbh_memory = [&](int z) -> Nbh {   // (an interior cell: every neighbour exists)
        const OFF i = (((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x) * (OFF)sizeof(T);
        const OFF e = (OFF)sizeof(T);
        const T *Um = src[VF + 0], *Vm = src[VF + 1], *Wm = src[VF + 2];
        return Nbh{*at_byte(Um, i), *at_byte(Um, i + e), *at_byte(Um, i - sy), *at_byte(Um, i + e - sy), *at_byte(Um, i - sz), *at_byte(Um, i + e - sz),
                   *at_byte(Vm, i), *at_byte(Vm, i - e), *at_byte(Vm, i + sy), *at_byte(Vm, i + sy - e), *at_byte(Vm, i - sz), *at_byte(Vm, i + sy - sz),
                   *at_byte(Wm, i), *at_byte(Wm, i - e), *at_byte(Wm, i + sz), *at_byte(Wm, i + sz - e), *at_byte(Wm, i - sy), *at_byte(Wm, i - sy + sz)};
    };
    // One interior cell: the three velocity samples and the density sample around the neighbourhood nb.
    // `smp(f, dx, dy, dz)` is the sampler; `first` says that this is the cell's first pass.
    auto cell = [&](auto&& smp, const Nbh& nb, uint32_t fl, bool first, T& nu, T& nv, T& nw, T& nd) -> T {
        const T four = (T)4, two = (T)2;
        const bool fluid = fl & FL_SELF_FLUID;
        const T u_c = nb.u_c, u_xp = nb.u_xp, u_ym = nb.u_ym, u_xp_ym = nb.u_xp_ym, u_zm = nb.u_zm, u_xp_zm = nb.u_xp_zm;
        const T v_c = nb.v_c, v_xm = nb.v_xm, v_yp = nb.v_yp, v_xm_yp = nb.v_xm_yp, v_zm = nb.v_zm, v_yp_zm = nb.v_yp_zm;
        const T w_c = nb.w_c, w_xm = nb.w_xm, w_zp = nb.w_zp, w_xm_zp = nb.w_xm_zp, w_ym = nb.w_ym, w_ym_zp = nb.w_ym_zp;
        if (first && res_part) {   // the cells and the value of k_div_sumsq
            const uint32_t movable = (fl & FL_SELF_FLUID) ? (fl & FL_NBR) : (fl & (FL_XP | FL_YP | FL_ZP));
            if (relaxes(fl) && movable) {
                const double dv = (double)((u_xp - u_c) + (v_yp - v_c) + (w_zp - w_c));
                racc += dv * dv;
            }
        }
        // face-collocated velocities: main.cpp:261-265, :269-273, :276-280
        const T uy = (v_c + v_xm + v_yp + v_xm_yp) / four, uz = (w_c + w_xm + w_zp + w_xm_zp) / four;
        const T vx = (u_c + u_ym + u_xp + u_xp_ym) / four, vz = (w_c + w_ym + w_zp + w_ym_zp) / four;
        const T wx = (u_c + u_zm + u_xp + u_xp_zm) / four, wy = (v_c + v_zm + v_yp + v_yp_zm) / four;
        const T su = smp(0, u_c, uy, uz), sv = smp(1, vx, v_c, vz), sw = smp(2, wx, wy, w_c);
        if (fluid) { nu = su; nv = sv; nw = sw; }   // main.cpp:260
        // main.cpp:284-289
        T uc, vc, wc;
        if (density_mode == 1) {
            uc = (nu + u_xp) / two; vc = (nv + v_yp) / two; wc = (nw + w_zp) / two;
        } else {
            uc = (u_c + u_xp) / two; vc = (v_c + v_yp) / two; wc = (w_c + w_zp) / two;
        }
        const T sd = smp(3, uc, vc, wc);
        if (!(fl & FL_SELF_OBJECT)) nd = sd;
        return (su + sv) + (sw + sd);   // non-finite if a sample is (the caller's test for the general path)
    };

    // One location of plane z by the in-window path; true if the cell has to be computed again by the general path.
    auto compute = [&](auto safe_tag, int z, uint32_t fl, T (&res)[4]) -> bool {
        const int gz = z + g.gz0;
        T nu = plane(0, z)[wl], nv = plane(1, z)[wl], nw = plane(2, z)[wl], nd = plane(3, z)[wl];
        const bool interior = x > 0 && y > 0 && gz > 0 && x < g.nx - 1 && y < g.ny - 1 && gz < g.gnz - 1;
        bool bad = false;
        if (interior) {
            const T zf = (T)gz;
            const uint32_t sA = (uint32_t)((z - 1) & (NPL - 1)) * PS * ESZ, sB = (uint32_t)(z & (NPL - 1)) * PS * ESZ,
                           sC = (uint32_t)((z + 1) & (NPL - 1)) * PS * ESZ;
            const uint32_t dAB = sA - sB, dBC = sB - sC;
            const uint32_t zedge = (gz == 1 ? 1u : 0u) | (gz == g.gnz - 2 ? 2u : 0u);
            T reach = 0;
            const T chk = cell([&](int f, T dx, T dy, T dz) { return lds_sample(safe_tag, f, dx, dy, dz, zf, zedge, sB, sC, dAB, dBC, reach); },
                               nbh_staged(z), fl, true, nu, nv, nw, nd);
            bad = !(reach < lim) || !isfinite(chk);
        }
        res[0] = nu; res[1] = nv; res[2] = nw; res[3] = nd;
        return bad;
    };
    // The same location by the general path, after the march (rare): everything comes from memory, the cell's own
    // carried values as the staging formed them.
    auto compute_general = [&](int z, T (&res)[4]) {
        const int gz = z + g.gz0;
        const OFF i = (((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x) * (OFF)sizeof(T);
        const uint32_t fl = flags[((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x];
        T own[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            own[f] = *at_byte(src[f], i);
            if (REFL && f < 3) own[f] = (T)2 * *at_byte(src[4 + f], i) - own[f];
        }
        T nu = own[0], nv = own[1], nw = own[2], nd = own[3];
        bool oob_v = false, oob_d = false;
        cell([&](int f, T dx, T dy, T dz) { return slow_sample(f, dx, dy, dz, gz, f == 3 ? oob_d : oob_v); }, nbh_memory(z), fl, false, nu, nv, nw, nd);
        if (SLAB) {
            const bool oob = ((fl & FL_SELF_FLUID) && oob_v) || (!(fl & FL_SELF_OBJECT) && oob_d);
            if (oob) *oob_flag = 1;
        }
        res[0] = nu; res[1] = nv; res[2] = nw; res[3] = nd;
    };
    // the results of a plane leave after the next plane but one has been committed to LDS: a wait for the staged
    // loads then does not wait for these stores as well (loads and stores share one counter)
    // (non-temporal stores: nothing reads the results again before the launch ends; with them the reflected variant,
    // which has the most streams in flight, is 7 % faster at 512^3)
    auto store = [&](int z, const T (&res)[4]) {
        const OFF i = (((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x) * (OFF)sizeof(T);
        __builtin_nontemporal_store(res[0], const_cast<T*>(at_byte<T, OFF>(ou, i))); __builtin_nontemporal_store(res[1], const_cast<T*>(at_byte<T, OFF>(ov, i)));
        __builtin_nontemporal_store(res[2], const_cast<T*>(at_byte<T, OFF>(ow, i))); __builtin_nontemporal_store(res[3], const_cast<T*>(at_byte<T, OFF>(od, i)));
        if (cu) {
            __builtin_nontemporal_store(res[0], const_cast<T*>(at_byte<T, OFF>(cu, i))); __builtin_nontemporal_store(res[1], const_cast<T*>(at_byte<T, OFF>(cv, i)));
            __builtin_nontemporal_store(res[2], const_cast<T*>(at_byte<T, OFF>(cw, i)));
        }
    };

    uint32_t again[4] = {0u, 0u, 0u, 0u};   // planes of this chunk (at most 128: kMaxChunk) whose cell goes to the general path
    for (int p = c0 - R; p <= c0 + R; ++p) { fetch(p); commit(p); }
    __syncthreads();
    for (int z = c0; z < c1; ++z) {
        // the flag byte is asked for before the plane that is fetched ahead: waiting for it then leaves
        // those loads in flight (loads return in order)
        uint32_t fl = FL_SELF_OBJECT;
        if (mine) fl = flags[((OFF)z * (OFF)g.ny + (OFF)y) * (OFF)g.nx + (OFF)x];
        fetch(z + R + 1);
        const int gz = z + g.gz0;
        T res[4];
        if (mine) {
            const bool safe_z = gz >= 2 && gz <= g.gnz - 3;
            bool bad;
            if (safe_x && safe_y && safe_z) bad = compute(AdvSafe<true, true, true>{}, z, fl, res);
            else if (safe_y && safe_z) bad = compute(AdvSafe<false, true, true>{}, z, fl, res);
            else if (safe_x && safe_z) bad = compute(AdvSafe<true, false, true>{}, z, fl, res);
            else if (safe_x && safe_y) bad = compute(AdvSafe<true, true, false>{}, z, fl, res);
            else bad = compute(AdvSafe<false, false, false>{}, z, fl, res);
            // a cell that the step could not serve is computed again after the march (its plane is noted; what the step
            // made of it is stored meanwhile and overwritten then): the general path, large and rare, stays out of the loop
            if (__builtin_expect(bad, 0)) {
                const int k = z - c0;
                again[0] |= k < 32 ? 1u << k : 0u;
                again[1] |= k >= 32 && k < 64 ? 1u << (k - 32) : 0u;
                again[2] |= k >= 64 && k < 96 ? 1u << (k - 64) : 0u;
                again[3] |= k >= 96 ? 1u << (k - 96) : 0u;
            }
        }
        commit(z + R + 1);     // into the slot of plane z-R-1, which no wave reads any more
        asm volatile("" ::: "memory");
        if (mine) store(z, res);
        __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        for (uint32_t m = again[kk]; m; m &= m - 1) {
            const int z = c0 + 32 * kk + (__ffs(m) - 1);
            T res[4];
            compute_general(z, res);
            store(z, res);
        }
    if (res_part) {   // one partial per workgroup, added in a fixed order
        double* red = reinterpret_cast<double*>(lds_raw);
        racc = wave_sum(racc);
        if ((tid & 63) == 0) red[tid >> 6] = racc;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < C::NT / 64; ++k) t += red[k];
            const int bid = tile;   // (the partials are added in the order of the tiles, wherever they ran)
            res_part[2 * bid] = t;
            res_part[2 * bid + 1] = 0;
        }
    }
}

}  // namespace fsamd
