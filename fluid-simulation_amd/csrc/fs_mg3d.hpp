// Multigrid correction for the projection (not in the reference).
//
// The Gauss-Seidel visit of advection-reflection/main.cpp:198-233 relaxes the pressure Poisson equation
// written on the velocities: a visit subtracts the gradient of a one-cell pressure.  A V-cycle adds coarse
// corrections of the same form, so the fixed point is the one the sweeps alone converge to.
//   fine level      nu1 red-black sweeps (the sweep kernels); r = -div on Fluid cells, summed over the
//                   2x2x2 children of every level-1 cell; [coarse solve]; every face between two Fluid cells
//                   gets  face -= alpha * (e_high - e_low)  with e the level-1 error of the cell's parent;
//                   nu2 red-black sweeps.
//   coarse level l  weights of the low faces wx, wy, wz = the number of open fine faces they gather (level 1:
//                   pairs of Fluid cells; deeper: sums of the child weights), diag = sum of the six; red-black
//                   Gauss-Seidel on  diag*e - sum w_nb e_nb = r;  restriction = sum of the children's
//                   residuals; prolongation e_child += alpha * e_parent (piecewise constant; alpha = 2 makes
//                   up for the stiffness of the summed operator).
// Every value is computed by one thread in the order of the CPU restatement, so the two agree bit for bit.
#pragma once
#include "fs_kernels3d.hpp"

namespace fsamd {

struct MgDims { int nx, ny, nz; };

template <typename T>
struct MgLevel {     // device arrays of one coarse level, x fastest
    MgDims d;
    T *wx, *wy, *wz, *diag, *e, *r;
};

__device__ __forceinline__ int64_t mgi(const MgDims& d, int x, int y, int z) { return ((int64_t)z * d.ny + y) * d.nx + x; }
__device__ __forceinline__ bool mg_in(const MgDims& d, int x, int y, int z) {
    return x >= 0 && y >= 0 && z >= 0 && x < d.nx && y < d.ny && z < d.nz;
}
template <typename T>
__device__ __forceinline__ T mg_at(const MgDims& d, const T* a, int x, int y, int z) {   // 0 outside the level
    return mg_in(d, x, y, z) ? a[mgi(d, x, y, z)] : (T)0;
}

#define FS_MG_CELL(d)                                            \
    const int x = blockIdx.x * blockDim.x + threadIdx.x;         \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;         \
    const int z = blockIdx.z;                                    \
    if (x >= (d).nx || y >= (d).ny) return;

// level-1 weights from the fine cell types (cells: 0 = Fluid).  The level-1 arrays span the whole grid; a launch
// fills the planes from zc0 on (blockIdx.z counts them) from one slab's cells, whose local plane 0 is the global
// plane g.gz0 (a slab starts on an even plane, so the children of its level-1 cells are its own; the cells
// below a child, read for wz, are at worst in its halo).
template <typename T>
__global__ void k_mg_weights_cells(Grid g, const uint8_t* __restrict__ cells, MgLevel<T> L, int zc0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z + zc0;
    if (x >= L.d.nx || y >= L.d.ny) return;
    auto fluid = [&](int fx, int fy, int fz) { return cells[gidx(g, fx, fy, fz - g.gz0)] == 0; };   // fz global
    T wx = 0, wy = 0, wz = 0;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            int fx = 2 * x, fy = 2 * y + a, fz = 2 * z + b;
            if (fx >= 1 && fx < g.nx && fy < g.ny && fz < g.gnz && fluid(fx, fy, fz) && fluid(fx - 1, fy, fz)) wx += 1;
            fx = 2 * x + a; fy = 2 * y; fz = 2 * z + b;
            if (fy >= 1 && fy < g.ny && fx < g.nx && fz < g.gnz && fluid(fx, fy, fz) && fluid(fx, fy - 1, fz)) wy += 1;
            fx = 2 * x + a; fy = 2 * y + b; fz = 2 * z;
            if (fz >= 1 && fz < g.gnz && fx < g.nx && fy < g.ny && fluid(fx, fy, fz) && fluid(fx, fy, fz - 1)) wz += 1;
        }
    const int64_t i = mgi(L.d, x, y, z);
    L.wx[i] = wx; L.wy[i] = wy; L.wz[i] = wz;
}

template <typename T>
__global__ void k_mg_weights_level(MgLevel<T> F, MgLevel<T> L) {
    FS_MG_CELL(L.d)
    T wx = 0, wy = 0, wz = 0;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            wx += mg_at(F.d, F.wx, 2 * x, 2 * y + a, 2 * z + b);
            wy += mg_at(F.d, F.wy, 2 * x + a, 2 * y, 2 * z + b);
            wz += mg_at(F.d, F.wz, 2 * x + a, 2 * y + b, 2 * z);
        }
    const int64_t i = mgi(L.d, x, y, z);
    L.wx[i] = wx; L.wy[i] = wy; L.wz[i] = wz;
}

template <typename T>
__global__ void k_mg_diag(MgLevel<T> L) {
    FS_MG_CELL(L.d)
    const int64_t i = mgi(L.d, x, y, z);
    L.diag[i] = L.wx[i] + mg_at(L.d, L.wx, x + 1, y, z) + L.wy[i] + mg_at(L.d, L.wy, x, y + 1, z) + L.wz[i] +
                mg_at(L.d, L.wz, x, y, z + 1);
}

// sum over the six neighbours of weight * e, in the order x-, x+, y-, y+, z-, z+
template <typename T>
__device__ __forceinline__ T mg_offdiag(const MgLevel<T>& L, int x, int y, int z) {
    const int64_t i = mgi(L.d, x, y, z);
    T a = 0;
    a += L.wx[i] * mg_at(L.d, L.e, x - 1, y, z);
    a += mg_at(L.d, L.wx, x + 1, y, z) * mg_at(L.d, L.e, x + 1, y, z);
    a += L.wy[i] * mg_at(L.d, L.e, x, y - 1, z);
    a += mg_at(L.d, L.wy, x, y + 1, z) * mg_at(L.d, L.e, x, y + 1, z);
    a += L.wz[i] * mg_at(L.d, L.e, x, y, z - 1);
    a += mg_at(L.d, L.wz, x, y, z + 1) * mg_at(L.d, L.e, x, y, z + 1);
    return a;
}

// one colour of a red-black Gauss-Seidel sweep on a coarse level (x counts cells of the colour)
template <typename T>
__global__ void k_mg_smooth(MgLevel<T> L, int colour) {
    const int hx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    const int x = 2 * hx + ((colour + y + z) & 1);
    if (x >= L.d.nx || y >= L.d.ny) return;
    const int64_t i = mgi(L.d, x, y, z);
    const T dg = L.diag[i];
    if (dg > (T)0) L.e[i] = (L.r[i] + mg_offdiag(L, x, y, z)) / dg;
}

// r of level C = sum over the 8 children of (r - (diag*e - offdiag)) of level F, children in z, y, x order
template <typename T>
__global__ void k_mg_restrict(MgLevel<T> F, MgLevel<T> C) {
    FS_MG_CELL(C.d)
    T acc = 0;
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                const int fx = 2 * x + a, fy = 2 * y + b, fz = 2 * z + c;
                if (fx >= F.d.nx || fy >= F.d.ny || fz >= F.d.nz) continue;
                const int64_t i = mgi(F.d, fx, fy, fz);
                const T dg = F.diag[i];
                if (dg > (T)0) acc += F.r[i] - (dg * F.e[i] - mg_offdiag(F, fx, fy, fz));
            }
    C.r[mgi(C.d, x, y, z)] = acc;
}

template <typename T>
__global__ void k_mg_prolong(MgLevel<T> F, MgLevel<T> C, T alpha) {
    FS_MG_CELL(F.d)
    const int64_t i = mgi(F.d, x, y, z);
    if (F.diag[i] > (T)0) F.e[i] += alpha * C.e[mgi(C.d, x / 2, y / 2, z / 2)];
}

// level-1 right-hand side: -div summed over the Fluid children (z, y, x order) of every level-1 cell; the
// planes from zc0 on, from one slab's velocity (see k_mg_weights_cells)
template <typename T>
__global__ void k_mg_fine_residual(Grid g, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ w,
                                   const uint8_t* __restrict__ flags, MgLevel<T> C, int zc0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z + zc0;
    if (x >= C.d.nx || y >= C.d.ny) return;
    T acc = 0;
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                const int fx = 2 * x + a, fy = 2 * y + b, fz = 2 * z + c;   // fz global
                if (fx < 1 || fy < 1 || fz < 1 || fx >= g.nx - 1 || fy >= g.ny - 1 || fz >= g.gnz - 1) continue;
                const int64_t i = gidx(g, fx, fy, fz - g.gz0);
                if (!(flags[i] & FL_SELF_FLUID)) continue;
                const T div = (u[i + 1] - u[i]) + (v[i + g.sy] - v[i]) + (w[i + g.sz] - w[i]);
                acc += -div;
            }
    C.r[mgi(C.d, x, y, z)] = acc;
}

// faces between two Fluid cells take the gradient of the level-1 error of their cells' parents
// (the local planes from zb on of one slab; the level-1 array spans the whole grid)
template <typename T>
__global__ void k_mg_apply(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w,
                           const uint8_t* __restrict__ flags, MgLevel<T> C, T alpha, int zb) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    const int z = blockIdx.z + zb;
    if (x >= g.nx - 1 || y >= g.ny - 1) return;
    const int64_t i = gidx(g, x, y, z);
    const uint32_t fl = flags[i];
    if (!(fl & FL_SELF_FLUID)) return;
    const int gz = z + g.gz0;
    const T eh = C.e[mgi(C.d, x / 2, y / 2, gz / 2)];
    if (fl & FL_XM) u[i] -= alpha * (eh - C.e[mgi(C.d, (x - 1) / 2, y / 2, gz / 2)]);
    if (fl & FL_YM) v[i] -= alpha * (eh - C.e[mgi(C.d, x / 2, (y - 1) / 2, gz / 2)]);
    if (fl & FL_ZM) w[i] -= alpha * (eh - C.e[mgi(C.d, x / 2, y / 2, (gz - 1) / 2)]);
}

// The levels of at most 4096 cells in one workgroup: the sub-cycle below a level costs ~80 launches of a
// few microseconds each when every colour of every sweep is a kernel of its own, and ~80 round trips to L2
// when one workgroup runs them on global arrays.  Here the errors e (and the restricted right-hand sides) of
// all these levels live in LDS, and while a level is being smoothed every thread keeps the coefficients of
// its (up to four) cells in registers, so a colour of a sweep is six LDS reads and a barrier.  Same
// operations in the same per-value order as the per-level kernels.  Only e of the first level goes back
// to global memory (the caller prolongs it).
template <typename T, int MAXL>
struct MgSmall { int n; MgLevel<T> L[MAXL]; };

template <typename T, int MAXL>
__host__ __device__ inline int mg_small_cells(const MgSmall<T, MAXL>& S, int l) { return S.L[l].d.nx * S.L[l].d.ny * S.L[l].d.nz; }
template <typename T, int MAXL>
inline size_t mg_small_lds_bytes(const MgSmall<T, MAXL>& S) {
    size_t total = 0;
    for (int l = 0; l < S.n; ++l) total += (size_t)mg_small_cells(S, l);
    return (2 * total + (size_t)mg_small_cells(S, 0)) * sizeof(T);
}

template <typename T, int MAXL>
__global__ void __launch_bounds__(1024)
k_mg_small_cycle(MgSmall<T, MAXL> S, int nu1, int nu2, int nu_bottom, T alpha) {
    constexpr int KMAX = 4, NT = 1024;   // cells per thread: levels hold at most KMAX * NT cells
    extern __shared__ __align__(16) unsigned char mg_lds_raw[];
    const int tid = threadIdx.x;
    int total = 0;
    for (int l = 0; l < S.n; ++l) total += mg_small_cells(S, l);
    T* E = reinterpret_cast<T*>(mg_lds_raw);   // e of every level, one after the other
    T* R = E + total;                          // r of every level but the first (same offsets)
    T* RES = R + total;                        // residuals of the level being restricted
    auto offset = [&](int l) { int o = 0; for (int k = 0; k < l; ++k) o += mg_small_cells(S, k); return o; };

    // coefficients of this thread's cells on the level being worked on
    T dg[KMAX], rr[KMAX], w[KMAX][6];
    int mask[KMAX];   // bits 0-5: neighbour x-, x+, y-, y+, z-, z+ exists; bit 6: colour; bit 7: cell exists
    int nx = 1, nxy = 1, n = 0;
    T* e = E;
    auto load = [&](int l) {
        const MgLevel<T>& L = S.L[l];
        nx = L.d.nx; nxy = L.d.nx * L.d.ny; n = nxy * L.d.nz;
        e = E + offset(l);
        const T* rl = R + offset(l);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * NT;
            mask[k] = 0; dg[k] = 0; rr[k] = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) w[k][j] = 0;
            if (idx < n) {
                const int x = idx % nx, y = (idx / nx) % L.d.ny, z = idx / nxy;
                dg[k] = L.diag[idx];
                rr[k] = l == 0 ? L.r[idx] : rl[idx];
                w[k][0] = L.wx[idx]; w[k][1] = mg_at(L.d, L.wx, x + 1, y, z);
                w[k][2] = L.wy[idx]; w[k][3] = mg_at(L.d, L.wy, x, y + 1, z);
                w[k][4] = L.wz[idx]; w[k][5] = mg_at(L.d, L.wz, x, y, z + 1);
                mask[k] = (x > 0) | (x < nx - 1) << 1 | (y > 0) << 2 | (y < L.d.ny - 1) << 3 | (z > 0) << 4 |
                          (z < L.d.nz - 1) << 5 | ((x + y + z) & 1) << 6 | 1 << 7;
            }
        }
    };
    // sum over the six neighbours of weight * e in the order of mg_offdiag (e = 0 outside the level)
    auto offdiag = [&](int k, int idx) {
        const int m = mask[k];
        T a = 0;
        a += w[k][0] * ((m & 1) ? e[idx - 1] : (T)0);
        a += w[k][1] * ((m & 2) ? e[idx + 1] : (T)0);
        a += w[k][2] * ((m & 4) ? e[idx - nx] : (T)0);
        a += w[k][3] * ((m & 8) ? e[idx + nx] : (T)0);
        a += w[k][4] * ((m & 16) ? e[idx - nxy] : (T)0);
        a += w[k][5] * ((m & 32) ? e[idx + nxy] : (T)0);
        return a;
    };
    auto smooth = [&](int sweeps) {
        for (int it = 0; it < sweeps; ++it)
            for (int colour = 0; colour < 2; ++colour) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int idx = tid + k * NT;
                    if ((mask[k] & 128) && ((mask[k] >> 6) & 1) == colour && dg[k] > (T)0)
                        e[idx] = (rr[k] + offdiag(k, idx)) / dg[k];
                }
                __syncthreads();
            }
    };

    for (int l = 0; l < S.n; ++l) {
        load(l);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (mask[k] & 128) e[tid + k * NT] = 0;
        __syncthreads();
        if (l == S.n - 1) { smooth(nu_bottom); break; }
        smooth(nu1);
        // r of level l+1 = sum over the 8 children (z, y, x order) of r - (diag*e - offdiag), cells without
        // a diagonal contributing nothing
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * NT;
            if (mask[k] & 128) RES[idx] = dg[k] > (T)0 ? rr[k] - (dg[k] * e[idx] - offdiag(k, idx)) : (T)0;
        }
        __syncthreads();
        const MgDims fd = S.L[l].d, cd = S.L[l + 1].d;
        T* rc = R + offset(l + 1);
        for (int idx = tid; idx < cd.nx * cd.ny * cd.nz; idx += NT) {
            const int x = idx % cd.nx, y = (idx / cd.nx) % cd.ny, z = idx / (cd.nx * cd.ny);
            T acc = 0;
            for (int c = 0; c < 2; ++c)
                for (int b = 0; b < 2; ++b)
                    for (int a = 0; a < 2; ++a) {
                        const int fx = 2 * x + a, fy = 2 * y + b, fz = 2 * z + c;
                        if (fx >= fd.nx || fy >= fd.ny || fz >= fd.nz) continue;
                        acc += RES[(fz * fd.ny + fy) * fd.nx + fx];
                    }
            rc[idx] = acc;
        }
        __syncthreads();
    }
    for (int l = S.n - 2; l >= 0; --l) {
        load(l);
        const MgDims fd = S.L[l].d, cd = S.L[l + 1].d;
        const T* ec = E + offset(l + 1);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * NT;
            if ((mask[k] & 128) && dg[k] > (T)0) {
                const int x = idx % fd.nx, y = (idx / fd.nx) % fd.ny, z = idx / (fd.nx * fd.ny);
                e[idx] += alpha * ec[((z / 2) * cd.ny + y / 2) * cd.nx + x / 2];
            }
        }
        __syncthreads();
        smooth(nu2);
    }
    // (the last smooth ends on a barrier, and e points at the first level again)
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (mask[k] & 128) S.L[0].e[tid + k * NT] = e[tid + k * NT];
}

#undef FS_MG_CELL

}  // namespace fsamd
