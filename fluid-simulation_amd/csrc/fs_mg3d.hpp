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

// level-1 weights from the fine cell types (cells: 0 = Fluid)
template <typename T>
__global__ void k_mg_weights_cells(Grid g, const uint8_t* __restrict__ cells, MgLevel<T> L) {
    FS_MG_CELL(L.d)
    auto fluid = [&](int fx, int fy, int fz) { return cells[gidx(g, fx, fy, fz)] == 0; };
    T wx = 0, wy = 0, wz = 0;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            int fx = 2 * x, fy = 2 * y + a, fz = 2 * z + b;
            if (fx >= 1 && fx < g.nx && fy < g.ny && fz < g.nzl && fluid(fx, fy, fz) && fluid(fx - 1, fy, fz)) wx += 1;
            fx = 2 * x + a; fy = 2 * y; fz = 2 * z + b;
            if (fy >= 1 && fy < g.ny && fx < g.nx && fz < g.nzl && fluid(fx, fy, fz) && fluid(fx, fy - 1, fz)) wy += 1;
            fx = 2 * x + a; fy = 2 * y + b; fz = 2 * z;
            if (fz >= 1 && fz < g.nzl && fx < g.nx && fy < g.ny && fluid(fx, fy, fz) && fluid(fx, fy, fz - 1)) wz += 1;
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

// level-1 right-hand side: -div summed over the Fluid children (z, y, x order) of every level-1 cell
template <typename T>
__global__ void k_mg_fine_residual(Grid g, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ w,
                                   const uint8_t* __restrict__ flags, MgLevel<T> C) {
    FS_MG_CELL(C.d)
    T acc = 0;
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                const int fx = 2 * x + a, fy = 2 * y + b, fz = 2 * z + c;
                if (fx < 1 || fy < 1 || fz < 1 || fx >= g.nx - 1 || fy >= g.ny - 1 || fz >= g.nzl - 1) continue;
                const int64_t i = gidx(g, fx, fy, fz);
                if (!(flags[i] & FL_SELF_FLUID)) continue;
                const T div = (u[i + 1] - u[i]) + (v[i + g.sy] - v[i]) + (w[i + g.sz] - w[i]);
                acc += -div;
            }
    C.r[mgi(C.d, x, y, z)] = acc;
}

// faces between two Fluid cells take the gradient of the level-1 error of their cells' parents
template <typename T>
__global__ void k_mg_apply(Grid g, T* __restrict__ u, T* __restrict__ v, T* __restrict__ w,
                           const uint8_t* __restrict__ flags, MgLevel<T> C, T alpha) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int y = blockIdx.y * blockDim.y + threadIdx.y + 1;
    const int z = blockIdx.z + 1;
    if (x >= g.nx - 1 || y >= g.ny - 1) return;
    const int64_t i = gidx(g, x, y, z);
    const uint32_t fl = flags[i];
    if (!(fl & FL_SELF_FLUID)) return;
    const T eh = C.e[mgi(C.d, x / 2, y / 2, z / 2)];
    if (fl & FL_XM) u[i] -= alpha * (eh - C.e[mgi(C.d, (x - 1) / 2, y / 2, z / 2)]);
    if (fl & FL_YM) v[i] -= alpha * (eh - C.e[mgi(C.d, x / 2, (y - 1) / 2, z / 2)]);
    if (fl & FL_ZM) w[i] -= alpha * (eh - C.e[mgi(C.d, x / 2, y / 2, (z - 1) / 2)]);
}

// The levels of at most a few thousand cells in one workgroup: the sub-cycle below a level costs ~80
// launches of a few microseconds each when every colour of every sweep is a kernel of its own.  Same
// operations in the same per-value order as the per-level kernels; colours and stages are separated by
// workgroup barriers.
template <typename T, int MAXL>
struct MgSmall { int n; MgLevel<T> L[MAXL]; };

template <typename T, int MAXL>
__global__ void __launch_bounds__(1024)
k_mg_small_cycle(MgSmall<T, MAXL> S, int nu1, int nu2, int nu_bottom, T alpha) {
    const int tid = threadIdx.x;
    auto cells = [](const MgDims& d) { return d.nx * d.ny * d.nz; };
    auto smooth = [&](const MgLevel<T>& L, int sweeps) {
        const int n = cells(L.d);
        for (int it = 0; it < sweeps; ++it)
            for (int colour = 0; colour < 2; ++colour) {
                for (int idx = tid; idx < n; idx += 1024) {
                    const int x = idx % L.d.nx, y = (idx / L.d.nx) % L.d.ny, z = idx / (L.d.nx * L.d.ny);
                    if (((x + y + z) & 1) != colour) continue;
                    const T dg = L.diag[idx];
                    if (dg > (T)0) L.e[idx] = (L.r[idx] + mg_offdiag(L, x, y, z)) / dg;
                }
                __syncthreads();
            }
    };
    for (int l = 0; l < S.n; ++l) {
        const MgLevel<T>& L = S.L[l];
        for (int idx = tid; idx < cells(L.d); idx += 1024) L.e[idx] = 0;
        __syncthreads();
        if (l == S.n - 1) { smooth(L, nu_bottom); break; }
        smooth(L, nu1);
        const MgLevel<T>& C = S.L[l + 1];
        for (int idx = tid; idx < cells(C.d); idx += 1024) {
            const int x = idx % C.d.nx, y = (idx / C.d.nx) % C.d.ny, z = idx / (C.d.nx * C.d.ny);
            T acc = 0;
            for (int c = 0; c < 2; ++c)
                for (int b = 0; b < 2; ++b)
                    for (int a = 0; a < 2; ++a) {
                        const int fx = 2 * x + a, fy = 2 * y + b, fz = 2 * z + c;
                        if (fx >= L.d.nx || fy >= L.d.ny || fz >= L.d.nz) continue;
                        const int64_t i = mgi(L.d, fx, fy, fz);
                        const T dg = L.diag[i];
                        if (dg > (T)0) acc += L.r[i] - (dg * L.e[i] - mg_offdiag(L, fx, fy, fz));
                    }
            C.r[idx] = acc;
        }
        __syncthreads();
    }
    for (int l = S.n - 2; l >= 0; --l) {
        const MgLevel<T>& L = S.L[l];
        const MgLevel<T>& C = S.L[l + 1];
        for (int idx = tid; idx < cells(L.d); idx += 1024) {
            const int x = idx % L.d.nx, y = (idx / L.d.nx) % L.d.ny, z = idx / (L.d.nx * L.d.ny);
            if (L.diag[idx] > (T)0) L.e[idx] += alpha * C.e[mgi(C.d, x / 2, y / 2, z / 2)];
        }
        __syncthreads();
        smooth(L, nu2);
    }
}

#undef FS_MG_CELL

}  // namespace fsamd
