// Diagnostics and hand-off kernels either side of the step: vorticity, energy / mass sums, and the
// block-averaged density a renderer fetches when the full grid per frame is too much.
#pragma once
#include "fs_kernels3d.hpp"

namespace fsamd {

// |curl u| with the central differences of 3_fluid/FluidSim.h:220-229 (face values two cells apart,
// times 0.5) on global cells 2..n-3, zero elsewhere; part[] receives sum |curl|^2 (and its maximum).
template <typename T>
__global__ void __launch_bounds__(256)
k_vorticity(Grid g, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ w, int zb, int ze,
            T* __restrict__ out, double* __restrict__ part) {
    const int64_t rows = (int64_t)g.ny * (ze - zb);
    const T half = (T)0.5;
    double s = 0, m = 0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int y = (int)(r % g.ny), z = (int)(r / g.ny) + zb;
        const int gz = z + g.gz0;
        const int64_t row = gidx(g, 0, y, z);
        const bool yz_in = y >= 2 && y < g.ny - 2 && gz >= 2 && gz < g.gnz - 2;
        for (int x = threadIdx.x; x < g.nx; x += 256) {
            const int64_t i = row + x;
            T mag = 0;
            if (yz_in && x >= 2 && x < g.nx - 2) {
                const T dwdy = (w[i + g.sy] - w[i - g.sy]) * half;
                const T dvdz = (v[i + g.sz] - v[i - g.sz]) * half;
                const T dudz = (u[i + g.sz] - u[i - g.sz]) * half;
                const T dwdx = (w[i + 1] - w[i - 1]) * half;
                const T dvdx = (v[i + 1] - v[i - 1]) * half;
                const T dudy = (u[i + g.sy] - u[i - g.sy]) * half;
                const T cx = dwdy - dvdz, cy = dudz - dwdx, cz = dvdx - dudy;
                const double e = (double)cx * cx + (double)cy * cy + (double)cz * cz;
                mag = (T)sqrt(e);
                s += e;
                m = fmax(m, e);
            }
            if (out) out[i] = mag;
        }
    }
    block_sum_max(s, m);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = m; }
}

// Over the interior cells that are not Object, with the cell-centred velocity of draw_velocity_field
// (main.cpp:138-142).  what 0: sum of (u^2+v^2+w^2)/2 and the largest norm; what 1: sum of the density.
template <typename T>
__global__ void __launch_bounds__(256)
k_energy_mass(Grid g, const T* __restrict__ u, const T* __restrict__ v, const T* __restrict__ w,
              const T* __restrict__ d, const uint8_t* __restrict__ flags, int zb, int ze, int what,
              double* __restrict__ part) {
    const int64_t rows = (int64_t)(g.ny - 2) * (ze - zb);
    const T two = (T)2;
    double s = 0, m = 0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int y = (int)(r % (g.ny - 2)) + 1, z = (int)(r / (g.ny - 2)) + zb;
        const int64_t row = gidx(g, 0, y, z);
        for (int x = threadIdx.x + 1; x < g.nx - 1; x += 256) {
            const int64_t i = row + x;
            if (flags[i] & FL_SELF_OBJECT) continue;
            if (what == 0) {
                const T uc = (u[i] + u[i + 1]) / two, vc = (v[i] + v[i + g.sy]) / two, wc = (w[i] + w[i + g.sz]) / two;
                const double n2 = (double)uc * uc + (double)vc * vc + (double)wc * wc;
                s += 0.5 * n2;
                m = fmax(m, sqrt(n2));
            } else {
                s += (double)d[i];
            }
        }
    }
    block_sum_max(s, m);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s; part[2 * blockIdx.x + 1] = m; }
}

// out[bz][by][bx] = mean of d over the block of f*f*f cells (smaller at the high ends), summed in double
// in x-fastest order by one thread per block of cells.  Local planes [zb, ze), zb a multiple of f in
// global coordinates; out holds the blocks of those planes only.
template <typename T>
__global__ void __launch_bounds__(256)
k_downsample(Grid g, const T* __restrict__ d, int f, int zb, int ze, int mx, int my, T* __restrict__ out) {
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;
    const int by = blockIdx.y * blockDim.y + threadIdx.y;
    const int bz = blockIdx.z;
    if (bx >= mx || by >= my) return;
    double sum = 0;
    int cnt = 0;
    for (int z = zb + bz * f; z < zb + (bz + 1) * f && z < ze; ++z)
        for (int y = by * f; y < (by + 1) * f && y < g.ny; ++y)
            for (int x = bx * f; x < (bx + 1) * f && x < g.nx; ++x) { sum += (double)d[gidx(g, x, y, z)]; ++cnt; }
    out[((int64_t)bz * my + by) * mx + bx] = (T)(sum / cnt);
}

}  // namespace fsamd
