// C ABI of the 2D fp64 smoke solver (fs2d_* in include/fluidsim_amd.h): the
// reference's class FluidSim of 3_fluid/FluidSim.{h,cpp} on the GPU.
//
// Arrays keep the reference's Array2T layout (include/Array2T.h:59-63):
// element (x, y) at data[x*size1 + y], y fastest; cell fields are rx*ry, u is
// (rx+1)*ry, v is rx*(ry+1) (include/MACGrid2.h:14-15).  Threads run along y.
// Every kernel performs the reference's fp64 operations in the reference's
// order (no fused multiply-add), so each stage is bit-identical to the CPU
// code; the pressure solve is offered in the reference's lexicographic order
// (exact, executed as skewed wavefronts by one workgroup) and in red-black
// order (parallel, converges to the same tolerance).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "fs_common.hpp"

using namespace fsamd;

namespace {

struct G2 {
    int rx, ry;
    double dx, idx;
};

__device__ __forceinline__ size_t ic(const G2& g, int x, int y) { return (size_t)x * g.ry + y; }          // cells, u
__device__ __forceinline__ size_t iv(const G2& g, int x, int y) { return (size_t)x * (g.ry + 1) + y; }    // v

#define FS2D_XY(nx_, ny_)                                          \
    const int y = blockIdx.x * blockDim.x + threadIdx.x;           \
    const int x = blockIdx.y * blockDim.y + threadIdx.y;           \
    if (x >= (nx_) || y >= (ny_)) return

// Grid2::applySource (include/Grid2.h:55-61)
__global__ void k2_source(G2 g, double* d, int x0, int x1, int y0, int y1) {
    FS2D_XY(g.rx, g.ry);
    if (x >= x0 && x < x1 && y >= y0 && y < y1) d[ic(g, x, y)] = 1.0;
}

// addBuoyancy (3_fluid/FluidSim.h:109-118)
__global__ void k2_buoyancy(G2 g, double* fy, const double* d) {
    FS2D_XY(g.rx, g.ry + 1);
    if (y < 1 || y >= g.ry) return;
    const double scaling = 64.0 / g.rx;
    fy[iv(g, x, y)] += 0.1 * (d[ic(g, x, y - 1)] + d[ic(g, x, y)]) / 2.0 * scaling;
}

// addWind (FluidSim.h:129-133) with the scalar computed on the host, and addForce (FluidSim.h:136-148)
__global__ void k2_add_scalar(size_t n, double* a, double s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += s;
}
__global__ void k2_axpy(size_t n, double* a, double dt, const double* f) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += dt * f[i];
}

// setNeumann (FluidSim.h:169-187), setZero (FluidSim.h:189-207)
__global__ void k2_neumann(G2 g, double* u, double* v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < g.ry) {
        u[ic(g, 0, t)] = u[ic(g, 2, t)];
        u[ic(g, g.rx, t)] = u[ic(g, g.rx - 2, t)];
    }
    if (t < g.rx) {
        v[iv(g, t, 0)] = v[iv(g, t, 2)];
        v[iv(g, t, g.ry)] = v[iv(g, t, g.ry - 2)];
    }
}
__global__ void k2_zero_walls(G2 g, double* u, double* v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < g.rx + 1) {
        u[ic(g, t, 0)] = 0;
        u[ic(g, t, g.ry - 1)] = 0;
    }
    if (t < g.ry + 1) {
        v[iv(g, 0, t)] = 0;
        v[iv(g, g.rx - 1, t)] = 0;
    }
}

// computeDivergence (FluidSim.h:209-218)
__global__ void k2_divergence(G2 g, double* div, const double* u, const double* v) {
    FS2D_XY(g.rx, g.ry);
    if (x < 1 || y < 1 || x >= g.rx - 1 || y >= g.ry - 1) return;
    const double dudx = (u[ic(g, x + 1, y)] - u[ic(g, x, y)]) * g.idx;
    const double dvdy = (v[iv(g, x, y + 1)] - v[iv(g, x, y)]) * g.idx;
    div[ic(g, x, y)] = dudx + dvdy;
}

// computeVorticity (FluidSim.h:220-229)
__global__ void k2_vorticity(G2 g, double* vort, const double* u, const double* v) {
    FS2D_XY(g.rx, g.ry);
    if (x < 2 || y < 2 || x >= g.rx - 2 || y >= g.ry - 2) return;
    const double dudy = (u[ic(g, x, y + 1)] - u[ic(g, x, y - 1)]) * g.idx * 0.5;
    const double dvdx = (v[iv(g, x + 1, y)] - v[iv(g, x - 1, y)]) * g.idx * 0.5;
    vort[ic(g, x, y)] = dvdx - dudy;
}

// copyBorder (FluidSim.h:231-243): the x borders first, then the y borders (corners take the second)
__global__ void k2_border_x(G2 g, double* p) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= g.ry) return;
    p[ic(g, 0, y)] = p[ic(g, 1, y)];
    p[ic(g, g.rx - 1, y)] = p[ic(g, g.rx - 2, y)];
}
__global__ void k2_border_y(G2 g, double* p) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= g.rx) return;
    p[ic(g, x, 0)] = p[ic(g, x, 1)];
    p[ic(g, x, g.ry - 1)] = p[ic(g, x, g.ry - 2)];
}

// correctVelocity (3_fluid/FluidSim.cpp:50-66)
__global__ void k2_correct(G2 g, double* u, double* v, const double* p, double dt) {
    FS2D_XY(g.rx + 1, g.ry + 1);
    if (y >= 1 && y < g.ry - 1 && x >= 1 && x < g.rx) u[ic(g, x, y)] -= dt * (p[ic(g, x, y)] - p[ic(g, x - 1, y)]) * g.idx;
    if (y >= 1 && y < g.ry && x >= 1 && x < g.rx - 1) v[iv(g, x, y)] -= dt * (p[ic(g, x, y)] - p[ic(g, x, y - 1)]) * g.idx;
}

// --- solvePoisson (3_fluid/FluidSim.cpp:6-48), the whole loop in one workgroup -------------------
__device__ __forceinline__ void poisson_visit(const G2& g, double* p, const double* div, int x, int y, double dt,
                                              double dx2) {
    const double b = -div[ic(g, x, y)] / dt * 1.0;
    p[ic(g, x, y)] = (dx2 * b + p[ic(g, x - 1, y)] + p[ic(g, x + 1, y)] + p[ic(g, x, y - 1)] + p[ic(g, x, y + 1)]) / 4.0;
}
__device__ __forceinline__ double poisson_r2(const G2& g, const double* p, const double* div, int x, int y, double dt,
                                             double dx2) {
    const double b = -div[ic(g, x, y)] / dt * 1.0;
    const double r = b - (4 * p[ic(g, x, y)] - p[ic(g, x - 1, y)] - p[ic(g, x + 1, y)] - p[ic(g, x, y - 1)] -
                          p[ic(g, x, y + 1)]) / dx2;
    return r * r;
}

// ordering 0: the reference's lexicographic order (y outer, x inner).  Cell (x, y) runs at wavefront
// x + 2y: its left neighbour ran one front earlier, the cell below two fronts earlier, the right and
// upper neighbours have not run yet -- exactly the values the serial loop sees.  The residual is summed
// in the serial order by one wave.  ordering 1: red-black ((x+y)&1 == 0 first), tree-summed residual.
__global__ void __launch_bounds__(1024)
k2_poisson(G2 g, double* p, const double* div, double dt, double acc, int max_iter, int ordering, int* out_iters,
           double* out_residual) {
    __shared__ double s_part[16];
    __shared__ double s_res;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double dx2 = g.dx * g.dx;
    const int ix = g.rx - 2, iy = g.ry - 2;   // interior extents
    double residual = acc + 1;
    int it = 0;
    for (; residual > acc && it < max_iter; ++it) {
        if (ordering == 0) {
            const int fronts = ix + 2 * iy;
            for (int f = 0; f < fronts; ++f) {
                // rows with 0 <= f - 2*row < ix
                for (int row = tid; row < iy; row += nt) {
                    const int col = f - 2 * row;
                    if (col >= 0 && col < ix) poisson_visit(g, p, div, col + 1, row + 1, dt, dx2);
                }
                __syncthreads();
            }
        } else {
            for (int colour = 0; colour < 2; ++colour) {
                for (int c = tid; c < ix * iy; c += nt) {
                    const int xx = c / iy + 1, yy = c % iy + 1;
                    if (((xx + yy) & 1) == colour) poisson_visit(g, p, div, xx, yy, dt, dx2);
                }
                __syncthreads();
            }
        }
        // residual (FluidSim.cpp:25-43)
        if (ordering == 0) {
            if (tid < 64) {
                double acc_r = 0;
                const int n = ix * iy;
                for (int base = 0; base < n; base += 64) {
                    const int c = base + tid;   // serial order: y outer, x inner
                    double r2 = 0;
                    if (c < n) r2 = poisson_r2(g, p, div, c % ix + 1, c / ix + 1, dt, dx2);
                    for (int l = 0; l < 64; ++l) acc_r += __shfl(r2, l, 64);   // adding 0 past the end is exact
                }
                if (tid == 0) s_res = acc_r;
            }
        } else {
            double part = 0;
            for (int c = tid; c < ix * iy; c += nt) part += poisson_r2(g, p, div, c / iy + 1, c % iy + 1, dt, dx2);
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            if ((tid & 63) == 0) s_part[tid >> 6] = part;
            __syncthreads();
            if (tid == 0) {
                double tot = 0;
                for (int w = 0; w < (nt + 63) / 64; ++w) tot += s_part[w];
                s_res = tot;
            }
        }
        __syncthreads();
        residual = sqrt(s_res);
        residual /= (double)((g.rx - 2) * (g.ry - 2));
        __syncthreads();
    }
    if (tid == 0) { *out_iters = it; *out_residual = residual; }
}

// --- semi-Lagrangian advection (3_fluid/FluidSim.cpp:87-215) ----------------------------------------
// which: 0 density (cells), 1 u, 2 v.  src is the advected field, (u, v) the carrying velocity.
// Writes only the range the reference loops over; dst is zero elsewhere (its temporaries start at 0).
__device__ __forceinline__ void backtrace(const G2& g, int which, int x, int y, const double* u, const double* v,
                                          double dt, double* lx_out, double* ly_out) {
    double lxv, lyv, lx, ly;
    if (which == 0) {
        lxv = (u[ic(g, x, y)] + u[ic(g, x + 1, y)]) / 2;
        lyv = (v[iv(g, x, y)] + v[iv(g, x, y + 1)]) / 2;
        lx = x - dt * g.idx * lxv; ly = y - dt * g.idx * lyv;
        if (lx < 1) lx = 1;
        if (ly < 1) ly = 1;
        if (lx > g.rx - 2) lx = g.rx - 2;
        if (ly > g.ry - 2) ly = g.ry - 2;
    } else if (which == 1) {
        lxv = u[ic(g, x, y)];
        lyv = (v[iv(g, x, y)] + v[iv(g, x - 1, y)] + v[iv(g, x - 1, y + 1)] + v[iv(g, x, y + 1)]) / 4;
        lx = x - dt * g.idx * lxv; ly = y - dt * g.idx * lyv;
        if (lx < 1.5) lx = 1.5;
        if (ly < 1.5) ly = 1.5;
        if (lx > g.rx - 1.5) lx = g.rx - 1.5;
        if (ly > g.ry - 2.5) ly = g.ry - 2.5;
    } else {
        lxv = (u[ic(g, x, y)] + u[ic(g, x + 1, y)] + u[ic(g, x + 1, y - 1)] + u[ic(g, x, y - 1)]) / 4;
        lyv = v[iv(g, x, y)];
        lx = x - dt * g.idx * lxv; ly = y - dt * g.idx * lyv;
        if (lx < 1.5) lx = 1.5;
        if (ly < 1.5) ly = 1.5;
        if (lx > g.rx - 2.5) lx = g.rx - 2.5;
        if (ly > g.ry - 1.5) ly = g.ry - 1.5;
    }
    *lx_out = lx; *ly_out = ly;
}

__device__ __forceinline__ bool in_range(const G2& g, int which, int x, int y) {
    if (which == 0) return x >= 1 && x < g.rx - 1 && y >= 1 && y < g.ry - 1;
    if (which == 1) return x >= 1 && x < g.rx && y >= 1 && y < g.ry - 1;
    return x >= 1 && x < g.rx - 1 && y >= 1 && y < g.ry;
}
__device__ __forceinline__ size_t iw(const G2& g, int which, int x, int y) { return which == 2 ? iv(g, x, y) : ic(g, x, y); }

__global__ void k2_advect(G2 g, int which, double* dst, const double* src, const double* u, const double* v, double dt) {
    FS2D_XY(g.rx + 1, g.ry + 1);
    const int sx = which == 1 ? g.rx + 1 : g.rx, sy = which == 2 ? g.ry + 1 : g.ry;
    if (x >= sx || y >= sy) return;
    double out = 0;
    if (in_range(g, which, x, y)) {
        double lx, ly;
        backtrace(g, which, x, y, u, v, dt, &lx, &ly);
        const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
        const double xw = lx - xl, yw = ly - yl;
        out = xw * yw * src[iw(g, which, xh, yh)] + (1 - xw) * yw * src[iw(g, which, xl, yh)] +
              xw * (1 - yw) * src[iw(g, which, xh, yl)] + (1 - xw) * (1 - yw) * src[iw(g, which, xl, yl)];
    }
    dst[iw(g, which, x, y)] = out;
}

// MacCormackUpdate + MacCormackClamp (FluidSim.cpp:217-389) for one field:
// out = fwd + 0.5*(orig - bwd) inside the loop range, reverted to fwd where it leaves the min/max of
// the four corners of the ORIGINAL field around the forward backtrace; out = fwd elsewhere.
__global__ void k2_maccormack(G2 g, int which, double* out, const double* orig, const double* fwd, const double* bwd,
                              const double* u0, const double* v0, double dt) {
    FS2D_XY(g.rx + 1, g.ry + 1);
    const int sx = which == 1 ? g.rx + 1 : g.rx, sy = which == 2 ? g.ry + 1 : g.ry;
    if (x >= sx || y >= sy) return;
    const size_t i = iw(g, which, x, y);
    double val = fwd[i];
    if (in_range(g, which, x, y)) {
        val += 0.5 * (orig[i] - bwd[i]);
        double lx, ly;
        backtrace(g, which, x, y, u0, v0, dt, &lx, &ly);
        const int xl = (int)lx, yl = (int)ly, xh = xl + 1, yh = yl + 1;
        const double a = orig[iw(g, which, xl, yl)], b = orig[iw(g, which, xl, yh)], c = orig[iw(g, which, xh, yl)],
                     d = orig[iw(g, which, xh, yh)];
        double mn = 1e10, mx = -1e10;
        mn = fmin(a, mn); mn = fmin(b, mn); mn = fmin(c, mn); mn = fmin(d, mn);
        mx = fmax(a, mx); mx = fmax(b, mx); mx = fmax(c, mx); mx = fmax(d, mx);
        if (val < mn || val > mx) val = fwd[i];
    }
    out[i] = val;
}

}  // namespace

struct fs2d_sim {
    int device = 0;
    hipStream_t stream = nullptr;
    G2 g{};
    double dt = 0, acc = 1e-5;
    int iter = 1000, wind_on = 0, mac_on = 1, ordering = FS_ORDER_LEXICOGRAPHIC;
    double wind_r = 0, time = 0;
    unsigned long step = 0;
    int last_iters = 0;
    double last_residual = 0;
    // state: density, pressure, divergence, vorticity, u, v, fx, fy; scratch: orig/fwd/bwd of d,u,v
    double* f[8] = {nullptr};
    double* orig[3] = {nullptr}, *fwd[3] = {nullptr}, *bwd[3] = {nullptr};
    int* d_iters = nullptr;
    double* d_res = nullptr;
    std::string err;
};

namespace {

size_t count2(const fs2d_sim* s, int which) {
    const size_t rx = s->g.rx, ry = s->g.ry;
    if (which == 4 || which == 6) return (rx + 1) * ry;
    if (which == 5 || which == 7) return rx * (ry + 1);
    return rx * ry;
}
const int kAdv[3] = {0, 4, 5};   // state indices of density, u, v

#define S2_HIP(call) FS_HIP(s->err, call)
#define S2_GUARD(s)                                       \
    if (!(s)) return FS_EINVAL;                           \
    DeviceGuard guard__((s)->device);                     \
    if (!guard__.ok) { (s)->err = "hipSetDevice failed"; return FS_EHIP; }

dim3 blk2() { return dim3(64, 4, 1); }
dim3 grd2(int nx, int ny) { return dim3(div_up(ny, 64), div_up(nx, 4), 1); }

int st_source(fs2d_sim* s) {   // p_density->applySource(0.45, 0.55, 0.1, 0.15), FluidSim.h:45, :73
    const int x0 = (int)(0.45 * s->g.rx), x1 = (int)(0.55 * s->g.rx), y0 = (int)(0.1 * s->g.ry), y1 = (int)(0.15 * s->g.ry);
    hipLaunchKernelGGL(k2_source, grd2(s->g.rx, s->g.ry), blk2(), 0, s->stream, s->g, s->f[0], x0, x1, y0, y1);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_buoyancy(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_buoyancy, grd2(s->g.rx, s->g.ry + 1), blk2(), 0, s->stream, s->g, s->f[7], s->f[0]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_wind(fs2d_sim* s) {   // FluidSim.h:120-134; the cosine factor is a host scalar as in the reference
    const double scaling = 64.0 / s->g.rx;
    s->wind_r += 1;
    const double fx = 2e-2 * cos(5e-2 * s->wind_r) * cos(3e-2 * s->wind_r) * scaling;
    const size_t n = count2(s, 6);
    hipLaunchKernelGGL(k2_add_scalar, dim3((unsigned)div_up((int)n, 256)), dim3(256), 0, s->stream, n, s->f[6], fx);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_force(fs2d_sim* s) {
    size_t n = count2(s, 4);
    hipLaunchKernelGGL(k2_axpy, dim3((unsigned)div_up((int)n, 256)), dim3(256), 0, s->stream, n, s->f[4], s->dt, s->f[6]);
    n = count2(s, 5);
    hipLaunchKernelGGL(k2_axpy, dim3((unsigned)div_up((int)n, 256)), dim3(256), 0, s->stream, n, s->f[5], s->dt, s->f[7]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_neumann(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_neumann, dim3(div_up(std::max(s->g.rx, s->g.ry), 256)), dim3(256), 0, s->stream, s->g, s->f[4], s->f[5]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_zero(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_zero_walls, dim3(div_up(std::max(s->g.rx, s->g.ry) + 1, 256)), dim3(256), 0, s->stream, s->g, s->f[4], s->f[5]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_divergence(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_divergence, grd2(s->g.rx, s->g.ry), blk2(), 0, s->stream, s->g, s->f[2], s->f[4], s->f[5]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_vorticity(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_vorticity, grd2(s->g.rx, s->g.ry), blk2(), 0, s->stream, s->g, s->f[3], s->f[4], s->f[5]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_border(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_border_x, dim3(div_up(s->g.ry, 256)), dim3(256), 0, s->stream, s->g, s->f[1]);
    hipLaunchKernelGGL(k2_border_y, dim3(div_up(s->g.rx, 256)), dim3(256), 0, s->stream, s->g, s->f[1]);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_poisson(fs2d_sim* s, int ordering) {
    hipLaunchKernelGGL(k2_poisson, dim3(1), dim3(1024), 0, s->stream, s->g, s->f[1], s->f[2], s->dt, s->acc, s->iter,
                       ordering == FS_ORDER_RED_BLACK ? 1 : 0, s->d_iters, s->d_res);
    S2_HIP(hipGetLastError());
    S2_HIP(hipMemcpyAsync(&s->last_iters, s->d_iters, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    S2_HIP(hipMemcpyAsync(&s->last_residual, s->d_res, sizeof(double), hipMemcpyDeviceToHost, s->stream));
    S2_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}
int st_correct(fs2d_sim* s) {
    hipLaunchKernelGGL(k2_correct, grd2(s->g.rx + 1, s->g.ry + 1), blk2(), 0, s->stream, s->g, s->f[4], s->f[5], s->f[1], s->dt);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_pressure(fs2d_sim* s, int ordering) {   // solvePressure, FluidSim.h:150-167
    FS_TRY(st_neumann(s)); FS_TRY(st_zero(s)); FS_TRY(st_divergence(s)); FS_TRY(st_border(s));
    FS_TRY(st_poisson(s, ordering)); FS_TRY(st_correct(s)); FS_TRY(st_vorticity(s));
    return st_divergence(s);
}
int advect_all(fs2d_sim* s, double* const dst[3], double* const src[3], double dt) {
    for (int k = 0; k < 3; ++k)
        hipLaunchKernelGGL(k2_advect, grd2(s->g.rx + 1, s->g.ry + 1), blk2(), 0, s->stream, s->g, k, dst[k], src[k],
                           s->orig[1], s->orig[2], dt);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_advect(fs2d_sim* s) {   // advectValues, FluidSim.cpp:68-85
    double* state[3] = {s->f[0], s->f[4], s->f[5]};
    for (int k = 0; k < 3; ++k)
        S2_HIP(hipMemcpyAsync(s->orig[k], state[k], count2(s, kAdv[k]) * 8, hipMemcpyDeviceToDevice, s->stream));
    if (!s->mac_on) return advect_all(s, state, s->orig, s->dt) == FS_OK ? FS_OK : FS_EHIP;
    FS_TRY(advect_all(s, s->fwd, s->orig, s->dt));
    FS_TRY(advect_all(s, s->bwd, s->fwd, -s->dt));   // MacCormackUpdate: m_dt *= -1 (FluidSim.cpp:219)
    for (int k = 0; k < 3; ++k)
        hipLaunchKernelGGL(k2_maccormack, grd2(s->g.rx + 1, s->g.ry + 1), blk2(), 0, s->stream, s->g, k, state[k], s->orig[k],
                           s->fwd[k], s->bwd[k], s->orig[1], s->orig[2], s->dt);
    S2_HIP(hipGetLastError());
    return FS_OK;
}
int st_reset_force(fs2d_sim* s) {
    S2_HIP(hipMemsetAsync(s->f[6], 0, count2(s, 6) * 8, s->stream));
    S2_HIP(hipMemsetAsync(s->f[7], 0, count2(s, 7) * 8, s->stream));
    return FS_OK;
}

}  // namespace

extern "C" {

int fs2d_create(int res_x, int res_y, int device, fs2d_sim** out) {
    if (!out) return FS_EINVAL;
    *out = nullptr;
    if (res_x < 5 || res_y < 5) return FS_EINVAL;
    fs2d_sim* s = new (std::nothrow) fs2d_sim();
    if (!s) return FS_ENOMEM;
    *out = s;
    if (device >= 0) s->device = device;
    else S2_HIP(hipGetDevice(&s->device));
    DeviceGuard guard(s->device);
    if (!guard.ok) { s->err = "hipSetDevice failed"; return FS_EHIP; }
    // init(), 3_fluid/FluidSim.h:16-29
    s->g.rx = res_x; s->g.ry = res_y;
    const double size_x = res_x;
    s->g.dx = size_x / res_x;
    s->g.idx = res_x / size_x;
    s->dt = 0.005 * sqrt((res_x + res_y) * 0.5);
    S2_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (int k = 0; k < 8; ++k) S2_HIP(hipMalloc((void**)&s->f[k], count2(s, k) * 8));
    for (int k = 0; k < 3; ++k) {
        S2_HIP(hipMalloc((void**)&s->orig[k], count2(s, kAdv[k]) * 8));
        S2_HIP(hipMalloc((void**)&s->fwd[k], count2(s, kAdv[k]) * 8));
        S2_HIP(hipMalloc((void**)&s->bwd[k], count2(s, kAdv[k]) * 8));
    }
    S2_HIP(hipMalloc((void**)&s->d_iters, sizeof(int)));
    S2_HIP(hipMalloc((void**)&s->d_res, sizeof(double)));
    S2_HIP(hipMemsetAsync(s->f[3], 0, count2(s, 3) * 8, s->stream));
    return fs2d_reset(s);
}

void fs2d_destroy(fs2d_sim* s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto p : s->f) (void)hipFree(p);
    for (int k = 0; k < 3; ++k) { (void)hipFree(s->orig[k]); (void)hipFree(s->fwd[k]); (void)hipFree(s->bwd[k]); }
    (void)hipFree(s->d_iters); (void)hipFree(s->d_res);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

const char* fs2d_last_error(const fs2d_sim* s) { return s ? s->err.c_str() : "null handle"; }

int fs2d_reset(fs2d_sim* s) {   // Simulation::reset + FluidSim::resetMembers (FluidSim.h:43-50)
    S2_GUARD(s);
    s->time = 0; s->step = 0;
    for (int k : {0, 1, 2, 4, 5, 6, 7}) S2_HIP(hipMemsetAsync(s->f[k], 0, count2(s, k) * 8, s->stream));
    FS_TRY(st_source(s));
    S2_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

int fs2d_set_params(fs2d_sim* s, double dt, double acc, int iter, int wind, int maccormack, int ordering) {
    if (!s) return FS_EINVAL;
    if (ordering != FS_ORDER_RED_BLACK && ordering != FS_ORDER_LEXICOGRAPHIC) { s->err = "bad ordering"; return FS_EINVAL; }
    if (dt > 0) s->dt = dt;        // Simulation::setTimestep
    if (acc > 0) s->acc = acc;     // setAccuracy, FluidSim.h:263
    if (iter > 0) s->iter = iter;  // setIteration, FluidSim.h:264
    s->wind_on = wind != 0;        // setWind, FluidSim.h:265
    s->mac_on = maccormack != 0;   // setMacCormack, FluidSim.h:266
    s->ordering = ordering;
    return FS_OK;
}

int fs2d_get_info(fs2d_sim* s, double* dt, double* time, uint64_t* step, int* last_iters, double* last_residual) {
    if (!s) return FS_EINVAL;
    if (dt) *dt = s->dt;
    if (time) *time = s->time;
    if (step) *step = s->step;
    if (last_iters) *last_iters = s->last_iters;
    if (last_residual) *last_residual = s->last_residual;
    return FS_OK;
}

int fs2d_field_size(const fs2d_sim* s, int which, int* size0, int* size1) {
    if (!s || which < 0 || which > 7) return FS_EINVAL;
    if (size0) *size0 = (which == 4 || which == 6) ? s->g.rx + 1 : s->g.rx;
    if (size1) *size1 = (which == 5 || which == 7) ? s->g.ry + 1 : s->g.ry;
    return FS_OK;
}

int fs2d_get_field(fs2d_sim* s, int which, double* host, size_t count) {
    S2_GUARD(s);
    if (!host || which < 0 || which > 7 || count != count2(s, which)) { s->err = "bad field or count"; return FS_EINVAL; }
    S2_HIP(hipStreamSynchronize(s->stream));
    S2_HIP(hipMemcpy(host, s->f[which], count * 8, hipMemcpyDeviceToHost));
    return FS_OK;
}

int fs2d_set_field(fs2d_sim* s, int which, const double* host, size_t count) {
    S2_GUARD(s);
    if (!host || which < 0 || which > 7 || count != count2(s, which)) { s->err = "bad field or count"; return FS_EINVAL; }
    S2_HIP(hipStreamSynchronize(s->stream));
    S2_HIP(hipMemcpy(s->f[which], host, count * 8, hipMemcpyHostToDevice));
    return FS_OK;
}

int fs2d_stage(fs2d_sim* s, int op) {
    S2_GUARD(s);
    switch (op) {
        case 0: return st_source(s);
        case 1: return st_buoyancy(s);
        case 2: return st_wind(s);
        case 3: return st_force(s);
        case 4: return st_pressure(s, FS_ORDER_LEXICOGRAPHIC);
        case 5: return st_advect(s);
        case 6: return st_neumann(s);
        case 7: return st_zero(s);
        case 8: return st_divergence(s);
        case 9: return st_border(s);
        case 10: return st_poisson(s, FS_ORDER_LEXICOGRAPHIC);
        case 11: return st_correct(s);
        case 12: return st_vorticity(s);
        case 13: return st_reset_force(s);
        case 14: return st_pressure(s, FS_ORDER_RED_BLACK);
        case 15: return st_poisson(s, FS_ORDER_RED_BLACK);
    }
    s->err = "unknown stage";
    return FS_EINVAL;
}

int fs2d_advance(fs2d_sim* s) {   // FluidSim::advance, 3_fluid/FluidSim.h:71-96
    S2_GUARD(s);
    FS_TRY(st_source(s));
    FS_TRY(st_buoyancy(s));
    if (s->wind_on) FS_TRY(st_wind(s));
    FS_TRY(st_force(s));
    FS_TRY(st_pressure(s, s->ordering));
    FS_TRY(st_advect(s));
    FS_TRY(st_reset_force(s));
    s->time += s->dt;
    s->step++;
    return FS_OK;
}

}  // extern "C"
