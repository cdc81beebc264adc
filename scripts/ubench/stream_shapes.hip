// Which SHAPE of a three-field stream the memory system serves best at 512^3, the question behind every kernel of
// the step that marches a tile along z: all of them move u, v, w from one generation of the arrays to the other,
// as a copy does, and the plain copies differ by 20 % among themselves (one 16-byte element per thread and no
// loop: 6.3 TB/s on a single array; a contiguous piece per workgroup: 5.4 TB/s).  Variants here, all on three
// fields of 512 MiB each (read + write = 3.2 GB):
//   once        one 16-byte element of each field per thread, no loop (consecutive workgroups = a compact front)
//   march       a workgroup owns a tile of ROWS rows x 512 columns and marches it through ZC planes, the shape of
//               the sweep kernels (tile rows contiguous: ROWS x 2 KiB per plane and field); workgroups numbered
//               tile-major (all tiles of a z-chunk, then the next chunk) as the kernels number them
//   march-pm    the same with short chunks numbered so that the launch advances plane by plane
//   march-1   the march on ONE array per generation whose rows alternate between the three fields
// Build: hipcc --offload-arch=gfx950 -O3 stream_shapes.hip -o stream_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void k_once(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ x,
                       f4* __restrict__ y, f4* __restrict__ z, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f4 p = a[i], q = b[i], r = c[i];
    if (NT) { __builtin_nontemporal_store(p, x + i); __builtin_nontemporal_store(q, y + i); __builtin_nontemporal_store(r, z + i); }
    else { x[i] = p; y[i] = q; z[i] = r; }
}

// tile of `rows` rows x 512 columns (128 float4 per row: two per lane, a wave per row), ZC planes per workgroup,
// loads of a plane issued one step before they are stored (as the kernels fetch ahead)
template <int NT>
__global__ void __launch_bounds__(1024) k_march(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                                  f4* __restrict__ x, f4* __restrict__ y, f4* __restrict__ z, int n, int rows,
                                                  int zc, int ntiles) {
    const int tile = blockIdx.x % ntiles, chunk = blockIdx.x / ntiles;
    const int row = tile * rows + threadIdx.y;
    if (row >= n) return;
    const size_t plane4 = (size_t)n * n / 4;
    size_t i = (size_t)chunk * zc * plane4 + (size_t)row * (n / 4) + threadIdx.x;
    const int zend = min(zc, n - chunk * zc);
    f4 p0 = a[i], p1 = a[i + 64], q0 = b[i], q1 = b[i + 64], r0 = c[i], r1 = c[i + 64];
    for (int k = 0; k < zend; ++k) {
        const size_t j = i + plane4;
        f4 np0 = p0, np1 = p1, nq0 = q0, nq1 = q1, nr0 = r0, nr1 = r1;
        if (k + 1 < zend) { np0 = a[j]; np1 = a[j + 64]; nq0 = b[j]; nq1 = b[j + 64]; nr0 = c[j]; nr1 = c[j + 64]; }
        if (NT) {
            __builtin_nontemporal_store(p0, x + i); __builtin_nontemporal_store(p1, x + i + 64);
            __builtin_nontemporal_store(q0, y + i); __builtin_nontemporal_store(q1, y + i + 64);
            __builtin_nontemporal_store(r0, z + i); __builtin_nontemporal_store(r1, z + i + 64);
        } else { x[i] = p0; x[i + 64] = p1; y[i] = q0; y[i + 64] = q1; z[i] = r0; z[i + 64] = r1; }
        p0 = np0; p1 = np1; q0 = nq0; q1 = nq1; r0 = nr0; r1 = nr1;
        i = j;
    }
}

// the same march on ONE array per generation in which the rows of the three fields alternate (row y of u, of v, of w,
// then row y + 1 of u ...): one input and one output stream, each tile row 3 x 2 KiB contiguous
template <int NT>
__global__ void __launch_bounds__(1024) k_march_rows_interleaved(const f4* __restrict__ a, f4* __restrict__ x, int n, int rows, int zc, int ntiles) {
    const int tile = blockIdx.x % ntiles, chunk = blockIdx.x / ntiles;
    const int row = tile * rows + threadIdx.y;
    if (row >= n) return;
    const size_t row4 = n / 4, plane4 = (size_t)n * 3 * row4;
    size_t i = (size_t)chunk * zc * plane4 + (size_t)row * 3 * row4 + threadIdx.x;
    const int zend = min(zc, n - chunk * zc);
    f4 p[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = a[i + k * 64];
    for (int kz = 0; kz < zend; ++kz) {
        const size_t j = i + plane4;
        f4 q[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = p[k];
        if (kz + 1 < zend) {
#pragma unroll
            for (int k = 0; k < 6; ++k) q[k] = a[j + k * 64];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { if (NT) __builtin_nontemporal_store(p[k], x + i + k * 64); else x[i + k * 64] = p[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) p[k] = q[k];
        i = j;
    }
}

int main() {
    const int n = 512;
    const size_t cells = (size_t)n * n * n, bytes = cells * 4, n4 = cells / 4;
    const size_t slack = 64u << 20;
    char* arena; hipMalloc(&arena, 6 * (bytes + slack)); hipMemset(arena, 0, 6 * (bytes + slack));
    f4* f[6];
    for (int k = 0; k < 6; ++k) f[k] = (f4*)(arena + k * (bytes + slack));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name) {
        launch(); launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %8.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, 6.0 * bytes / (ms / 10 * 1e-3) / 1e12);
        fflush(stdout);
    };
    char name[160];
    for (int threads : {256, 512, 1024}) {
        snprintf(name, sizeof name, "once, %4d threads per workgroup", threads);
        timeit([&] { hipLaunchKernelGGL((k_once<0>), dim3((unsigned)((n4 + threads - 1) / threads)), dim3(threads), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], n4); }, name);
        snprintf(name, sizeof name, "once, %4d threads per workgroup, non-temporal stores", threads);
        timeit([&] { hipLaunchKernelGGL((k_once<1>), dim3((unsigned)((n4 + threads - 1) / threads)), dim3(threads), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], n4); }, name);
    }
    for (int rows : {16, 8, 4}) {
        const int ntiles = (n + rows - 1) / rows;
        for (int zc : {512, 128, 64, 32, 16, 8, 4, 2, 1}) {
            const int chunks = (n + zc - 1) / zc;
            snprintf(name, sizeof name, "march, tile %2d rows x 512, %3d planes per workgroup (%6d workgroups)", rows, zc, ntiles * chunks);
            timeit([&] { hipLaunchKernelGGL((k_march<0>), dim3(ntiles * chunks), dim3(64, rows), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], n, rows, zc, ntiles); }, name);
            if (zc == 64 || zc == 8 || zc == 1) {
                snprintf(name, sizeof name, "march, tile %2d rows x 512, %3d planes per workgroup, non-temporal stores", rows, zc);
                timeit([&] { hipLaunchKernelGGL((k_march<1>), dim3(ntiles * chunks), dim3(64, rows), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], n, rows, zc, ntiles); }, name);
            }
        }
    }
    // the three fields as alternating rows of one array: f[0..2] and f[3..5] are adjacent in the arena (with slack between them,
    // which the interleaved kernels do not step over: they use the first 3 * bytes of each half)
    for (int rows : {16, 8}) {
        const int ntiles = (n + rows - 1) / rows;
        for (int zc : {128, 64, 32, 16}) {
            const int chunks = (n + zc - 1) / zc;
            snprintf(name, sizeof name, "march, rows of u, v, w alternating in ONE array, tile %2d rows, %3d planes per workgroup", rows, zc);
            timeit([&] { hipLaunchKernelGGL((k_march_rows_interleaved<0>), dim3(ntiles * chunks), dim3(64, rows), 0, 0, (const f4*)arena, (f4*)(arena + 3 * (bytes + slack)), n, rows, zc, ntiles); }, name);
            snprintf(name, sizeof name, "  the same, non-temporal stores");
            timeit([&] { hipLaunchKernelGGL((k_march_rows_interleaved<1>), dim3(ntiles * chunks), dim3(64, rows), 0, 0, (const f4*)arena, (f4*)(arena + 3 * (bytes + slack)), n, rows, zc, ntiles); }, name);
        }
    }
    hipFree(arena);
    return 0;
}
