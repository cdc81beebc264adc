// What the memory system gives a kernel with the STREAMS of the reflected advection at 512^3 and nothing else:
// seven fields read (the carried velocity, the density, the carrying velocity), four written, a byte of flags per
// cell, in the shape of k_advect_tiled - a workgroup of 64 x 8 threads owns a tile of 64 x 8 cells and marches it
// through a chunk of planes, fetching the next plane before it stores the current one.  Variants:
//   plain    every thread reads its own element of each field (7 x 4 + 1 B read, 16 B written per cell)
//   apron    the workgroup reads the tile widened by one cell all round, as the staging does (66 x 10 elements per
//            plane and field: 1.29 x the loads, the extra ones shared with the neighbouring tiles through L2), through
//            LDS with one barrier per plane; the stores stay one per thread
// and both with the workgroups renumbered per XCD as the kernel does.  The sum of the seven values is stored, so
// that nothing is optimised away; there is no other arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 advect_streams.hip -o advect_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"

struct Ptrs { const float* in[7]; float* out[4]; const uint8_t* flags; };

template <bool APRON, bool XCD>
__global__ void __launch_bounds__(512) k_streams(Ptrs P, int n, int zchunk) {
    constexpr int TX = 64, TY = 8, WX = TX + 2, WY = TY + 2, PS = WX * WY, NT = TX * TY;
    __shared__ float L[APRON ? 7 * 2 * PS : 1];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * TX + tx;
    int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int total = gridDim.x * gridDim.y * gridDim.z;
    if (XCD) { const int xcd = lin & 7, qq = total >> 3, rem = total & 7; lin = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (lin >> 3); }
    const int bx = lin % gridDim.x, by = (lin / gridDim.x) % gridDim.y, bz = lin / (gridDim.x * gridDim.y);
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * zchunk, z1 = min(z0 + zchunk, n);
    const size_t sz = (size_t)n * n;
    const size_t own = (size_t)(y0 + ty) * n + x0 + tx;
    size_t soff[2];
    for (int k = 0; k < 2; ++k) {
        const int e = min(tid + k * NT, PS - 1), ry = e / WX, rx = e - ry * WX;
        const int gx = min(max(x0 - 1 + rx, 0), n - 1), gy = min(max(y0 - 1 + ry, 0), n - 1);
        soff[k] = (size_t)gy * n + gx;
    }
    const bool second = tid + NT < PS;
    float st[7][2];
    auto fetch = [&](int z) {
        const size_t po = (size_t)min(z, n - 1) * sz;
        if (APRON) {
#pragma unroll
            for (int f = 0; f < 7; ++f) st[f][0] = P.in[f][po + soff[0]];
            if (second) {
#pragma unroll
                for (int f = 0; f < 7; ++f) st[f][1] = P.in[f][po + soff[1]];
            }
        } else {
#pragma unroll
            for (int f = 0; f < 7; ++f) st[f][0] = P.in[f][po + own];
        }
    };
    fetch(z0);
    for (int z = z0; z < z1; ++z) {
        const uint32_t fl = P.flags[(size_t)z * sz + own];
        float s;
        if (APRON) {
            float* dst = L + (z & 1) * 7 * PS + tid;
#pragma unroll
            for (int f = 0; f < 7; ++f) dst[f * PS] = st[f][0];
            if (second) {
#pragma unroll
                for (int f = 0; f < 7; ++f) dst[f * PS + NT] = st[f][1];
            }
            fetch(z + 1);
            __syncthreads();
            const float* src = L + (z & 1) * 7 * PS + (ty + 1) * WX + tx + 1;
            s = 0;
#pragma unroll
            for (int f = 0; f < 7; ++f) s += src[f * PS] + src[f * PS - 1] + src[f * PS + WX];
        } else {
            s = 0;
#pragma unroll
            for (int f = 0; f < 7; ++f) s += st[f][0];
            fetch(z + 1);
        }
        s += (float)fl;
        const size_t i = (size_t)z * sz + own;
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(s + k, P.out[k] + i);
    }
}

int main() {
    const int n = 512;
    const size_t cells = (size_t)n * n * n, bytes = cells * 4, slack = 64u << 20;
    char* arena; hipMalloc(&arena, 11 * (bytes + slack) + cells + slack); hipMemset(arena, 0, 11 * (bytes + slack) + cells + slack);
    Ptrs P;
    for (int k = 0; k < 7; ++k) P.in[k] = (const float*)(arena + k * (bytes + slack));
    for (int k = 0; k < 4; ++k) P.out[k] = (float*)(arena + (7 + k) * (bytes + slack));
    P.flags = (const uint8_t*)(arena + 11 * (bytes + slack));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name) {
        launch(); launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.1f us  %6.2f TB/s on 45 B per cell\n", name, ms / 10 * 1e3, 45.0 * cells / (ms / 10 * 1e-3) / 1e12);
        fflush(stdout);
    };
    char name[160];
    for (int zc : {512, 128, 64, 32, 16}) {
        dim3 g(n / 64, n / 8, (n + zc - 1) / zc), b(64, 8);
        snprintf(name, sizeof name, "plain, %3d planes per workgroup", zc);
        timeit([&] { hipLaunchKernelGGL((k_streams<false, false>), g, b, 0, 0, P, n, zc); }, name);
        snprintf(name, sizeof name, "plain, %3d planes per workgroup, per-XCD numbering", zc);
        timeit([&] { hipLaunchKernelGGL((k_streams<false, true>), g, b, 0, 0, P, n, zc); }, name);
        snprintf(name, sizeof name, "apron, %3d planes per workgroup", zc);
        timeit([&] { hipLaunchKernelGGL((k_streams<true, false>), g, b, 0, 0, P, n, zc); }, name);
        snprintf(name, sizeof name, "apron, %3d planes per workgroup, per-XCD numbering", zc);
        timeit([&] { hipLaunchKernelGGL((k_streams<true, true>), g, b, 0, 0, P, n, zc); }, name);
    }
    hipFree(arena);
    return 0;
}
