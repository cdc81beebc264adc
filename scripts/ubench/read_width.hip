// Read rate at 512^3 (three fields far larger than the Infinity Cache) by load width and by the number of waves a CU
// holds: 4-byte, 8-byte and 16-byte loads per lane; 32, 16 and 8 waves per CU (limited by a dummy LDS allocation).
// Build: hipcc --offload-arch=gfx950 -O3 read_width.hip -o read_width
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename V, int UNROLL>
__global__ void __launch_bounds__(512) k_read(const V* a, const V* b, const V* c, float* out, size_t n) {
    extern __shared__ float lds[];
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    float s = 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)blockDim.x * UNROLL) {
        V p[UNROLL], q[UNROLL], r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * blockDim.x;
            if (j < hi) { p[u] = a[j]; q[u] = b[j]; r[u] = c[j]; } else { p[u] = q[u] = r[u] = V(0); }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (sizeof(V) == 4) s += p[u] + q[u] + r[u];
            else if constexpr (sizeof(V) == 8) s += p[u].x + p[u].y + q[u].x + q[u].y + r[u].x + r[u].y;
            else s += p[u].x + p[u].y + p[u].z + p[u].w + q[u].x + q[u].y + q[u].z + q[u].w + r[u].x + r[u].y + r[u].z + r[u].w;
        }
    }
    if (s == 12345.678f) { out[0] = s; lds[threadIdx.x] = s; }
}

int main() {
    const int n = 512;
    const size_t cells = (size_t)n * n * n, bytes = cells * 4;
    float *f[3], *out;
    for (auto& p : f) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
    hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name) {
        launch(); launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %8.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, 3.0 * bytes / (ms / 10 * 1e-3) / 1e12);
        fflush(stdout);
    };
    for (int lds_kb : {16, 40, 76}) {   // 512-thread workgroups per CU: 4, 3 (by LDS), 2
        const int wgs = 160 / lds_kb > 4 ? 4 : 160 / lds_kb;
        const size_t lds = (size_t)lds_kb * 1024;
        char name[160];
#define RUN(V, U, label)                                                                                              \
        hipFuncSetAttribute((const void*)k_read<V, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        snprintf(name, sizeof name, "%d workgroups of 8 waves per CU  %s  %d loads per field in flight", wgs, label, U); \
        timeit([&] { hipLaunchKernelGGL((k_read<V, U>), dim3(16384), dim3(512), lds, 0, (const V*)f[0], (const V*)f[1], (const V*)f[2], out, bytes / sizeof(V)); }, name);
        RUN(float, 1, "4-byte loads ") RUN(float, 2, "4-byte loads ") RUN(float, 4, "4-byte loads ")
        RUN(f2, 1, "8-byte loads ") RUN(f2, 2, "8-byte loads ")
        RUN(f4, 1, "16-byte loads") RUN(f4, 2, "16-byte loads")
    }
    return 0;
}
