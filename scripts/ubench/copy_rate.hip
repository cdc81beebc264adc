// What the memory system delivers to streaming kernels of the sweep's shape: three fields of n^3 floats read and
// three written (a copy), read only, written only.  Build: hipcc --offload-arch=gfx950 -O3 copy_rate.hip -o copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-result"

__global__ void k_copy3(const float4* a, const float4* b, const float4* c, float4* x, float4* y, float4* z, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        x[i] = a[i]; y[i] = b[i]; z[i] = c[i];
    }
}
__global__ void k_read3(const float4* a, const float4* b, const float4* c, float* out, size_t n4) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = a[i], q = b[i], r = c[i];
        s += p.x + p.y + p.z + p.w + q.x + q.y + q.z + q.w + r.x + r.y + r.z + r.w;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void k_write3(float4* x, float4* y, float4* z, size_t n4) {
    const float4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        x[i] = v; y[i] = v; z[i] = v;
    }
}

int main(int argc, char** argv) {
    for (int n : {256, 512}) {
        const size_t cells = (size_t)n * n * n, bytes = cells * 4, n4 = cells / 4;
        float4 *f[6];
        for (auto& p : f) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
        float* out; hipMalloc(&out, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blocks : {2048, 8192}) {
            auto timeit = [&](auto launch, const char* name, double bytes_moved) {
                launch(); launch();
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%d^3  %5d workgroups  %-28s %8.1f us  %6.2f TB/s\n", n, blocks, name, ms / 20 * 1e3, bytes_moved / (ms / 20 * 1e-3) / 1e12);
            };
            timeit([&] { hipLaunchKernelGGL(k_copy3, dim3(blocks), dim3(256), 0, 0, f[0], f[1], f[2], f[3], f[4], f[5], n4); }, "copy (3 read + 3 written)", 6.0 * bytes);
            timeit([&] { hipLaunchKernelGGL(k_read3, dim3(blocks), dim3(256), 0, 0, f[0], f[1], f[2], out, n4); }, "read 3", 3.0 * bytes);
            timeit([&] { hipLaunchKernelGGL(k_write3, dim3(blocks), dim3(256), 0, 0, f[3], f[4], f[5], n4); }, "write 3", 3.0 * bytes);
        }
        for (auto& p : f) hipFree(p);
    }
    return 0;
}
