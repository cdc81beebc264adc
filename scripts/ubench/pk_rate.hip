// Issue rate of packed against plain fp32 additions and multiplies on one wave64 per SIMD and on eight.
// Build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float c = out[0];
    const float2v c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // eight independent plain additions
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 1) {   // four independent packed additions: the same eight sums
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2));
        } else if (MODE == 2) {   // eight plain multiplies
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else {                  // four packed multiplies
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2));
        }
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
float run(float* d, int threads, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, (1 + 256 * 2048) * sizeof(float));
    hipMemset(d, 0, 4);
    const int iters = 200000;
    const char* names[4] = {"8 x v_add_f32   ", "4 x v_pk_add_f32", "8 x v_mul_f32   ", "4 x v_pk_mul_f32"};
    for (int waves = 1; waves <= 8; waves *= 2) {   // waves per SIMD: blocks of 256 threads = one wave per SIMD of a CU
        float ms[4] = {run<0>(d, 256, 256 * waves, iters), run<1>(d, 256, 256 * waves, iters), run<2>(d, 256, 256 * waves, iters),
                       run<3>(d, 256, 256 * waves, iters)};
        for (int m = 0; m < 4; ++m)
            printf("%d wave(s) per SIMD  %s per iteration: %.2f ms  -> %.2f cycles per iteration and wave at 2.4 GHz\n", waves, names[m],
                   ms[m], ms[m] * 1e-3 * 2.4e9 / iters / waves);
    }
    return 0;
}
