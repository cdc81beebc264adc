// Issue rate of the vector instructions the advection kernel is made of (fp32 arithmetic, conversions, floor,
// integer adds and logic, 24-bit multiply-add, compares, selects) with eight waves per SIMD: eight independent
// instructions of one kind per loop iteration.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"

#define REP8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int MODE>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = out[0];
    for (int i = 0; i < iters; ++i) {
#define I_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_CVT(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_MIN(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_CMP64(n) "v_cmp_lt_u32 s[20:21], %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_SUBREV(n) "v_subrev_u32 %" #n ", s20, %" #n "\n"
#define I_CND64(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_ADDFS(n) "v_add_f32 %" #n ", s20, %" #n "\n"
#define I_MULFS(n) "v_mul_f32 %" #n ", s20, %" #n "\n"
#define I_SUB1(n) "v_sub_f32 %" #n ", 1.0, %" #n "\n"
#define I_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %8\n"
#define I_BFE(n) "v_bfe_i32 %" #n ", %" #n ", 6, 1\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_CLASS(n) "v_cmp_class_f32 vcc, %" #n ", %8\n"
#define I_SUBU(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 2, %" #n "\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_ADDUS(n) "v_add_u32 %" #n ", s20, %" #n "\n"
#define I_CMPF(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_ADDUK(n) "v_add_u32 %" #n ", 0x2940, %" #n "\n"
#define I_CND64V(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define I_CNDC(n) "v_cndmask_b32 %" #n ", 0, %" #n ", vcc\n"
#define I_CNDSDWA(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
        if (MODE == 0) asm volatile(REP8(I_ADDF) : OPS : "v"(c));
        else if (MODE == 1) asm volatile(REP8(I_ADDU) : OPS : "v"(c));
        else if (MODE == 2) asm volatile(REP8(I_AND) : OPS : "v"(c));
        else if (MODE == 3) asm volatile(REP8(I_CVT) : OPS : "v"(c));
        else if (MODE == 4) asm volatile(REP8(I_FLOOR) : OPS : "v"(c));
        else if (MODE == 5) asm volatile(REP8(I_MAD24) : OPS : "v"(c));
        else if (MODE == 6) asm volatile(REP8(I_ADD3) : OPS : "v"(c));
        else if (MODE == 7) asm volatile(REP8(I_LSHLADD) : OPS : "v"(c));
        else if (MODE == 8) asm volatile(REP8(I_MIN) : OPS : "v"(c));
        else if (MODE == 9) asm volatile(REP8(I_CND) : OPS : "v"(c) : "vcc");
        else if (MODE == 10) asm volatile(REP8(I_CMP) : OPS : "v"(c) : "vcc");
        else if (MODE == 11) asm volatile(REP8(I_CMP64) : OPS : "v"(c) : "s20", "s21");
        else if (MODE == 12) asm volatile(REP8(I_FMA) : OPS : "v"(c));
        else if (MODE == 13) asm volatile(REP8(I_MOV) : OPS : "v"(c));
        else if (MODE == 14) asm volatile(REP8(I_SUBREV) : OPS : "v"(c) : "s20");
        else if (MODE == 15) asm volatile(REP8(I_CND64) : OPS : "v"(c) : "s20", "s21");
        else if (MODE == 16) asm volatile(REP8(I_ADDFS) : OPS : "v"(c) : "s20");
        else if (MODE == 17) asm volatile(REP8(I_MULFS) : OPS : "v"(c) : "s20");
        else if (MODE == 18) asm volatile(REP8(I_SUB1) : OPS : "v"(c));
        else if (MODE == 19) asm volatile(REP8(I_BFI) : OPS : "v"(c));
        else if (MODE == 20) asm volatile(REP8(I_BFE) : OPS : "v"(c));
        else if (MODE == 21) asm volatile(REP8(I_CLASS) : OPS : "v"(c) : "vcc");
        else if (MODE == 22) asm volatile(REP8(I_SUBU) : OPS : "v"(c));
        else if (MODE == 23) asm volatile(REP8(I_MINF) : OPS : "v"(c));
        else if (MODE == 24) asm volatile(REP8(I_LSHL) : OPS : "v"(c));
        else if (MODE == 25) asm volatile(REP8(I_MUL24) : OPS : "v"(c));
        else if (MODE == 26) asm volatile(REP8(I_ADDUS) : OPS : "v"(c) : "s20");
        else if (MODE == 27) asm volatile(REP8(I_CMPF) : OPS : "v"(c) : "vcc");
        else if (MODE == 28) asm volatile(REP8(I_ADDUK) : OPS : "v"(c));
        else if (MODE == 29) asm volatile(REP8(I_CND64V) : OPS : "v"(c) : "vcc");
        else if (MODE == 30) asm volatile(REP8(I_CNDC) : OPS : "v"(c) : "vcc");
        else if (MODE == 31) asm volatile(REP8(I_CNDSDWA) : OPS : "v"(c) : "vcc");
    }
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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

template <int M>
void all(float* d, int waves, int iters, const char* const* names) {
    if constexpr (M < 32) {
        const float ms = run<M>(d, 256, 256 * waves, iters);
        printf("%d waves per SIMD  8 x %-30s %7.2f ms -> %.2f ns per wave instruction and SIMD\n", waves, names[M], ms, ms * 1e6 / iters / waves / 8);
        all<M + 1>(d, waves, iters, names);
    }
}

int main() {
    float* d;
    hipMalloc(&d, (1 + 256 * 2048) * sizeof(float));
    hipMemset(d, 0, 4);
    const int iters = 100000;
    const char* names[32] = {"v_add_f32", "v_add_u32", "v_and_b32", "v_cvt_i32_f32", "v_floor_f32", "v_mad_u32_u24", "v_add3_u32",
                             "v_lshl_add_u32", "v_min_u32", "v_cndmask_b32 (vcc)", "v_cmp_lt_u32 -> vcc", "v_cmp_lt_u32 -> sgpr pair",
                             "v_fma_f32", "v_mov_b32", "v_subrev_u32 (sgpr operand)", "v_cndmask_b32 (sgpr pair)", "v_add_f32 (sgpr operand)",
                             "v_mul_f32 (sgpr operand)", "v_sub_f32 1.0 - x", "v_bfi_b32", "v_bfe_i32", "v_cmp_class_f32 -> vcc", "v_sub_u32",
                             "v_min_f32", "v_lshlrev_b32", "v_mul_u32_u24", "v_add_u32 (sgpr operand)", "v_cmp_lt_f32 -> vcc",
                             "v_add_u32 (32-bit literal)", "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32 0, v (vcc)", "v_cndmask_b32 c, v (vcc)"};
    for (int waves : {6, 8}) all<0>(d, waves, iters, names);
    return 0;
}
