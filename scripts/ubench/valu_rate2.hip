// Microbenchmark 2: issue cost of assorted VALU instructions on gfx950 (2 and 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long m = __ballot(threadIdx.x & 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) asm volatile(
#define S(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
                REP8(S)
#undef S
                : OPS : "v"(a) : "vcc");
            if (MODE == 1) asm volatile(
#define S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n"
                REP8(S)
#undef S
                : OPS : "v"(a), "s"(m));
            if (MODE == 2) asm volatile(
#define S(i) "v_mov_b32 %" #i ", %8\n"
                REP8(S)
#undef S
                : OPS : "v"(a));
            if (MODE == 3) asm volatile(
#define S(i) "v_add_f32 %" #i ", %" #i ", %8\n"
                REP8(S)
#undef S
                : OPS : "v"(a));
            if (MODE == 4) asm volatile(
#define S(i) "v_max_f32 %" #i ", %" #i ", %8\n"
                REP8(S)
#undef S
                : OPS : "v"(a));
            if (MODE == 5) asm volatile(
#define S(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
                REP8(S)
#undef S
                : OPS : "v"(a) : "vcc");
            if (MODE == 6) asm volatile(
#define S(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n s_nop 1\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
                REP8(S)
#undef S
                : OPS : "v"(a) : "vcc");
            if (MODE == 7) asm volatile(
#define S(i) "v_floor_f32 %" #i ", %" #i "\n"
                REP8(S)
#undef S
                : OPS);
            if (MODE == 8) asm volatile(
#define S(i) "v_rcp_f32 %" #i ", %" #i "\n"
                REP8(S)
#undef S
                : OPS);
            if (MODE == 9) asm volatile(
#define S(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
                REP8(S)
#undef S
                : OPS : "v"(a), "v"(b));
            if (MODE == 10) asm volatile(
#define S(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
                REP8(S)
#undef S
                : OPS : "v"(a), "s"(b));
            if (MODE == 11) asm volatile(
#define S(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
                REP8(S)
#undef S
                : OPS : "v"(a));
            if (MODE == 12) asm volatile(
#define S(i) "v_fmac_f32 %" #i ", %8, %9\n"
                REP8(S)
#undef S
                : OPS : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
void run(const char* name, int per) {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 10000;
    for (int wps : {2, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)iters * 64.0 * wps * per;
        printf("%-28s waves/SIMD=%d  %.3f ms  nominal cycles/instr/SIMD=%.2f\n", name, wps, ms, ms * 1e-3 * 2.4e9 / instr);
    }
    (void)hipFree(out);
}
int main() {
    run<10>("v_fma_f32 (sgpr operand)", 1);
    run<12>("v_fmac_f32", 1);
    run<11>("v_mul_f32", 1);
    run<3>("v_add_f32", 1);
    run<4>("v_max_f32", 1);
    run<9>("v_med3_f32", 1);
    run<2>("v_mov_b32", 1);
    run<0>("v_cndmask vcc", 1);
    run<1>("v_cndmask_e64 sgpr", 1);
    run<5>("v_cmp_lt_f32 vcc", 1);
    run<6>("v_cmp + nop + cndmask (x2)", 2);
    run<7>("v_floor_f32", 1);
    run<8>("v_rcp_f32", 1);
    return 0;
}
