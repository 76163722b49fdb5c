// Microbenchmark: what does an fp32 MFMA cost next to a dense fp32 VALU stream on gfx950?
// Question behind it: the gradient sums of the fit kernel are moments  sum_n u_nk * phi(x_n)  against
// block-independent pixel features; v_mfma_f32_* could take them off the VALU.  fp32 MFMA runs at the VALU's flop rate
// (MI355X_MICROARCH.md), so the gain would have to come from co-issue: matrix pipe busy while the VALU pipe works.
// Each kernel runs `iters` trips of a block of NV independent v_fma_f32 (8 chains) with NM MFMAs of one shape
// interleaved (independent accumulators), at 1..4 wavefronts per SIMD.  Reported: ns per trip per SIMD and the
// cycles the MFMAs added, per MFMA, at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE, int NM>     // SHAPE 0: none, 1: 4x4x1 (16 blocks), 2: 16x16x4, 3: 32x32x2
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    v4f acc4[4];
    v16f acc16[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {               // 8 x 8 = 64 v_fma per trip
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
            if (r < NM) {
                if (SHAPE == 1) acc4[r & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[r & 7], a, acc4[r & 3], 0, 0, 0);
                if (SHAPE == 2) acc4[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[r & 7], a, acc4[r & 3], 0, 0, 0);
                if (SHAPE == 3) acc16[r & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[r & 7], a, acc16[r & 1], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc16[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NM>
double run(int waves_per_simd, int iters) {
    const int cus = 256;
    const int blocks = cus * waves_per_simd;      // 256 threads = 4 wavefronts = one per SIMD
    float* d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, NM>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NM>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e6 / iters;                      // ns per trip (all resident wavefronts of a SIMD together)
}

int main() {
    const int iters = 20000;
    // warm the clocks
    for (int i = 0; i < 30; ++i) run<0, 0>(4, iters);
    const char* names[4] = {"none", "4x4x1 (16 blocks)", "16x16x4", "32x32x2"};
    printf("64 v_fma_f32 per trip and wavefront + NM fp32 MFMAs; ns per trip per SIMD; cycles at 2.4 GHz\n");
    for (int w = 1; w <= 4; ++w) {
        const double base = run<0, 0>(w, iters);
        printf("waves/SIMD=%d  VALU only: %8.1f ns/trip = %.2f cycles per v_fma\n", w, base, base * 2.4 / (64.0 * w));
        const double r[9] = {run<1, 2>(w, iters), run<1, 4>(w, iters), run<1, 8>(w, iters),
                             run<2, 2>(w, iters), run<2, 4>(w, iters), run<2, 8>(w, iters),
                             run<3, 2>(w, iters), run<3, 4>(w, iters), run<3, 8>(w, iters)};
        const int nm[3] = {2, 4, 8};
        for (int s = 0; s < 3; ++s)
            for (int q = 0; q < 3; ++q) {
                const double t = r[s * 3 + q];
                printf("   + %d x %-18s %8.1f ns/trip  (+%6.1f cycles per MFMA and wavefront; MFMA-only time would be %5.0f cycles)\n",
                       nm[q], names[s + 1], t, (t - base) * 2.4 / (nm[q] * w), (s == 0 ? 8.0 : (s == 1 ? 32.0 : 64.0)));
            }
    }
    return 0;
}
