// smoe_kernels.hip -- small per-block kernels (kernel-list readmission, best snapshot, scalar reduction) and the
// dispatch table over the per-(D, C, K) instantiations of smoe_block.hip.h (smoe_var_*.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smoe_block.hip.h"

namespace smoe {
// ---------------------------------------------------------------------------
// small per-block kernels
// ---------------------------------------------------------------------------
// update_kernel_list, smoe.py:2287-2365 (probe test smoe.py:806)
template <int D>
__global__ void readmit_kernel(ReadmitArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B * a.K) return;
    const int b = t / a.K;
    const int k = t - b * a.K;
    const long bk = (long)b * a.K + k;
    float A[D][D];
#pragma unroll
    for (int l = 0; l < D; ++l)
#pragma unroll
        for (int m = 0; m < D; ++m)
            A[l][m] = (l == m) ? a.p.A_diagonal[(bk * D + l) * D + m] : ((l > m) ? a.p.A_corr[(bk * D + l) * D + m] : 0.0f);
    int nprobe = 1;
#pragma unroll
    for (int l = 0; l < D; ++l) nprobe *= 3;
    bool near = false;
    for (int q = 0; q < nprobe; ++q) {
        float r[D];
        int rem = q;
#pragma unroll
        for (int l = D - 1; l >= 0; --l) {       // itertools.product order (last axis fastest); order is irrelevant to any()
            const int sel = rem % 3;
            rem /= 3;
            r[l] = a.probes[l * 3 + sel] - a.p.musX[bk * D + l];
        }
        float maha = 0.0f;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float zz = 0.0f;
            if (a.inverse_cov) {                      // r^T A r with the symmetric A (smoe.py:791-793)
#pragma unroll
                for (int l = 0; l < D; ++l) zz = fmaf(r[l], (l >= m) ? A[l][m] : A[m][l], zz);
                maha = fmaf(zz, r[m], maha);
            } else {
#pragma unroll
                for (int l = m; l < D; ++l) zz = fmaf(r[l], A[l][m], zz);
                maha = fmaf(zz, zz, maha);
            }
        }
        near = near || (maha < 800.0f);
    }
    if (near && a.p.pis[bk] > 0.0f) atomicOr(&a.active[b], 1u << k);
}

// checkpoint_best_op, smoe.py:861-896 (trigger 1574-1576), per block
__global__ void best_kernel(BestArgs a) {
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const bool better = a.loss[b] < a.best_loss[b];
    if (!better) return;
    const int K = a.K, D = a.D, C = a.C;
    for (int i = threadIdx.x; i < K; i += blockDim.x) a.best.pis[(long)b * K + i] = a.p.pis[(long)b * K + i];
    for (int i = threadIdx.x; i < K * D; i += blockDim.x) a.best.musX[(long)b * K * D + i] = a.p.musX[(long)b * K * D + i];
    for (int i = threadIdx.x; i < K * D * D; i += blockDim.x) {
        a.best.A_diagonal[(long)b * K * D * D + i] = a.p.A_diagonal[(long)b * K * D * D + i];
        a.best.A_corr[(long)b * K * D * D + i] = a.p.A_corr[(long)b * K * D * D + i];
    }
    for (int i = threadIdx.x; i < K * D * C; i += blockDim.x) a.best.gamma_e[(long)b * K * D * C + i] = a.p.gamma_e[(long)b * K * D * C + i];
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) a.best.nu_e[(long)b * K * C + i] = a.p.nu_e[(long)b * K * C + i];
    __syncthreads();
    if (threadIdx.x == 0) a.best_loss[b] = a.loss[b];
}

// host accumulation of smoe.py:1758-1761 -> three doubles.  Two stages with a fixed assignment of blocks
// to threads and a fixed tree order, so the result is deterministic: REDUCE_WGS workgroups write partial
// sums, one workgroup combines them (a single workgroup over 65 536 blocks took 190 us of load latency).
constexpr int REDUCE_WGS = 64;

__device__ __forceinline__ void reduce3_tree(double (&s)[3][256], int n) {
    for (int w = n / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s[0][threadIdx.x] += s[0][threadIdx.x + w];
            s[1][threadIdx.x] += s[1][threadIdx.x + w];
            s[2][threadIdx.x] += s[2][threadIdx.x + w];
        }
        __syncthreads();
    }
}

__global__ void reduce_scalars_kernel(ReduceArgs a) {
    __shared__ double s[3][256];
    double l = 0.0, e = 0.0, c = 0.0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < a.B; b += REDUCE_WGS * 256) {
        if (a.loss) l += (double)a.loss[b] * (double)a.N;
        if (a.sse) e += (double)a.sse[b];
        if (a.active) c += (double)__popc(a.active[b]);
    }
    s[0][threadIdx.x] = l; s[1][threadIdx.x] = e; s[2][threadIdx.x] = c;
    __syncthreads();
    reduce3_tree(s, 256);
    if (threadIdx.x == 0) {
        a.partials[blockIdx.x * 3 + 0] = s[0][0];
        a.partials[blockIdx.x * 3 + 1] = s[1][0];
        a.partials[blockIdx.x * 3 + 2] = s[2][0];
    }
}

__global__ void reduce_partials_kernel(ReduceArgs a) {
    __shared__ double s[3][256];
    const bool in = (int)threadIdx.x < REDUCE_WGS;
    s[0][threadIdx.x] = in ? a.partials[threadIdx.x * 3 + 0] : 0.0;
    s[1][threadIdx.x] = in ? a.partials[threadIdx.x * 3 + 1] : 0.0;
    s[2][threadIdx.x] = in ? a.partials[threadIdx.x * 3 + 2] : 0.0;
    __syncthreads();
    reduce3_tree(s, REDUCE_WGS);
    if (threadIdx.x == 0) { a.out[0] = s[0][0]; a.out[1] = s[1][0]; a.out[2] = s[2][0]; }
}

// one table per line of smoe_variants.def (smoe_var.hip)
#define SMOE_TRIPLE(D, C, K, FULL) const Variant* variants_d##D##c##C##k##K(int* count);
#include "smoe_variants.def"
#undef SMOE_TRIPLE

const Variant* variants(int* count) {
    static Variant table[256];
    static int n = 0;
    if (n == 0) {
        int m = 0;
#define SMOE_TRIPLE(D, C, K, FULL)                                   \
        {                                                            \
            int c = 0;                                               \
            const Variant* part = variants_d##D##c##C##k##K(&c);     \
            for (int i = 0; i < c && m < 256; ++i) table[m++] = part[i]; \
        }
#include "smoe_variants.def"
#undef SMOE_TRIPLE
        n = m;
    }
    *count = n;
    return table;
}

hipError_t launch_readmit(const ReadmitArgs& a, int D, hipStream_t st) {
    const int threads = 256;
    const int grid = (a.B * a.K + threads - 1) / threads;
    if (D == 2) hipLaunchKernelGGL(readmit_kernel<2>, dim3(grid), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL(readmit_kernel<3>, dim3(grid), dim3(threads), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_best(const BestArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(best_kernel, dim3(a.B), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(reduce_scalars_kernel, dim3(REDUCE_WGS), dim3(256), 0, st, a);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(REDUCE_WGS), 0, st, a);
    return hipGetLastError();
}

int reduce_partials_count() { return REDUCE_WGS * 3; }

}  // namespace smoe
