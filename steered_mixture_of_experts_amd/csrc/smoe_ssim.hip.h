// smoe_ssim.hip.h -- the LDS SSIM stage shared by the per-block kernels (one block per wavefront) and the shared-kernel
// mode (one batch per workgroup).
#ifndef SMOE_SSIM_CUH
#define SMOE_SSIM_CUH

#include <hip/hip_runtime.h>

namespace smoe {

__device__ __forceinline__ void wave_lds_sync() {
    // LDS hand-off between lanes of ONE wavefront: DS ops of a wave execute in order,
    // so only compiler ordering + completion of the stores is required.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------
// SSIM loss stage (ssim_opt; smoe.py:980-1011 -> ops/image_ops_impl.py:77-233), 2-d blocks (3-d: ssim_block3 below).
// One wavefront works on one block-channel plane held in LDS.  The reference pads the block SYMMETRIC
// by 5 (smoe.py:993-996) and correlates with the 11x11 Gaussian (sigma 1.5, VALID): per axis that is
// the b x b matrix  T[i][j] = sum_a g[a] * [mirror(i + a - 5) == j]  (symmetric, band |i - j| <= 5,
// built on the host), so the window statistic is  Tr * plane * Tc  and its adjoint is the same
// product on the coefficient maps -- no padded copy, no scatter.
// ---------------------------------------------------------------------------
constexpr float SSIM_C1 = 0.0001f;    // (0.01 * max_val)^2   image_ops_impl.py:74,110
constexpr float SSIM_C2 = 0.0009f;    // (0.03 * max_val)^2   image_ops_impl.py:75,111

// Walk of the outputs n = lane, lane + NT, ... of one plane as (row i, column j) without a division per step.
// NT = threads that share one plane: 64 (one wavefront per block, wave-level hand-off through LDS) or the whole
// workgroup (shared-kernel mode, __syncthreads).
template <int NT>
struct SsimWalk {
    int i, j, di, dj;      // current position; per-step increments NT / bw and NT % bw
    __device__ __forceinline__ SsimWalk(int lane, int bw) : i(lane / bw), j(lane - (lane / bw) * bw), di(NT / bw), dj(NT % bw) {}
    __device__ __forceinline__ void next(int bw) {
        j += dj; i += di;
        if (j >= bw) { j -= bw; i += 1; }
    }
};

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// Tap tables are banded: Tb[i][a] = T[i][i + a - 5], zero where i + a - 5 leaves the axis, so a tap is
// weight * plane[clamp(i + a - 5)] with no compare / select.
// dst[p][i][j] = sum_r Tr[i][r] * f_p(r, j): window sums along axis 0 of x, x^2, x*y, y, y^2
template <int NT>
__device__ __forceinline__ void ssim_cols_products(float* __restrict__ dst, const float* __restrict__ xp,
                                                   const float* __restrict__ yp, const float* __restrict__ Trb,
                                                   int bh, int bw, int N, int lane) {
    SsimWalk<NT> w(lane, bw);
    for (int n = lane; n < N; n += NT, w.next(bw)) {
        const float* tw = Trb + w.i * 11;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = clampi(w.i + a - 5, bh - 1) * bw + w.j;
            const float wt = tw[a];
            const float xv = xp[o], yv = yp[o];
            const float wx = wt * xv, wy = wt * yv;
            s0 += wx;
            s1 = fmaf(wx, xv, s1);
            s2 = fmaf(wx, yv, s2);
            s3 += wy;
            s4 = fmaf(wy, yv, s4);
        }
        dst[n] = s0;
        dst[N + n] = s1;
        dst[2 * N + n] = s2;
        dst[3 * N + n] = s3;
        dst[4 * N + n] = s4;
    }
}

// dst[p][i][j] = sum_c Tc[j][c] * src[p][i][c]
template <int NP, int NT>
__device__ __forceinline__ void ssim_rows(float* __restrict__ dst, const float* __restrict__ src,
                                          const float* __restrict__ Tcb, int bw, int N, int lane) {
    SsimWalk<NT> w(lane, bw);
    for (int n = lane; n < N; n += NT, w.next(bw)) {
        const float* tw = Tcb + w.j * 11;
        const int row = w.i * bw;
        float s[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = row + clampi(w.j + a - 5, bw - 1);
            const float wt = tw[a];
#pragma unroll
            for (int p = 0; p < NP; ++p) s[p] = fmaf(wt, src[p * N + o], s[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p * N + n] = s[p];
    }
}

// Row pass of the five sums + the SSIM formula per window position (image_ops_impl.py:110-129).
// Returns this lane's sum of luminance * contrast-structure; with GRAD the three coefficient maps
// scale * d(l*cs)/d{mu_x, E[x^2], E[xy]} go to dst.
template <bool GRAD, int NT>
__device__ __forceinline__ float ssim_rows_stats(float* __restrict__ dst, const float* __restrict__ src,
                                                 const float* __restrict__ Tcb, int bw, int N, int lane, float scale) {
    float part = 0.0f;
    SsimWalk<NT> w(lane, bw);
    for (int n = lane; n < N; n += NT, w.next(bw)) {
        const float* tw = Tcb + w.j * 11;
        const int row = w.i * bw;
        float mx = 0.0f, sx = 0.0f, pxy = 0.0f, my = 0.0f, sy = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = row + clampi(w.j + a - 5, bw - 1);
            const float wt = tw[a];
            mx = fmaf(wt, src[o], mx);
            sx = fmaf(wt, src[N + o], sx);
            pxy = fmaf(wt, src[2 * N + o], pxy);
            my = fmaf(wt, src[3 * N + o], my);
            sy = fmaf(wt, src[4 * N + o], sy);
        }
        const float num0 = mx * my * 2.0f;
        const float den0 = mx * mx + my * my;
        const float N0 = num0 + SSIM_C1, D0 = den0 + SSIM_C1;
        const float N1 = (pxy * 2.0f - num0) + SSIM_C2;
        const float D1 = ((sx + sy) - den0) + SSIM_C2;
        const float r0 = __builtin_amdgcn_rcpf(D0), r1 = __builtin_amdgcn_rcpf(D1);      // 1 ulp: far inside the 2e-5 loss tolerance
        const float lum = N0 * r0, cs = N1 * r1;
        part = fmaf(lum, cs, part);
        if (GRAD) {
            const float dl = (2.0f * my - lum * (2.0f * mx)) * r0;             // d lum / d mu_x
            const float dc = (cs * (2.0f * mx) - 2.0f * my) * r1;              // d cs / d mu_x
            dst[n] = scale * fmaf(cs, dl, lum * dc);
            dst[N + n] = scale * (-(lum * cs) * r1);                            // d / d E[x^2]
            dst[2 * N + n] = scale * ((lum + lum) * r1);                        // d / d E[xy]
        }
    }
    return part;
}

// Column pass of the adjoint + assembly of dL/dq:  g = Ga + 2 x Gb + y Gc, written over x in place.
template <int NT>
__device__ __forceinline__ void ssim_cols_adjoint(float* __restrict__ xp, const float* __restrict__ yp,
                                                  const float* __restrict__ src, const float* __restrict__ Trb,
                                                  int bh, int bw, int N, int lane) {
    SsimWalk<NT> w(lane, bw);
    for (int n = lane; n < N; n += NT, w.next(bw)) {
        const float* tw = Trb + w.i * 11;
        float ga = 0.0f, gb = 0.0f, gc = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = clampi(w.i + a - 5, bh - 1) * bw + w.j;
            const float wt = tw[a];
            ga = fmaf(wt, src[o], ga);
            gb = fmaf(wt, src[N + o], gb);
            gc = fmaf(wt, src[2 * N + o], gc);
        }
        const float xv = xp[n];
        xp[n] = fmaf(yp[n], gc, fmaf(xv + xv, gb, ga));
    }
}

// 1 - SSIM of one block: returns this lane's share of -sum_c sw_c * mean(l * cs); with GRAD the plane
// X[c] is replaced by dL/dq.  sw[c] = channel weight / window count (kc.sw).
template <int NT>
__device__ __forceinline__ void ssim_sync() {
    if (NT == 64) wave_lds_sync();
    else __syncthreads();
}

template <int C, bool GRAD, int NT = 64>
__device__ __forceinline__ float ssim_block(float* __restrict__ X, const float* __restrict__ tgt,
                                            float* __restrict__ wa, float* __restrict__ wb,
                                            const float* __restrict__ Tr, const float* __restrict__ Tc,
                                            const float* __restrict__ sw, int bh, int bw, int N, int lane) {
    float part = 0.0f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        float* xp = X + c * N;
        const float* yp = tgt + c * N;
        ssim_cols_products<NT>(wa, xp, yp, Tr, bh, bw, N, lane);
        ssim_sync<NT>();
        const float swc = (c == 0) ? sw[0] : ((c == 1) ? sw[1] : sw[2]);   // no dynamic indexing of kernel arguments
        part -= swc * ssim_rows_stats<GRAD, NT>(wb, wa, Tc, bw, N, lane, -swc);
        if (GRAD) {
            ssim_sync<NT>();
            ssim_rows<3, NT>(wa, wb, Tc, bw, N, lane);
            ssim_sync<NT>();
            ssim_cols_adjoint<NT>(xp, yp, wa, Tr, bh, bw, N, lane);
        }
        ssim_sync<NT>();
    }
    return part;
}

// ---------------------------------------------------------------------------
// 3-d blocks [b0][b1][b2] (smoe.py:999-1003: SYMMETRIC pad by 5 on the three axes, custom_ssim(..., ndim=3) = conv3d with
// the 11x11x11 Gaussian, which is the product of the three 1-d windows): the same banded per-axis matrices, three axis
// passes forward (products along axis 0, then axis 1, then axis 2 + the SSIM formula) and three back.
// wa, wb: 5 planes each.
// ---------------------------------------------------------------------------
struct SsimPos3 { int i[3]; };
__device__ __forceinline__ SsimPos3 ssim_pos3(int n, int b1, int b2) {
    SsimPos3 p;
    const int s = b1 * b2;
    p.i[0] = n / s;
    const int r = n - p.i[0] * s;
    p.i[1] = r / b2;
    p.i[2] = r - p.i[1] * b2;
    return p;
}

// dst[p][n] = sum_a T_AX[i_AX][a] * src[p][n with i_AX -> clamp(i_AX + a - 5)]
template <int NP, int AX, int NT>
__device__ __forceinline__ void ssim3_axis(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ Tb,
                                           int b0, int b1, int b2, int N, int lane) {
    const int len = (AX == 0) ? b0 : ((AX == 1) ? b1 : b2);
    const int stride = (AX == 0) ? b1 * b2 : ((AX == 1) ? b2 : 1);
    for (int n = lane; n < N; n += NT) {
        const SsimPos3 q = ssim_pos3(n, b1, b2);
        const int ix = q.i[AX];
        const float* tw = Tb + ix * 11;
        float s[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = n + (clampi(ix + a - 5, len - 1) - ix) * stride;
            const float wt = tw[a];
#pragma unroll
            for (int p = 0; p < NP; ++p) s[p] = fmaf(wt, src[p * N + o], s[p]);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p * N + n] = s[p];
    }
}

// axis 0 with the products: window sums of x, x^2, x*y, y, y^2
template <int NT>
__device__ __forceinline__ void ssim3_products(float* __restrict__ dst, const float* __restrict__ xp, const float* __restrict__ yp,
                                               const float* __restrict__ T0, int b0, int b1, int b2, int N, int lane) {
    const int stride = b1 * b2;
    for (int n = lane; n < N; n += NT) {
        const int i0 = n / stride;
        const float* tw = T0 + i0 * 11;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = n + (clampi(i0 + a - 5, b0 - 1) - i0) * stride;
            const float wt = tw[a];
            const float xv = xp[o], yv = yp[o];
            const float wx = wt * xv, wy = wt * yv;
            s0 += wx;
            s1 = fmaf(wx, xv, s1);
            s2 = fmaf(wx, yv, s2);
            s3 += wy;
            s4 = fmaf(wy, yv, s4);
        }
        dst[n] = s0; dst[N + n] = s1; dst[2 * N + n] = s2; dst[3 * N + n] = s3; dst[4 * N + n] = s4;
    }
}

// axis 2 of the five sums + the SSIM formula per window position; GRAD: the three coefficient maps go to dst
template <bool GRAD, int NT>
__device__ __forceinline__ float ssim3_stats(float* __restrict__ dst, const float* __restrict__ src, const float* __restrict__ T2,
                                             int b1, int b2, int N, int lane, float scale) {
    float part = 0.0f;
    for (int n = lane; n < N; n += NT) {
        const int i2 = n % b2;
        const float* tw = T2 + i2 * 11;
        float mx = 0.0f, sx = 0.0f, pxy = 0.0f, my = 0.0f, sy = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = n + (clampi(i2 + a - 5, b2 - 1) - i2);
            const float wt = tw[a];
            mx = fmaf(wt, src[o], mx);
            sx = fmaf(wt, src[N + o], sx);
            pxy = fmaf(wt, src[2 * N + o], pxy);
            my = fmaf(wt, src[3 * N + o], my);
            sy = fmaf(wt, src[4 * N + o], sy);
        }
        const float num0 = mx * my * 2.0f;
        const float den0 = mx * mx + my * my;
        const float N0 = num0 + SSIM_C1, D0 = den0 + SSIM_C1;
        const float N1 = (pxy * 2.0f - num0) + SSIM_C2;
        const float D1 = ((sx + sy) - den0) + SSIM_C2;
        const float r0 = __builtin_amdgcn_rcpf(D0), r1 = __builtin_amdgcn_rcpf(D1);
        const float lum = N0 * r0, cs = N1 * r1;
        part = fmaf(lum, cs, part);
        if (GRAD) {
            const float dl = (2.0f * my - lum * (2.0f * mx)) * r0;
            const float dc = (cs * (2.0f * mx) - 2.0f * my) * r1;
            dst[n] = scale * fmaf(cs, dl, lum * dc);
            dst[N + n] = scale * (-(lum * cs) * r1);
            dst[2 * N + n] = scale * ((lum + lum) * r1);
        }
    }
    return part;
}

// axis 0 of the adjoint + assembly of dL/dq:  g = Ga + 2 x Gb + y Gc, written over x in place
template <int NT>
__device__ __forceinline__ void ssim3_adjoint0(float* __restrict__ xp, const float* __restrict__ yp, const float* __restrict__ src,
                                               const float* __restrict__ T0, int b0, int b1, int b2, int N, int lane) {
    const int stride = b1 * b2;
    for (int n = lane; n < N; n += NT) {
        const int i0 = n / stride;
        const float* tw = T0 + i0 * 11;
        float ga = 0.0f, gb = 0.0f, gc = 0.0f;
#pragma unroll
        for (int a = 0; a < 11; ++a) {
            const int o = n + (clampi(i0 + a - 5, b0 - 1) - i0) * stride;
            const float wt = tw[a];
            ga = fmaf(wt, src[o], ga);
            gb = fmaf(wt, src[N + o], gb);
            gc = fmaf(wt, src[2 * N + o], gc);
        }
        const float xv = xp[n];
        xp[n] = fmaf(yp[n], gc, fmaf(xv + xv, gb, ga));
    }
}

template <int C, bool GRAD, int NT = 64>
__device__ __forceinline__ float ssim_block3(float* __restrict__ X, const float* __restrict__ tgt,
                                             float* __restrict__ wa, float* __restrict__ wb,
                                             const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2,
                                             const float* __restrict__ sw, int b0, int b1, int b2, int N, int lane) {
    float part = 0.0f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        float* xp = X + c * N;
        const float* yp = tgt + c * N;
        ssim3_products<NT>(wa, xp, yp, T0, b0, b1, b2, N, lane);
        ssim_sync<NT>();
        ssim3_axis<5, 1, NT>(wb, wa, T1, b0, b1, b2, N, lane);
        ssim_sync<NT>();
        const float swc = (c == 0) ? sw[0] : ((c == 1) ? sw[1] : sw[2]);
        part -= swc * ssim3_stats<GRAD, NT>(wa, wb, T2, b1, b2, N, lane, -swc);
        if (GRAD) {
            ssim_sync<NT>();
            ssim3_axis<3, 2, NT>(wb, wa, T2, b0, b1, b2, N, lane);
            ssim_sync<NT>();
            ssim3_axis<3, 1, NT>(wa, wb, T1, b0, b1, b2, N, lane);
            ssim_sync<NT>();
            ssim3_adjoint0<NT>(xp, yp, wa, T0, b0, b1, b2, N, lane);
        }
        ssim_sync<NT>();
    }
    return part;
}

}  // namespace smoe
#endif
