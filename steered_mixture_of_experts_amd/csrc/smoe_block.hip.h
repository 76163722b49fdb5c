// smoe_block.hip.h -- CDNA4 (gfx950) device templates of the per-block SMoE hot path: packed parameter layout, the
// per-pixel forward / backward, the LDS tile, the fit and forward kernels with their SSIM and fake-quant
// variants, and the launcher templates.  Included by one translation unit per (D, C, K) instantiation
// (smoe_var_*.hip), which are compiled in parallel, and by smoe_kernels.hip (small kernels + the dispatch table).
//
// What one workgroup does: WAVES wavefronts, each wavefront owns 64/G image blocks
// (G lanes per block, N/G pixels per lane).  The block's parameters live in LDS
// (broadcast reads at the top of every iteration) and in registers during the pixel
// loop; targets / coordinates are staged once per launch into LDS with coalesced HBM
// reads; the K*P gradient sums are reduced across the G lanes through an LDS transpose
// (every lane finishes ONE sum per pass: "owner" lanes), the owner applies TF1-Adam to
// its parameter and publishes it back to LDS.  No inter-workgroup communication.
//
// Maths: SURVEY.md Appendix A; reference lines are cited at each step
// (paths relative to /root/reference).
#ifndef SMOE_BLOCK_HIP_H
#define SMOE_BLOCK_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "smoe_device.h"
#include "smoe_ssim.hip.h"
#include "smoe_fq.hip.h"

namespace smoe {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Fair share of a SIMD for the wavefronts of a ONE-ROUND launch (FitArgs::prio_rotate, set by the host when every wavefront of
// the launch is resident at once).  The instruction arbiter serves the oldest wavefront first: of the two or three wavefronts
// that start a launch together on a SIMD the first runs at nearly the lone-wavefront rate and the last finishes ~50 % later,
// and the launch lasts as long as its slowest wavefront (12 288 blocks of 16x16: three per SIMD, 0.915 ms per 100 iterations,
// while every FURTHER 768 workgroups of a longer launch cost 0.75 ms).  Rotating the user priority (s_setprio beats age) by one
// level per iteration, offset by the hardware wave slot, gives every wavefront the same share: 12 288 blocks 336 -> 368
// Gpx-it/s, 8 192: 302 -> 329, 4 096 (32 lanes): 235 -> 257, 2 048 (64 lanes): 166 -> 176.  In a launch of several rounds the
// unfairness is useful (early finishers make room for the next round: 16 384 blocks lose 2 % with the rotation): not used there.
__device__ __forceinline__ uint32_t hw_wave_slot() {
    return __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));          // HW_REG_HW_ID, WAVE_ID[3:0]: the wave slot on its SIMD
}
__device__ __forceinline__ void rotate_priority(int it, uint32_t slot) {
    const uint32_t pr = ((uint32_t)it + slot) % 3u;
    if (pr == 0u) __builtin_amdgcn_s_setprio(0);
    else if (pr == 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(2);
}
// (the slot is read from the hardware register every iteration: kept in a scalar register across the iteration loop it cost the
// headline kernel, which sits at the scalar-register limit, two VALU instructions in its pixel loop: -1 %)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
constexpr int tri_index(int l, int m) { return l * (l + 1) / 2 + m; }   // l >= m

// Packed per-block parameter vector: for every kernel k
//   [ pi | mu[0..D) | A lower-tri row-major (l>=m) | nu[0..C) | gamma[l][c] ]
// Gradient sums use the same indexing ("slot j <-> packed parameter j"); three kinds
// of extra slots follow: loss, sse, and the K influence counters (smoe.py:829).
template <int D, int C, int K>
struct Layout {
    static constexpr int TRI = D * (D + 1) / 2;
    static constexpr int O_PI = 0;
    static constexpr int O_MU = 1;
    static constexpr int O_A = 1 + D;
    static constexpr int O_NU = O_A + TRI;
    static constexpr int O_GA = O_NU + C;
    static constexpr int PK = O_GA + D * C;
    static constexpr int NPAR = K * PK;
    static constexpr int S_LOSS = NPAR;
    static constexpr int S_SSE = NPAR + 1;
    static constexpr int S_CNT = NPAR + 2;
    static constexpr int NSLOT = NPAR + 2 + K;
    // LDS image of one block: packed params, K active flags, frozen flag, K fake-quantised pis (quantize_pis)
    static constexpr int LP_ACT = NPAR;
    static constexpr int LP_FROZEN = NPAR + K;
    static constexpr int LP_QPI = NPAR + K + 1;
    static constexpr int LP_ZERO = NPAR + 2 * K + 1;    // constant cells 0.0f / 1.0f: operands of the unused terms of a
    static constexpr int LP_ONE = NPAR + 2 * K + 2;     // slot's gradient descriptor (owner_gradient)
    static constexpr int LP_STRIDE = round_up(NPAR + 2 * K + 3, 4);
};

// Where packed parameter j of block b lives in the reference's tensors
// (get_params layout, smoe.py:1795-1800).  tensor: 0 pis 1 musX 2 A_diagonal 3 A_corr
// 4 gamma_e 5 nu_e.
template <int D, int C, int K>
__device__ __forceinline__ void decode_slot(int j, int b, int& tensor, long& off, int& kern) {
    using Lt = Layout<D, C, K>;
    // branch-free on purpose (selects instead of a chain of divergent branches): every owner runs this for each of its
    // slots in the prologue and the epilogue of a launch
    const int k = j / Lt::PK;
    const int o = j - k * Lt::PK;
    const long bk = (long)b * K + k;
    kern = k;
    const int t = o - Lt::O_A;                                        // steering entry: lower triangle, row-major
    const int l = (t >= 3) ? 2 : ((t >= 1) ? 1 : 0);
    const int m = t - l * (l + 1) / 2;
    const bool is_pi = o == Lt::O_PI, is_mu = o < Lt::O_A, is_a = o < Lt::O_NU, is_nu = o < Lt::O_GA;
    tensor = is_pi ? 0 : (is_mu ? 1 : (is_a ? ((l == m) ? 2 : 3) : (is_nu ? 5 : 4)));
    off = is_pi ? bk
        : (is_mu ? bk * D + (o - Lt::O_MU)
        : (is_a ? (bk * D + l) * D + m
        : (is_nu ? bk * C + (o - Lt::O_NU)
        : bk * (D * C) + (o - Lt::O_GA))));
}

__device__ __forceinline__ float* pick(const smoe_params& s, int tensor) {
    switch (tensor) {
        case 0: return s.pis;
        case 1: return s.musX;
        case 2: return s.A_diagonal;
        case 3: return s.A_corr;
        case 4: return s.gamma_e;
        default: return s.nu_e;
    }
}

// ---------------------------------------------------------------------------
// per-pixel forward (+ optional backward accumulation)
// ---------------------------------------------------------------------------
// sqrt(0.5 * log2(e)): with A' = SQ * A, |A'^T r|^2 = maha * 0.5*log2(e), so
// exp(-maha/2) = exp2(-|z'|^2) and the per-kernel scale multiply disappears.
// 64-lane tiling: cross-lane reduction in registers (reduce_slots_regs) instead of the LDS transpose.  Built, parity-green
// (224 GPU tests) and measured SLOWER (scripts/pair_check.py, one session): ONE image 0.303 -> 0.328 ms per 100 iterations,
// its two-wavefront form 0.282 -> 0.294, 1 020 blocks of 16x16x4 RGB 1.13 -> 1.27 ms; only 32x32 / K = 8 gained (1.39 ->
// 1.34 ms).  The swaps and the four dependent DPP rotations per slot cost more than the 17 LDS waits they remove.  Off;
// make EXTRA=-DSMOE_REGRED=1 builds it.
#ifndef SMOE_REGRED
#define SMOE_REGRED 0
#endif
// fit_min_waves: triples with this many slots are bound to two wavefronts per SIMD on 64 lanes.  Window from the table
// profiles/r02/bench_rich_triples.txt (2 040 blocks, bound vs unbound, Gpx-it/s): 32x32 RGB K=8 (130 slots) 102 vs 85,
// 32x32 K=12 (122) 111 vs 80, 32x32 RGB K=9 (143) 84 vs 66, 16x16x4 K=8 (122) 141 vs 99, 16x16x4 RGB K=6 (140) 118 vs 84,
// 32x32 RGB K=6 (98) 136 vs 134; above the window the parked values no longer fit beside the pixel loop:
// 16x16x4 RGB K=8 (186 slots) 58 vs 66, 32x32 K=16 (162) 20 vs 54.
#ifndef SMOE_W2_SLOTS
#define SMOE_W2_SLOTS 96
#endif
// the same bound on the 16-lane hoisting kernels of the triples with 90..150 slots (large batches of small blocks with many
// kernels: 32 768 blocks of 16x16 K=12 76 -> 112 Gpx-it/s, 8x8x4 K=6 104 -> 164, 8x8x4 RGB K=4 94 -> 111; none slower:
// profiles/r02/bench_rich_triples.txt)
#ifndef SMOE_W2_G16
#define SMOE_W2_G16 1
#endif
// ... and on the 32-lane ones (mid-size batches: 4 096 blocks of 16x16 RGB K=8 54.6 -> 73.8 Gpx-it/s, 3 100: 41.8 -> 57.1)
#ifndef SMOE_W2_G32
#define SMOE_W2_G32 1
#endif
#ifndef SMOE_W2_SLOTS_MAX
#define SMOE_W2_SLOTS_MAX 150
#endif
#ifndef SMOE_NT_STORES
#define SMOE_NT_STORES 1
#endif
// one block per wavefront, parameter-rich triples (K * C >= 24): the wave-uniform constants of the pixel loop in scalar registers
// (32x32 / K = 8 / RGB bound to 256 VGPRs: 61 -> 30 parked dwords, 110.8 -> 113.6 Gpx-it/s)
#ifndef SMOE_SGPR_CONSTS
#define SMOE_SGPR_CONSTS 1
#endif
// Diagnostic build (make EXTRA=-DSMOE_PHASE_CLOCKS=1; scripts/phase_clocks.py): lane 0 of every wavefront of workgroup 0 sums the
// shader-clock cycles it spends in each phase of the fit iteration and leaves them in loss_out[wave * 8 + phase] -- the
// loss outputs of the first blocks are garbage in such a build.
#ifndef SMOE_PHASE_CLOCKS
#define SMOE_PHASE_CLOCKS 0
#endif
#if SMOE_PHASE_CLOCKS
#define SMOE_CLK(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); clk[i] += (float)(_t - clk_last); clk_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMOE_CLK(i) do { } while (0)
#endif
// Debug build (make EXTRA=-DSMOE_DEBUG=1): every kernel compares the end of the LDS carve-up IT will address -- from its own
// template flags and run-time arguments -- with the dynamic LDS the launcher reserved; a mismatch sets a bit of FitArgs::dbg /
// FwdArgs::dbg and the workgroup returns before touching LDS (no fault: a faulting kernel can reset the GPUs of the host for
// everyone on it).  The host layer reads the word after every launch of such a build and fails the call.
#ifndef SMOE_DEBUG
#define SMOE_DEBUG 0
#endif
#if SMOE_DEBUG
#define SMOE_LDS_CHECK(end_floats, code)                                                   \
    do {                                                                                   \
        if ((long)(end_floats) > (long)a.lds_floats) {                                     \
            if (threadIdx.x == 0 && a.dbg != nullptr) atomicOr(a.dbg, (uint32_t)(code));   \
            return;                                                                        \
        }                                                                                  \
    } while (0)
#else
#define SMOE_LDS_CHECK(end_floats, code) do { } while (0)
#endif
#define SMOE_SQ 0.84932180028801904272f
#define SMOE_INV_SQ 1.17740022503374817543f

template <int D, int C, int K>
struct BlockRegs {
    using Lt = Layout<D, C, K>;
    float P[Lt::LP_STRIDE];   // packed params + flags, filled from LDS
    float As[K][Lt::TRI];     // A' = SQ * A (lower triangle)
    float cz[K][D];           // c = A'^T mu, so z' = A'^T x - c
    float coef[K];            // pi * prod diag(A) / sqrt((2pi)^d), 0 when the kernel is inactive
    // hoisting (HL = number of trailing coordinates that are the same for all pixels of the lane):
    float hz[K][D];           // sum_{l>=D-HL} x_l A'[l][m] - c[m]   (z' = sum_{l<D-HL} x_l A'[l][m] + hz[m])
    float he[K][C];           // nu + sum_{l>=D-HL} gamma[l] x_l      (e  = he + sum_{l<D-HL} gamma[l] x_l)
    float hq[K];              // train_inverse_cov: the part of r^T A' r made of hoisted coordinates only

    __device__ __forceinline__ float pi(int k) const { return P[k * Lt::PK + Lt::O_PI]; }
    __device__ __forceinline__ float mu(int k, int l) const { return P[k * Lt::PK + Lt::O_MU + l]; }
    __device__ __forceinline__ float A(int k, int l, int m) const { return P[k * Lt::PK + Lt::O_A + tri_index(l, m)]; }
    __device__ __forceinline__ float nu(int k, int c) const { return P[k * Lt::PK + Lt::O_NU + c]; }
    __device__ __forceinline__ float ga(int k, int l, int c) const { return P[k * Lt::PK + Lt::O_GA + l * C + c]; }
    __device__ __forceinline__ bool flag(int k) const { return P[Lt::LP_ACT + k] != 0.0f; }
    __device__ __forceinline__ bool frozen() const { return P[Lt::LP_FROZEN] != 0.0f; }
    __device__ __forceinline__ bool act(int k) const { return flag(k) && (pi(k) > 0.0f); }

    __device__ __forceinline__ void load(const float* __restrict__ lds_block) {
        const float4* src = reinterpret_cast<const float4*>(lds_block);
#pragma unroll
        for (int i = 0; i < Lt::LP_STRIDE / 4; ++i) {
            const float4 q = src[i];
            P[4 * i + 0] = q.x; P[4 * i + 1] = q.y; P[4 * i + 2] = q.z; P[4 * i + 3] = q.w;
        }
    }

    // smoe.py:480,738 (bool_mask = kernel_list & pis>0), 809-819 (determinant factor, * pis)
    // IC (train_inverse_cov, smoe.py:734-735,791-793): A is symmetric and maha = r^T A r, so
    // exp(-maha/2) = exp2(-r^T A' r) with A' = SQ^2 A; As holds the coefficients c_lm of the quadratic
    // form over l >= m (c_ll = A'_ll, c_lm = 2 A'_lm).
    template <bool IC = false>
    __device__ __forceinline__ void derive(const KernelConsts& kc) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float det = 1.0f;
#pragma unroll
            for (int l = 0; l < D; ++l) det *= A(k, l, l);
            const float nq = kc.use_det ? det * kc.inv_n_dis : 1.0f;   // n_quo = n_div / n_dis (reciprocal from the host)
            coef[k] = act(k) ? nq * pi(k) : 0.0f;
#pragma unroll
            for (int l = 0; l < D; ++l)
#pragma unroll
                for (int m = 0; m <= l; ++m)
                    As[k][tri_index(l, m)] = (IC ? ((l == m) ? SMOE_SQ * SMOE_SQ : 2.0f * SMOE_SQ * SMOE_SQ) : SMOE_SQ) * A(k, l, m);
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float c = 0.0f;
                if (!IC) {
#pragma unroll
                    for (int l = m; l < D; ++l) c = fmaf(mu(k, l), As[k][tri_index(l, m)], c);
                }
                cz[k][m] = c;
            }
            if (!kc.train_gammas) {          // smoe.py:841-848: the slopes are not part of the graph
#pragma unroll
                for (int i = 0; i < D * C; ++i) P[k * Lt::PK + Lt::O_GA + i] = 0.0f;
            }
            if (kc.only_y_gamma) {           // smoe.py:725-729: qgamma_e * gamma_mask (channel 0 only)
#pragma unroll
                for (int l = 0; l < D; ++l)
#pragma unroll
                    for (int c = 1; c < C; ++c) P[k * Lt::PK + Lt::O_GA + l * C + c] = 0.0f;
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Fake-quantised parameters (quantize_pis, quantization_mode 2 / 3; smoe.py:474-538): TF
// fake_quant_with_min_max_{args,vars}.  Every lane holds all K kernels of its block, so the ranges of
// mode 3 (min / max over the kernels with qpis > 0) need no cross-lane step.
// ---------------------------------------------------------------------------
// Ranges of one block in mode 3 from its RAW packed parameters P (and the already quantised pis in qpi[]):
// A_diagonal: offset form over the diagonals; A_corr: over the whole d x d matrices (the structural zeros
// of the variable keep 0 inside the range); musX, gamma_e: plain; nu_e: offset form (smoe.py:497-530).
template <int D, int C, int K>
struct BlockRanges {
    FqRange ad, ac, mu, nu, ga;
    float lo[5], hi[5];     // 0 A_diag, 1 A_corr, 2 musX, 3 nu_e, 4 gamma_e (raw extremes, for the tie tests)
    __device__ __forceinline__ void compute(const float* P, const bool (&keep)[K], const KernelConsts& kc) {
        using Lt = Layout<D, C, K>;
        constexpr float INF = __builtin_huge_valf();
#pragma unroll
        for (int t = 0; t < 5; ++t) { lo[t] = INF; hi[t] = -INF; }
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (!keep[k]) continue;
            any = true;
            const float* p = P + k * Lt::PK;
#pragma unroll
            for (int l = 0; l < D; ++l) {
                lo[2] = fminf(lo[2], p[Lt::O_MU + l]); hi[2] = fmaxf(hi[2], p[Lt::O_MU + l]);
#pragma unroll
                for (int m = 0; m <= l; ++m) {
                    const float a = p[Lt::O_A + tri_index(l, m)];
                    const int t = (l == m) ? 0 : 1;
                    lo[t] = fminf(lo[t], a); hi[t] = fmaxf(hi[t], a);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) { lo[3] = fminf(lo[3], p[Lt::O_NU + c]); hi[3] = fmaxf(hi[3], p[Lt::O_NU + c]); }
#pragma unroll
            for (int i = 0; i < D * C; ++i) { lo[4] = fminf(lo[4], p[Lt::O_GA + i]); hi[4] = fmaxf(hi[4], p[Lt::O_GA + i]); }
        }
        lo[1] = fminf(lo[1], 0.0f); hi[1] = fmaxf(hi[1], 0.0f);          // diagonal / upper entries of the A_corr variable
        if (!any) {
#pragma unroll
            for (int t = 0; t < 5; ++t) { lo[t] = 0.0f; hi[t] = 0.0f; }
        }
        ad = fq_vars(lo[0], hi[0], kc.q_levels[0], true, kc.radial != 0);
        ac = fq_vars(lo[1], hi[1], kc.q_levels[0], false);
        mu = fq_vars(lo[2], hi[2], kc.q_levels[1], false);
        nu = fq_vars(lo[3], hi[3], kc.q_levels[2], true);
        ga = fq_vars(lo[4], hi[4], kc.q_levels[4], false);
    }
};

// kernel_count_as_norm_l1 (smoe.py:1012,1022-1023): number of kernels with (q)pis > 0, at least 1; P already quantised
template <int D, int C, int K>
__device__ __forceinline__ float count_pis(const float* P) {
    using Lt = Layout<D, C, K>;
    float cnt = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) cnt += (P[k * Lt::PK + Lt::O_PI] > 0.0f) ? 1.0f : 0.0f;
    return fmaxf(cnt, 1.0f);
}

// Replace the packed parameters P by the fake-quantised values the graph is built on (forward and readmission
// kernels: single pass; the fit kernel keeps a quantised LDS image instead, see refresh_quantised_image).
// FULL = false: only the pis (quantize_pis, the reference CLI default) -- a few instructions, kept as a
// run-time branch in the default kernels; modes 2 / 3 live in their own instantiations (QUANT) so that their
// register footprint does not reach the hot kernels.
// G (use_diff_center, smoe.py:390-394,746-747): the block's kernel-grid centres [K][D] or null -- the quantised variable is
// the OFFSET musX - grid, the graph reads fake_quant(offset) + grid.
template <int D, int C, int K, bool FULL>
__device__ __forceinline__ void quantize_packed(float* P, const KernelConsts& kc, const float* __restrict__ G = nullptr) {
    using Lt = Layout<D, C, K>;
    if (kc.qpis) {
        const FqRange r = fq_fixed(kc, 3);
#pragma unroll
        for (int k = 0; k < K; ++k) P[k * Lt::PK + Lt::O_PI] = fq_val(P[k * Lt::PK + Lt::O_PI], r);
    }
    if constexpr (!FULL) return;
    float gr[K * D];
    const bool centred = (G != nullptr) && (kc.qmode >= 2);
    if (centred) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int l = 0; l < D; ++l) { gr[k * D + l] = G[k * D + l]; P[k * Lt::PK + Lt::O_MU + l] -= gr[k * D + l]; }
    }
    if (kc.qmode == 2) {
        const FqRange ra = fq_fixed(kc, 0), rm = fq_fixed(kc, 1), rn = fq_fixed(kc, 2), rg = fq_fixed(kc, 4);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float* p = P + k * Lt::PK;
#pragma unroll
            for (int l = 0; l < D; ++l) p[Lt::O_MU + l] = fq_val(p[Lt::O_MU + l], rm);
#pragma unroll
            for (int i = 0; i < Lt::TRI; ++i) p[Lt::O_A + i] = fq_val(p[Lt::O_A + i], ra);
#pragma unroll
            for (int c = 0; c < C; ++c) p[Lt::O_NU + c] = fq_val(p[Lt::O_NU + c], rn);
#pragma unroll
            for (int i = 0; i < D * C; ++i) p[Lt::O_GA + i] = fq_val(p[Lt::O_GA + i], rg);
        }
    } else if (kc.qmode == 3) {
        bool keep[K];
#pragma unroll
        for (int k = 0; k < K; ++k) keep[k] = P[k * Lt::PK + Lt::O_PI] > 0.0f;     // pis_mask = qpis > 0
        BlockRanges<D, C, K> br;
        br.compute(P, keep, kc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float* p = P + k * Lt::PK;
#pragma unroll
            for (int l = 0; l < D; ++l) {
                if (kc.q_musx) p[Lt::O_MU + l] = fq_val(p[Lt::O_MU + l], br.mu);
#pragma unroll
                for (int m = 0; m <= l; ++m)
                    p[Lt::O_A + tri_index(l, m)] = fq_val(p[Lt::O_A + tri_index(l, m)], (l == m) ? br.ad : br.ac);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) p[Lt::O_NU + c] = fq_val(p[Lt::O_NU + c], br.nu);
#pragma unroll
            for (int i = 0; i < D * C; ++i) p[Lt::O_GA + i] = fq_val(p[Lt::O_GA + i], br.ga);
        }
    }
    if (centred) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int l = 0; l < D; ++l) P[k * Lt::PK + Lt::O_MU + l] += gr[k * D + l];
    }
}

// (Measured dead end, kept as a note: with G == 64 the block's derived constants are wave-uniform and could live in
// SGPRs, but VALU forms with an SGPR source issue at ~4.8 vs ~3.2 cycles and the readfirstlanes add ~100 instructions
// per iteration: 32x32 / K=8 / C=3 went 63 -> 49 Gpx-it/s.)
template <int D, int C, int K, int HL, bool IC = false>
__device__ __forceinline__ void hoist_const(BlockRegs<D, C, K>& R, const float (&xc)[D]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (IC) {
            // r_l = x_l - mu_l is a lane constant for l >= D-HL: hz[m] (m < D-HL) = sum_l c_lm r_l, hq = the terms
            // with both indices hoisted;  maha' = hq + sum_{m < D-HL} (c_mm r_m + sum_{m' < m} c_mm' r_m' + hz[m]) r_m
            float rh[D];
#pragma unroll
            for (int l = 0; l < D; ++l) rh[l] = (l >= D - HL) ? xc[l] - R.mu(k, l) : 0.0f;
            float q = 0.0f;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float h = 0.0f;
#pragma unroll
                for (int l = D - HL; l < D; ++l) {
                    if (l > m && m < D - HL) h = fmaf(R.As[k][tri_index(l, m)], rh[l], h);
                    if (l >= m && m >= D - HL) q = fmaf(R.As[k][tri_index(l, m)] * rh[l], rh[m], q);
                }
                R.hz[k][m] = h;
            }
            R.hq[k] = q;
        } else {
        float q = 0.0f;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float h = -R.cz[k][m];
#pragma unroll
            for (int l = D - 1; l >= D - HL; --l)
                if (l >= m) h = fmaf(xc[l], R.As[k][tri_index(l, m)], h);
            R.hz[k][m] = h;
            if (m >= D - HL) q = fmaf(h, h, q);            // z'_m is a lane constant there: its square is pre-summed
        }
        R.hq[k] = q;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float h = R.nu(k, c);
#pragma unroll
            for (int l = D - HL; l < D; ++l) h = fmaf(R.ga(k, l, c), xc[l], h);
            R.he[k][c] = h;
        }
    }
}

template <int D, int C, int K>
struct PixelOut {
    float wt[K];    // masked gate
    float q[C];     // quantised reconstruction
};

// One pixel.  TRAIN: accumulate raw gradient sums into acc[] (layout = packed params,
// "raw" meaning before the per-lane linear post-transform, see finish_partials):
//   acc[pi_k]      += u_k                      u_k = dL/dlog g_k
//   acc[mu_k,m]    += u_k z'_m                 z' = A'^T (x - mu)
//   acc[A_k,l,m]   += u_k x_l z'_m             (x, not x - mu: corrected in finish_partials)
//   acc[nu_k,c]    += wt_k G_c     acc[ga_k,l,c] += wt_k G_c x_l
// The influence slot accumulates sum_n wt_k (> 0 iff some pixel passes the mask, smoe.py:829).
// EXTG (ssim_opt): dL/dq of the pixel comes from the caller (gext[c], the SSIM adjoint) instead of the
// margin loss; the clip / fake-quant straight-through mask is still applied here and the loss slot is
// left to the caller.
// FLAGS: the influence test (smoe.py:829: any w~ > 0 over the block's pixels) is kept as one wavefront-wide lane mask per
// kernel, OR-ed on the scalar unit from the compare that feeds the mask anyway, instead of a VALU add per kernel-pixel;
// the caller turns the masks into the S_CNT partials after its loop.
// RAWLOSS (the training loop without per-pixel loss weights): the margin loss is accumulated per channel as sum (|diff| - eps)^2
// in lraw[c] -- one fused multiply-add per channel and pixel -- and weighted by the channel weight once after the loop.
template <int D, int C, int K, bool TRAIN, int HL = 0, bool EXTG = false, bool IC = false, bool FLAGS = false, bool RAWLOSS = false>
__device__ __forceinline__ void pixel(const BlockRegs<D, C, K>& R, const KernelConsts& kc,
                                      const float (&x)[D], const float (&t)[C], float lw,
                                      float* __restrict__ acc, PixelOut<D, C, K>& o,
                                      const float* __restrict__ gext = nullptr,
                                      unsigned long long* __restrict__ flags = nullptr, bool fed = true,
                                      float* __restrict__ lraw = nullptr, unsigned long long fedm = ~0ull) {
    // fedm (FLAGS): lane mask of `fed`, voted once per pixel by the caller -- `__ballot(infl && fed)` per kernel made the
    // compiler materialise the conjunction on the VALU (a v_cndmask + v_cmp per kernel and pixel in the loss-weight loop)
    // fed = false: the pixel was not drawn by the sub-sampled pass (smoe.py:1664-1667: the reference feeds the drawn pixels
    // only), so it takes no part in the influence test that prunes the kernel list (smoe.py:829,1763-1766); its loss weight 0
    // keeps it out of the loss and of every gradient
    using Lt = Layout<D, C, K>;
    float z[K][D], g[K];
    float S = 0.0f;
    // smoe.py:777-782,796,807: z = A^T (x - mu) ; n = exp(-|z|^2 / 2)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float maha = 0.0f;
        if (IC) {
            // smoe.py:791-793: maha = r^T A r, evaluated on the non-hoisted coordinates (see hoist_const);
            // z[k][l] keeps r_l for the reverse pass
            maha = (HL > 0) ? R.hq[k] : 0.0f;
#pragma unroll
            for (int l = 0; l < D - HL; ++l) {
                z[k][l] = x[l] - R.mu(k, l);
                float tq = (HL > 0) ? R.hz[k][l] : 0.0f;
#pragma unroll
                for (int m = 0; m <= l; ++m) tq = fmaf(R.As[k][tri_index(l, m)], z[k][m], tq);
                maha = fmaf(tq, z[k][l], maha);
            }
#pragma unroll
            for (int l = D - HL; l < D; ++l) z[k][l] = 0.0f;
        } else {
        maha = (HL > 0) ? R.hq[k] : 0.0f;                 // sum of the squared lane-constant components z'_m, m >= D-HL
#pragma unroll
        for (int m = 0; m < D - HL; ++m) {
            float zz = (HL > 0) ? R.hz[k][m] : -R.cz[k][m];
#pragma unroll
            for (int l = D - 1 - HL; l >= m; --l) zz = fmaf(x[l], R.As[k][tri_index(l, m)], zz);
            z[k][m] = zz;
            maha = (HL == 0 && m == 0) ? zz * zz : fmaf(zz, zz, maha);
        }
#pragma unroll
        for (int m = D - HL; m < D; ++m) z[k][m] = 0.0f;   // not read: the reverse pass completes these after the loop
        }
        g[k] = R.coef[k] * fast_exp2(-maha);
        S = (k == 0) ? g[k] : S + g[k];
    }
    // smoe.py:820-827: normalise, floor 1e-11, min-influence mask
    const float inv = fast_rcp(fmaxf(S, 10e-12f));
    float w[K], e[K][C], y[C];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        w[k] = g[k] * inv;
        const bool infl = w[k] > kc.tau;
        o.wt[k] = infl ? w[k] : 0.0f;
        if constexpr (FLAGS) flags[k] |= __ballot(infl) & fedm;
        else acc[Lt::S_CNT + k] += fed ? o.wt[k] : 0.0f;
        // smoe.py:840-848: e = nu + gamma^T x ; y = sum_k wt e
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float ee = (HL > 0) ? R.he[k][c] : R.nu(k, c);
#pragma unroll
            for (int l = 0; l < D - HL; ++l) ee = fmaf(R.ga(k, l, c), x[l], ee);
            e[k][c] = ee;
            y[c] = (k == 0) ? o.wt[k] * ee : fmaf(o.wt[k], ee, y[c]);
        }
    }
    // smoe.py:857,899 (clip + fake quant), 905-937 (mse / margin loss)
    float Gc[C];
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        // clip_by_value(0,1) then the fake-quant clamp to its nudged range: kc.nudged_max = min(1, nudged max)
        const float yc = __builtin_amdgcn_fmed3f(y[c], 0.0f, kc.nudged_max);
        o.q[c] = floorf(fmaf(yc, kc.inv_scale, 0.5f)) * kc.scale;
        const float diff = o.q[c] - t[c];
        const float ad = fabsf(diff) - kc.epsm;
        acc[Lt::S_SSE] = fmaf(diff, diff, acc[Lt::S_SSE]);
        const float cwl = kc.cw[c] * lw;
        if constexpr (RAWLOSS) lraw[c] = fmaf(ad, ad, lraw[c]);
        else if (!EXTG) acc[Lt::S_LOSS] = fmaf(cwl, ad * ad, acc[Lt::S_LOSS]);
        if (TRAIN) {
            // 2 cwl sign(diff): |diff| is either 0 or >= 2^-30, so diff * 2^100 saturates the clamp to +-2 cwl (loss weights are
            // not negative); ad * (+-2 cwl) rounds like 2 cwl * (+-ad): one multiply less than sign first, weight second
            const float c2 = cwl + cwl;
            const float gm = EXTG ? gext[c] : ad * __builtin_amdgcn_fmed3f(diff * 1.2676506e30f, -c2, c2);
            // clip_by_value / fake-quant straight-through: gradient only where neither clamp acted
            Gc[c] = (yc == y[c]) ? gm : 0.0f;
            dot = (c == 0) ? Gc[c] * y[c] : fmaf(Gc[c], y[c], dot);   // sum_k h_k w_k == sum_c G_c y_c
        }
    }
    if (!TRAIN) return;
    // ---- reverse pass, SURVEY Appendix A.4 (tf.gradients, smoe.py:1148) -------
    // u_k = w_k (h_k - dot) with h_k = M_k (e_k.G)  ==  wt_k (e_k.G) - w_k dot ;
    // when the normaliser sits on its 1e-11 floor it is a constant and the dot term drops.
    dot = (S > 10e-12f) ? dot : 0.0f;
    float Gx[(D - HL > 0) ? D - HL : 1][C];
#pragma unroll
    for (int l = 0; l < D - HL; ++l)
#pragma unroll
        for (int c = 0; c < C; ++c) Gx[l][c] = Gc[c] * x[l];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float eg = e[k][0] * Gc[0];
#pragma unroll
        for (int c = 1; c < C; ++c) eg = fmaf(e[k][c], Gc[c], eg);
        float u = fmaf(o.wt[k], eg, -(w[k] * dot));
        // One kernel per block: w = g / g = 1 and u = e.G - G.y vanishes identically (the restatement's IEEE division gives
        // exactly 0); with the hardware reciprocal a rounding residue would be left, which Adam normalises into a step of
        // the size of the learning rate on pis / musX / A.  Only on the 1e-11 floor of the normaliser the gate still moves.
        if constexpr (K == 1) u = (S > 10e-12f) ? 0.0f : u;
        float* a = acc + k * Lt::PK;
        a[Lt::O_PI] += u;
        if (IC) {
            // raw sums of u r_l and u r_l r_m over the non-hoisted coordinates (the rest follows from them, complete_const)
#pragma unroll
            for (int l = 0; l < D - HL; ++l) {
                const float ur = u * z[k][l];
                a[Lt::O_MU + l] += ur;
#pragma unroll
                for (int m = 0; m <= l; ++m) a[Lt::O_A + tri_index(l, m)] = fmaf(ur, z[k][m], a[Lt::O_A + tri_index(l, m)]);
            }
        } else {
#pragma unroll
        for (int m = 0; m < D - HL; ++m) {                 // z'_m is a lane constant for m >= D-HL: sum u z'_m = z'_m sum u,
            const float uz = u * z[k][m];                  // completed after the loop (complete_const)
            a[Lt::O_MU + m] += uz;
#pragma unroll
            for (int l = m; l < D - HL; ++l)               // hoisted rows l >= D-HL are x_l * sum(uz), done after the loop
                a[Lt::O_A + tri_index(l, m)] = fmaf(x[l], uz, a[Lt::O_A + tri_index(l, m)]);
        }
        }
        // sum wt_k G_c and sum wt_k (G_c x_l) as fused multiply-adds on the per-pixel products G_c x_l (one instruction per
        // accumulator; the product wt_k G_c of every kernel and channel is never formed)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a[Lt::O_NU + c] = fmaf(o.wt[k], Gc[c], a[Lt::O_NU + c]);
#pragma unroll
            for (int l = 0; l < D - HL; ++l)
                a[Lt::O_GA + l * C + c] = fmaf(o.wt[k], Gx[l][c], a[Lt::O_GA + l * C + c]);
        }
    }
}

// Hoisting: complete the accumulators whose x factor is one of the lane-constant coordinates.
template <int D, int C, int K, int HL, bool IC = false>
__device__ __forceinline__ void complete_const(const BlockRegs<D, C, K>& R, const float (&xc)[D], float* __restrict__ acc) {
    using Lt = Layout<D, C, K>;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float* a = acc + k * Lt::PK;
        if (!IC) {          // z'_m (m >= D-HL) depends on hoisted coordinates only: it is R.hz[k][m] for every pixel of the lane
#pragma unroll
            for (int m = D - HL; m < D; ++m) a[Lt::O_MU + m] = R.hz[k][m] * a[Lt::O_PI];
        }
#pragma unroll
        for (int l = D - HL; l < D; ++l) {
            if (IC) {       // sum u r_l = r_l sum u ; sum u r_l r_m = r_l sum u r_m (m not hoisted) = r_l r_m sum u (m hoisted)
                const float rl = xc[l] - R.mu(k, l);
#pragma unroll
                for (int m = 0; m <= l; ++m)
                    a[Lt::O_A + tri_index(l, m)] = (m < D - HL) ? rl * a[Lt::O_MU + m] : rl * ((xc[m] - R.mu(k, m)) * a[Lt::O_PI]);
                a[Lt::O_MU + l] = rl * a[Lt::O_PI];
            } else {
#pragma unroll
            for (int m = 0; m <= l; ++m) a[Lt::O_A + tri_index(l, m)] = xc[l] * a[Lt::O_MU + m];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) a[Lt::O_GA + l * C + c] = xc[l] * a[Lt::O_NU + c];
        }
    }
}

// (Measured dead end: the post-transform applied by the slot OWNERS to the block totals after the reduction -- each owner
// reading the few totals / parameters its slot involves from LDS -- is 145 instructions and ~15 LDS waits per owned slot
// with its divergent per-slot-type paths, against 124 instructions for all slots here: headline 386 -> 362 Gpx-it/s.)
// Per-lane linear post-transform of the raw partial sums into partial gradients (all maps
// are linear in the sums and use block-uniform parameters, so they commute with the
// cross-lane reduction).  With suz'_m = sum u z'_m (z' = SQ z) and sxz'_lm = sum u x_l z'_m:
//   d/dpi   = (sum u) / pi
//   d/dmu_l = sum_m A[l][m] suz_m                        (dm/dmu = -2 A z, dL/dm = -u/2)
//   d/dA_lm = -(sxz_lm - mu_l suz_m) + [l==m, use_det] (sum u)/A_ll
// radial_as (smoe.py:714-719): one steering value per kernel tiled over the diagonal -> its gradient is the trace of
// dL/dA; every diagonal slot receives it so that the d copies of the variable (and their Adam slots) stay equal.
template <int D, int C, int K>
__device__ __forceinline__ void tie_diagonal(float* __restrict__ a) {
    using Lt = Layout<D, C, K>;
    float tr = 0.0f;
#pragma unroll
    for (int l = 0; l < D; ++l) tr += a[Lt::O_A + tri_index(l, l)];
#pragma unroll
    for (int l = 0; l < D; ++l) a[Lt::O_A + tri_index(l, l)] = tr;
}

//   train_inverse_cov (IC), raw sums sur_l = sum u r_l, surr_lm = sum u r_l r_m:
//   d/dmu_l = sum_m A_lm sur_m (A symmetric) ;  d/dA_ll = -surr_ll / 2 + [use_det](sum u)/A_ll ;  d/dA_corr[l,m] = -surr_lm
template <int D, int C, int K, bool IC = false>
__device__ __forceinline__ void finish_partials(const BlockRegs<D, C, K>& R, const KernelConsts& kc,
                                                float* __restrict__ acc) {
    using Lt = Layout<D, C, K>;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float* a = acc + k * Lt::PK;
        const bool act = R.act(k);
        const float su = a[Lt::O_PI];
        if (IC) {
            float sur[D];
#pragma unroll
            for (int l = 0; l < D; ++l) sur[l] = a[Lt::O_MU + l];
#pragma unroll
            for (int l = 0; l < D; ++l) {
                float gm = 0.0f;
#pragma unroll
                for (int m = 0; m < D; ++m) gm = fmaf((l >= m) ? R.A(k, l, m) : R.A(k, m, l), sur[m], gm);
                a[Lt::O_MU + l] = gm;
#pragma unroll
                for (int m = 0; m <= l; ++m) {
                    float v = (l == m) ? -0.5f * a[Lt::O_A + tri_index(l, m)] : -a[Lt::O_A + tri_index(l, m)];
                    if (l == m && kc.use_det) v = fmaf(su, act ? fast_rcp(R.A(k, l, l)) : 0.0f, v);
                    a[Lt::O_A + tri_index(l, m)] = v;
                }
            }
            a[Lt::O_PI] = su * (act ? fast_rcp(R.pi(k)) : 0.0f);
            if (kc.radial) tie_diagonal<D, C, K>(a);
            continue;
        }
        float suz[D];
#pragma unroll
        for (int m = 0; m < D; ++m) suz[m] = a[Lt::O_MU + m] * SMOE_INV_SQ;
#pragma unroll
        for (int l = 0; l < D; ++l) {
            float gm = 0.0f;
#pragma unroll
            for (int m = 0; m <= l; ++m) gm = fmaf(R.A(k, l, m), suz[m], gm);
            a[Lt::O_MU + l] = gm;
        }
#pragma unroll
        for (int l = 0; l < D; ++l) {
#pragma unroll
            for (int m = 0; m <= l; ++m) {
                float v = fmaf(R.mu(k, l), suz[m], -(a[Lt::O_A + tri_index(l, m)] * SMOE_INV_SQ));
                if (l == m && kc.use_det) v = fmaf(su, act ? fast_rcp(R.A(k, l, l)) : 0.0f, v);
                a[Lt::O_A + tri_index(l, m)] = v;
            }
        }
        a[Lt::O_PI] = su * (act ? fast_rcp(R.pi(k)) : 0.0f);
        if (kc.radial) tie_diagonal<D, C, K>(a);
    }
}

// Owner-side post-transform (tilings with G >= 32).  After the reduction the owner of slot j holds the raw total; its
// gradient is linear in a few raw totals of the same kernel with coefficients that are parameters of that kernel:
//   g = c_self * tot + P1*T1 + P2*T2 + P3*T3 + (act ? 1/PR : 0) * TR
// (the same maps as finish_partials: d/dpi = su/pi; d/dmu_l = sum_m A[l][m] suz_m; d/dA_lm = mu_l suz_m - sxz_lm
// [+ su/A_ll on the diagonal with use_determinant]; IC: d/dmu_l = sum_m A_lm sur_m, d/dA_ll = -surr_ll/2 [+ su/A_ll],
// d/dA_corr = -surr_lm; nu_e / gamma_e: the raw sums).  The MU slots are published scaled by 1/SQ (suz = suz'/SQ).
// A descriptor holds the LDS byte addresses of the operands; unused terms point at the constant cells LP_ZERO / LP_ONE.
struct SlotDesc {
    float c_self, pub_scale;
    uint32_t T1, T2, T3, TR, P1, P2, P3, PR, FLAG, PIV;
};

template <int D, int C, int K, bool IC>
__device__ __forceinline__ SlotDesc build_slot_desc(int j, uint32_t tot_base, uint32_t img_base, uint32_t par_base,
                                                    bool patch_pis, bool use_det) {
    using Lt = Layout<D, C, K>;
    const uint32_t ZERO = par_base + 4u * Lt::LP_ZERO, ONE = par_base + 4u * Lt::LP_ONE;
    const int k = j / Lt::PK, o = j - k * Lt::PK;
    const uint32_t Tk = tot_base + 4u * (uint32_t)(k * Lt::PK), Pk = img_base + 4u * (uint32_t)(k * Lt::PK);
    SlotDesc d;
    d.c_self = 1.0f; d.pub_scale = 1.0f;
    d.T1 = d.T2 = d.T3 = d.TR = ZERO;
    d.P1 = d.P2 = d.P3 = ZERO;
    d.PR = ONE;
    d.FLAG = img_base + 4u * (uint32_t)(Lt::LP_ACT + k);
    d.PIV = patch_pis ? par_base + 4u * (uint32_t)(Lt::LP_QPI + k) : Pk + 4u * Lt::O_PI;
    if (o == Lt::O_PI) {
        d.c_self = 0.0f; d.TR = Tk + 4u * Lt::O_PI; d.PR = d.PIV;
    } else if (o < Lt::O_A) {
        const int l = o - Lt::O_MU;
        d.c_self = 0.0f;
        d.pub_scale = IC ? 1.0f : SMOE_INV_SQ;
        int n = 0;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            if (IC || m <= l) {
                const int hi = (l > m) ? l : m, lo = (l > m) ? m : l;
                const uint32_t P = Pk + 4u * (uint32_t)(Lt::O_A + hi * (hi + 1) / 2 + lo), T = Tk + 4u * (uint32_t)(Lt::O_MU + m);
                if (n == 0) { d.P1 = P; d.T1 = T; } else if (n == 1) { d.P2 = P; d.T2 = T; } else { d.P3 = P; d.T3 = T; }
                ++n;
            }
        }
    } else if (o < Lt::O_NU) {
        const int t = o - Lt::O_A;
        const int l = (t >= 3) ? 2 : ((t >= 1) ? 1 : 0);
        const int m = t - l * (l + 1) / 2;
        if (IC) {
            d.c_self = (l == m) ? -0.5f : -1.0f;
        } else {
            d.c_self = -SMOE_INV_SQ;
            d.P1 = Pk + 4u * (uint32_t)(Lt::O_MU + l);
            d.T1 = Tk + 4u * (uint32_t)(Lt::O_MU + m);
        }
        if (l == m && use_det) { d.TR = Tk + 4u * Lt::O_PI; d.PR = Pk + 4u * (uint32_t)(Lt::O_A + l * (l + 1) / 2 + l); }
    }
    return d;
}

// the addresses are byte offsets from the start of the workgroup's dynamic LDS
__device__ __forceinline__ float lds_f32(const float* __restrict__ lds, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + byte_off);
}

__device__ __forceinline__ float eval_slot_desc(const float* __restrict__ lds, const SlotDesc& d, float tot) {
    const float t1 = lds_f32(lds, d.T1), t2 = lds_f32(lds, d.T2), t3 = lds_f32(lds, d.T3), tr = lds_f32(lds, d.TR);
    const float p1 = lds_f32(lds, d.P1), p2 = lds_f32(lds, d.P2), p3 = lds_f32(lds, d.P3), pr = lds_f32(lds, d.PR);
    // (both operands loaded before the test: a short-circuit `&&` made the prior a dependent LDS read behind a branch)
    const float flag = lds_f32(lds, d.FLAG), piv = lds_f32(lds, d.PIV);
    const bool act = (flag != 0.0f) & (piv > 0.0f);                                             // smoe.py:480,738
    float g = d.c_self * tot;
    g = fmaf(p1, t1, g);
    g = fmaf(p2, t2, g);
    g = fmaf(p3, t3, g);
    return fmaf(act ? fast_rcp(pr) : 0.0f, tr, g);
}

// ---------------------------------------------------------------------------
// LDS carve-up shared by the fit and forward kernels
// ---------------------------------------------------------------------------
// The parameter tensors (get_params layout, leading block axis) of the workgroup's NB consecutive blocks are six
// contiguous runs: they move between global memory and an LDS tile with coalesced accesses, and the slot owners pick /
// deposit their packed slots in the tile (decode_slot with the block index inside the workgroup).
template <int D, int C, int K>
struct ParamTile {
    __host__ __device__ static constexpr int elems(int t) {            // floats per kernel of tensor t
        return (t == 0) ? 1 : ((t == 1) ? D : ((t == 2 || t == 3) ? D * D : ((t == 4) ? D * C : C)));
    }
    __host__ __device__ static constexpr int before(int t) {           // floats per kernel in front of tensor t
        int n = 0;
        for (int u = 0; u < t; ++u) n += elems(u);
        return n;
    }
    static constexpr int PER_KERNEL = before(6);                       // 1 + D + 2 D^2 + D C + C
};

template <int D, int C, int K, int G, int WAVES>
struct Tile {
    using Lt = Layout<D, C, K>;
    static constexpr int BPW = 64 / G;                  // blocks per wavefront
    static constexpr int NB = WAVES * BPW;              // blocks per workgroup
    // Cross-lane reduction geometry (reduce_slots): every lane stores its partial of slot j in row j of the wavefront's
    // scratch (64 floats + pad); a row is read back in units of U floats, one unit per lane, RPR rows per round:
    //   G = 16, 32: unit = the G partials of one block, lane (grp, sub) sums row `sub` of block `grp`;
    //   G = 64    : unit = half a row, lanes 2r and 2r+1 sum the halves of row r and exchange them (DPP quad_perm).
    //               parameter-rich triples (>= 90 slots: large blocks, where LDS decides how many wavefronts a CU holds):
    //               quarter rows, four lanes per row and two exchanges -- half the scratch (16x16x4 / 1 024-pixel blocks then
    //               fit four workgroups per CU instead of three)
    static constexpr int U = (G == 64) ? ((Lt::NSLOT >= 90) ? 16 : 32) : G;
    static constexpr int UL = G / U;                    // lanes that share a row (1, 2 or 4)
    static constexpr int RPR = 64 / UL / (64 / G);      // rows per round = lanes of a block / lanes per row
    static constexpr int NROUND = (Lt::NSLOT + RPR - 1) / RPR;
    // rows of one pass = rows of the scratch: one round per pass.  (Two rounds per pass -- one LDS hand-off for 64
    // slots -- were measured: 1 024 blocks +2 %, but the 17 KB of scratch per wavefront cost the 1 024-pixel blocks a
    // workgroup per CU: 16x16x4 153 -> 106 Gpx-it/s.)
    static constexpr int CH = RPR;
    static constexpr int RPP = CH / RPR;                // rounds per pass
    static constexpr int ROW = 64 + 4;                  // padded row (bank-conflict-free b128 reads)
    static constexpr int NCHUNK = (Lt::NSLOT + CH - 1) / CH;
    // slots owned per lane: all UL lanes of a row hold its total after the exchange; lane u of them owns the slot of the
    // rounds q with q % UL == u
    // G = 64: the reduction runs in registers instead (reduce_slots_regs: v_permlane32_swap / v_permlane16_swap halve the
    // slot set twice, DPP rotations finish inside the rows of 16 lanes): row r = lane / 16 ends up with the totals of the
    // RM slots [r RM, (r + 1) RM), lane c of the row owns slots r RM + c, r RM + c + 16, ...
    static constexpr bool REGRED = (G == 64) && (SMOE_REGRED != 0);
    static constexpr int RM = (Lt::NSLOT + 3) / 4;
    static constexpr int SPL = REGRED ? (RM + 15) / 16 : (NROUND + UL - 1) / UL;
    __host__ __device__ static constexpr int slot_of(int sub, int s) {
        if (REGRED) {
            const int i = (sub % 16) + 16 * s;
            return (i < RM) ? (sub / 16) * RM + i : Lt::NSLOT;          // NSLOT = no slot
        }
        return (UL * s + (sub % UL)) * RPR + sub / UL;
    }
    static constexpr int THREADS = WAVES * 64;
    static constexpr int FWD_PXR = 16;                  // evaluation kernel: pixels per lane it keeps in registers (no staged planes)
    static_assert(NB * K * ParamTile<D, C, K>::PER_KERNEL <= WAVES * CH * ROW, "the parameter tile of a workgroup is staged in the reduction scratch");
    static constexpr int MV_STRIDE = round_up(2 * Lt::NPAR, 4);   // Adam m,v image of one block
    // Distance between the blocks of a wavefront in the target / loss-weight planes.  With 16 lanes per block the two
    // blocks of a 32-lane group read the same LDS banks (block size = multiple of 32 floats: SQ_LDS_BANK_CONFLICT = 18 % of
    // the LDS cycles on the headline kernel); padding the stride by 16 floats removes the conflict and was measured
    // SLOWER (A/B in one session, three alternating runs each: 384.2 vs 371.2 Gpx-it/s) -- LDS is not the binding pipe
    // and the unpadded planes keep the block base a multiple of 1 KB.  Kept at 0.
    static constexpr int TGT_PAD = 0;
    __host__ __device__ static constexpr int tgt_stride(int N) { return C * N + TGT_PAD; }
    __host__ __device__ static constexpr int lw_stride(int N) { return N + TGT_PAD; }
    
    // float offsets inside dynamic LDS
    // CR = coordinate rows staged in LDS (D, or D - HL when the trailing HL axes are hoisted)
    __host__ __device__ static int off_coords() { return 0; }
    __host__ __device__ static int off_par(int N, int CR) { return round_up(CR * N, 4); }
    __host__ __device__ static int off_mv(int N, int CR) { return off_par(N, CR) + NB * Lt::LP_STRIDE; }
    // The wavefront-per-block tiling turns the raw sums into gradients on the OWNER side when the blocks are small (eval_slot_desc):
    // per block a table of 12-dword descriptors, one per packed parameter, built once per launch.  (The 16-lane tiling
    // owns three slots per lane and keeps the per-lane transform before the reduction, finish_partials: measured there.)
    static constexpr bool OWNER_POST = (G >= 64);        // (32 lanes per block: measured, no gain -- 209.5 vs 209.9 Gpx-it/s at 4 096 blocks)
    static constexpr int DESC_DW = 12;
    static constexpr int DESC_STRIDE = OWNER_POST ? Lt::NPAR * DESC_DW : 0;
    static constexpr int TOT_STRIDE = round_up(Lt::NSLOT, 4);          // published raw totals of one block (in the scratch)
    static_assert(!OWNER_POST || BPW * TOT_STRIDE <= CH * ROW, "the published totals must fit into the reduction scratch");
    // the descriptor tables sit behind everything else (FitArgs::desc_off, set by the launcher when the blocks are small:
    // few pixels per lane, LDS to spare; 0 = keep the per-lane transform)
    __host__ __device__ static bool wants_owner_post(int N) { return OWNER_POST && N <= 8 * G; }
    __host__ __device__ static int off_scratch(int N, int CR) { return off_mv(N, CR) + NB * MV_STRIDE; }
    __host__ __device__ static int off_tgt(int N, int CR) { return off_scratch(N, CR) + WAVES * CH * ROW; }
    __host__ __device__ static int off_lw(int N, int CR) { return off_tgt(N, CR) + NB * tgt_stride(N); }
    // hq: the quantised image is only carved out when the graph is fake-quantised (fit kernels)
    __host__ __device__ static size_t bytes(int N, bool has_lw, int CR = D, bool hq = false) {
        return sizeof(float) * (size_t)off_ssim(N, has_lw, CR, hq);
    }
    // fake-quantised graph (quantize_pis / quantization_mode 2, 3): per block a second parameter image holding the
    // quantised values the forward / backward read (written by the slot owners after every Adam step), the five
    // mode-3 range records (8 floats each) and a 12-float hand-off area of the block-wide all-reduce
    static constexpr int QI_RNG = 40;
    static constexpr int QI_OUT = 12;
    static constexpr int QI_STRIDE = round_up(Lt::LP_STRIDE + QI_RNG + QI_OUT, 4);
    __host__ __device__ static int off_qimg(int N, bool has_lw, int CR) { return round_up(off_lw(N, CR) + (has_lw ? NB * lw_stride(N) : 0), 4); }
    // ssim_opt (G == 64, one block per wavefront): the two tap tables of the workgroup, then per wavefront
    // the planes X [C][N] (quantised reconstruction -> dL/dq), Wa [5][N] (column sums of x, x^2, xy, y, y^2;
    // later the row pass of the adjoint) and Wb [3][N] (coefficient maps)
    __host__ __device__ static int off_ssim(int N, bool has_lw, int CR, bool hq = false) { return off_qimg(N, has_lw, CR) + (hq ? NB * QI_STRIDE : 0); }
    // G == 16 (16x16 blocks only): the SSIM stage runs in registers (ssim_block16), LDS holds just X per block
    // 3-d blocks [bh][bw][bt]: a third tap table, and Wb holds five planes (three axis passes each way)
    __host__ __device__ static int ssim_tabs(int bh, int bw, int bt = 0) { return (G == 16) ? 0 : round_up(11 * (bh + bw + bt), 4); }
    __host__ __device__ static int ssim_wave(int N) { return (G == 16) ? BPW * C * N : round_up(C * N + ((D == 3) ? 10 : 8) * N, 4); }
    __host__ __device__ static size_t bytes_ssim(int N, bool has_lw, int CR, int bh, int bw, bool hq = false, int bt = 0) {
        return sizeof(float) * (size_t)(off_ssim(N, has_lw, CR, hq) + ssim_tabs(bh, bw, bt) + WAVES * ssim_wave(N));
    }
};

// Coordinates, targets and loss weights of the workgroup's blocks -> LDS.  The NB blocks are consecutive in the
// [B, C, N] / [B, N] arrays, so each plane set is ONE contiguous run of floats: 16-byte global loads and LDS stores
// when the block size is a multiple of four pixels (blocks past the end of the batch re-read the last block).
template <int D, int C, int K, int G, int WAVES, int CR = D>
__device__ __forceinline__ void stage_inputs(const float* __restrict__ coords, const float* __restrict__ target,
                                             const float* __restrict__ loss_w, int B, int N, int blk0,
                                             float* __restrict__ lds, bool planes = true) {
    using T = Tile<D, C, K, G, WAVES>;
    static_assert(T::TGT_PAD == 0, "the staged planes of a workgroup are one contiguous run");
    float* s_coords = lds + T::off_coords();
    float* s_tgt = lds + T::off_tgt(N, CR);
    float* s_lw = lds + T::off_lw(N, CR);
    // the NB blocks follow each other in global memory AND in LDS: a straight copy of NB * per floats (no per-element
    // block arithmetic -- a run-time integer division costs ~40 instructions).  Elements past the end of the batch
    // re-read the last 16 bytes of the array (valid memory; those blocks' results are never stored).
    auto copy_planes = [&](const float* __restrict__ src, float* __restrict__ dst, int per) {
        const size_t g0 = (size_t)blk0 * per;
        const size_t gend = (size_t)B * per;
        if ((per & 3) == 0) {
            const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
            const size_t last4 = (gend >> 2) - 1;
            for (int i = threadIdx.x; i < T::NB * (per >> 2); i += T::THREADS)
                reinterpret_cast<float4*>(dst)[i] = src4[min((g0 >> 2) + (size_t)i, last4)];
        } else {
            for (int i = threadIdx.x; i < T::NB * per; i += T::THREADS) dst[i] = src[min(g0 + (size_t)i, gend - 1)];
        }
    };
    for (int i = threadIdx.x; i < CR * N; i += T::THREADS) s_coords[i] = coords[i];
    if (!planes) return;
    copy_planes(target, s_tgt, C * N);
    if (loss_w != nullptr) copy_planes(loss_w, s_lw, N);
}

// The tile is the concatenation of the six runs; element i of it belongs to tensor t(i).  Fetch = global -> registers
// (all loads of a thread go out together: ONE memory latency for parameters AND Adam slots), put = registers -> LDS.
template <int D, int C, int K, int NB, int THREADS>
struct TileIO {
    using PT = ParamTile<D, C, K>;
    static constexpr int TOTAL = NB * K * PT::PER_KERNEL;
    static constexpr int MAXE = (TOTAL + THREADS - 1) / THREADS;
    // tensor of tile element i and its index inside that tensor's run
    __device__ static __forceinline__ void locate(int i, int& t, int& local) {
        t = 0; local = i;
#pragma unroll
        for (int u = 1; u < 6; ++u)
            if (i >= NB * K * PT::before(u)) { t = u; local = i - NB * K * PT::before(u); }
    }
    __device__ static __forceinline__ void fetch(const smoe_params& s, int blk0, int B, float (&v)[MAXE]) {
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int i = threadIdx.x + e * THREADS;
            v[e] = 0.0f;
            if (i < TOTAL) {
                int t, local;
                locate(i, t, local);
                const int el = PT::elems(t);
                const long g = min((long)blk0 * K * el + local, (long)B * K * el - 1);     // blocks past the batch: valid memory
                v[e] = pick(s, t)[g];
            }
        }
    }
    __device__ static __forceinline__ void put(const float (&v)[MAXE], float* __restrict__ tile) {
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int i = threadIdx.x + e * THREADS;
            if (i < TOTAL) tile[i] = v[e];
        }
    }
};

// index of packed slot j of workgroup-local block lb inside the tile
template <int D, int C, int K, int NB>
__device__ __forceinline__ int tile_index(int j, int lb) {
    int tensor, kern; long off;
    decode_slot<D, C, K>(j, lb, tensor, off, kern);
    using PT = ParamTile<D, C, K>;
    const int before = (tensor == 0) ? PT::before(0) : ((tensor == 1) ? PT::before(1) : ((tensor == 2) ? PT::before(2)
                     : ((tensor == 3) ? PT::before(3) : ((tensor == 4) ? PT::before(4) : PT::before(5)))));
    return NB * K * before + (int)off;
}

// Cross-lane reduction of acc[FIRST..NSLOT) over the G lanes of a block through an LDS transpose: the lane that owns
// slot j = Tile::slot_of(sub, s) ends up with its total in total[s].  A pass moves CH slots: every lane stores its
// partials of these slots as rows (conflict-free 4-byte stores); in each round every lane then sums one unit of one
// row with 16-byte reads (rows are padded by 4 floats so the b128 reads do not conflict).  All reads of a pass are
// issued from clamped addresses without branches, so they go out back to back behind ONE wait.
// G = 64, in registers.  A swap exchanges the upper half of its first operand with the lower half of its second
// (v_permlane32_swap: halves of the wavefront; v_permlane16_swap: odd / even rows of 16 lanes), so ONE swap + ONE add
// reduce TWO slots over the pair of lanes: the first slot's sum stays in the lower half, the second's in the upper.  Two
// such levels leave row r of the wavefront with the partial sums of slots [r RM, (r + 1) RM) over lanes {c, c+16, c+32,
// c+48}; four DPP row rotations (8, 4, 2, 1) finish them inside the row.  No LDS traffic and no waits (the transpose costs
// 223 instructions + 17 waits for 42 slots against ~130 here) -- and slower all the same: see SMOE_REGRED.
template <int D, int C, int K, int G, int WAVES, int FIRST>
__device__ __forceinline__ void reduce_slots_regs(const float* __restrict__ acc, int lane, float (&total)[Tile<D, C, K, G, WAVES>::SPL]) {
    using T = Tile<D, C, K, G, WAVES>;
    using Lt = Layout<D, C, K>;
    constexpr int NS = Lt::NSLOT, M = T::RM;
    auto wanted = [](int j) constexpr { return j >= FIRST && j < NS; };
    float v1[2 * M];
#pragma unroll
    for (int j = 0; j < 2 * M; ++j) {
        v1[j] = 0.0f;
        if (!wanted(j) && !wanted(j + 2 * M)) continue;                    // compile-time
        const float x = (j < NS) ? acc[j] : 0.0f;
        const float y = (j + 2 * M < NS) ? acc[j + 2 * M] : 0.0f;
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        v1[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    float v2[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        v2[i] = 0.0f;
        if (!wanted(i) && !wanted(i + M) && !wanted(i + 2 * M) && !wanted(i + 3 * M)) continue;
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v1[i]), __float_as_uint(v1[i + M]), false, false);
        float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, true));    // row_ror:8
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xf, 0xf, true));    // row_ror:4
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x122, 0xf, 0xf, true));    // row_ror:2
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x121, 0xf, 0xf, true));    // row_ror:1
        v2[i] = t;
    }
    const int c = lane & 15;
#pragma unroll
    for (int s = 0; s < T::SPL; ++s) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (16 * s + i < M) t = (c == i) ? v2[16 * s + i] : t;
        total[s] = t;
    }
}

// Partial sums into a transpose scratch (row = slot, column = lane) with ds_write_addtid_b32 (address = M0[15:0] + offset + 4 * lane: no address register, two
// LDS-path cycles per instruction instead of the four of ds_write_b32 -- MI355X_MICROARCH.md, LDS): row j of the scratch starts
// j * ROWB bytes after the wavefront's first row (M0 = the LDS address of the wavefront's column 0 of row 0).  The instruction reaches the first 64 KB of the LDS only (M0[15:0]) and is
// invisible to the compiler's wait counting: the last block waits for the stores itself.  Blocks of at most eight stores, M0
// re-based per block so that the 16-bit offset field never overflows.
template <int ROWB, int N, bool LAST>
__device__ __forceinline__ void addtid_store_block(const float* __restrict__ v, uint32_t m0) {
    static_assert(N >= 1 && N <= 8 && 7 * ROWB < 65536, "addtid_store_block");
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = v[(i < N) ? i : N - 1];
    if constexpr (N == 8) {
        asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %0 offset:%9\n\tds_write_addtid_b32 %1 offset:%10\n\t"
                     "ds_write_addtid_b32 %2 offset:%11\n\tds_write_addtid_b32 %3 offset:%12\n\t"
                     "ds_write_addtid_b32 %4 offset:%13\n\tds_write_addtid_b32 %5 offset:%14\n\t"
                     "ds_write_addtid_b32 %6 offset:%15\n\tds_write_addtid_b32 %7 offset:%16"
                     :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(m0),
                        "i"(0 * ROWB), "i"(1 * ROWB), "i"(2 * ROWB), "i"(3 * ROWB), "i"(4 * ROWB), "i"(5 * ROWB), "i"(6 * ROWB), "i"(7 * ROWB)
                     : "m0", "memory");
    } else {
        // a short last block: one store per statement (M0 stays put: every statement names it as clobbered and sets it again)
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" :: "v"(x[i]), "s"(m0 + (uint32_t)(i * ROWB)) : "m0", "memory");
    }
    if constexpr (LAST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int ROWB, int NSLOT, int J0 = 0>
__device__ __forceinline__ void addtid_store_rows(const float* __restrict__ acc, uint32_t m0) {
    constexpr int N = (NSLOT - J0 >= 8) ? 8 : NSLOT - J0;
    constexpr bool LAST = J0 + N >= NSLOT;
    addtid_store_block<ROWB, N, LAST>(acc + J0, m0 + (uint32_t)(J0 * ROWB));
    if constexpr (!LAST) addtid_store_rows<ROWB, NSLOT, J0 + N>(acc, m0);
}
__device__ __forceinline__ uint32_t lds_address(const float* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

template <int D, int C, int K, int G, int WAVES, int FIRST>
__device__ __forceinline__ void reduce_slots_lds(const float* __restrict__ acc, float* __restrict__ scratch_wave,
                                                 int lane, float (&total)[Tile<D, C, K, G, WAVES>::SPL]) {
    using T = Tile<D, C, K, G, WAVES>;
    using Lt = Layout<D, C, K>;
    // (opaque lane index: the row / unit / scratch addresses derived from it are loop invariant, and hoisted out of the
    // iteration loop they stay live across the pixel loop -- the 16x16x4 kernel sits at 253 of 256 VGPRs)
    int ln = lane;
    if constexpr (G == 64) asm volatile("" : "+v"(ln));       // (the 16-lane kernels have the registers and lose 1 % to the recomputation)
    const int rho = (G == 64) ? (ln / T::UL) : (ln % G);                      // the lane's row within a round
    const int base = (G == 64) ? (ln % T::UL) * T::U : (ln / G) * G;          // its unit within the row
#pragma unroll
    for (int s = 0; s < T::SPL; ++s) total[s] = 0.0f;
    // (ds_write_addtid_b32 for these stores as in the duo kernel: measured, no difference on any shape -- the stores of this
    // tiling overlap with the other wavefronts' pixel loops)
#pragma unroll
    for (int c = 0; c < T::NCHUNK; ++c) {
        if ((c + 1) * T::CH <= FIRST) continue;            // compile-time: nothing wanted in this pass
#pragma unroll
        for (int a = 0; a < T::CH; ++a) {
            const int j = c * T::CH + a;
            if (j >= FIRST && j < Lt::NSLOT) scratch_wave[a * T::ROW + ln] = acc[j];
        }
        wave_lds_sync();
        float tot[T::RPP];
#pragma unroll
        for (int r = 0; r < T::RPP; ++r) {
            const int q = c * T::RPP + r;                   // global round
            tot[r] = 0.0f;
            if (q * T::RPR >= Lt::NSLOT || (q + 1) * T::RPR <= FIRST) continue;       // compile-time
            // rows past the last slot are not written: clamp the row (those sums are never used)
            const int rows_here = (Lt::NSLOT - q * T::RPR < T::RPR) ? (Lt::NSLOT - q * T::RPR) : T::RPR;
            const int rr = (rho < rows_here) ? rho : (rows_here - 1);
            const float4* row = reinterpret_cast<const float4*>(scratch_wave + (r * T::RPR + rr) * T::ROW + base);
            float4 sum = row[0];
#pragma unroll
            for (int i = 1; i < T::U / 4; ++i) {
                const float4 v = row[i];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            tot[r] = (sum.x + sum.y) + (sum.z + sum.w);
        }
#pragma unroll
        for (int r = 0; r < T::RPP; ++r) {
            const int q = c * T::RPP + r;
            if (q * T::RPR >= Lt::NSLOT || (q + 1) * T::RPR <= FIRST) continue;
            float t = tot[r];
            if (G == 64) {
                // the other parts of the row sit in the neighbouring lanes: quad_perm [1,0,3,2] (then [2,3,0,1]); every
                // lane of the group ends up with the sum
                t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xf, 0xf, true));
                if (T::UL == 4) t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xf, 0xf, true));
                if ((q % T::UL) == (ln % T::UL)) total[q / T::UL] = t;
            } else {
                total[q] = t;
            }
        }
        wave_lds_sync();
    }
}

template <int D, int C, int K, int G, int WAVES, int FIRST>
__device__ __forceinline__ void reduce_slots(const float* __restrict__ acc, float* __restrict__ scratch_wave,
                                             int lane, float (&total)[Tile<D, C, K, G, WAVES>::SPL]) {
    if constexpr (Tile<D, C, K, G, WAVES>::REGRED) reduce_slots_regs<D, C, K, G, WAVES, FIRST>(acc, lane, total);
    else reduce_slots_lds<D, C, K, G, WAVES, FIRST>(acc, scratch_wave, lane, total);
}

// ---------------------------------------------------------------------------
// SSIM stage for 16x16 blocks on the 16-lanes-per-block tiling, entirely in registers: lane `sub` of a
// block owns image column `sub` (pixels n = i*16 + sub), so
//   * the column pass (axis 0) is in-lane: out[i] = sum_r T16[i][r] in[r] with COMPILE-TIME weights
//     (the 16x16 tap matrix is folded to literals, zero taps vanish);
//   * the row pass (axis 1) is across the 16 lanes of the block = one DPP row: eleven row_shl / row_shr
//     shifted operands (bound_ctrl: lanes outside the row read 0) times per-lane weights
//     wj[a] = T16[sub][sub + a - 5].
// No LDS traffic besides reading q / target and writing dL/dq, no barriers.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr float ssim_gauss(int a) {      // image_ops_impl.py:132-149, size 11, sigma 1.5
    constexpr float g[11] = {1.0283801239e-03f, 7.5987582095e-03f, 3.6000773311e-02f, 1.0936068743e-01f,
                             2.1300554276e-01f, 2.6601171494e-01f, 2.1300554276e-01f, 1.0936068743e-01f,
                             3.6000773311e-02f, 7.5987582095e-03f, 1.0283801239e-03f};
    return g[a];
}
__host__ __device__ constexpr float ssim_t16(int i, int r) {  // SYMMETRIC pad 5 + 11 taps on a 16-sample axis
    float s = 0.0f;
    for (int a = 0; a < 11; ++a) {
        int m = i + a - 5;
        if (m < 0) m = -1 - m;
        if (m >= 16) m = 31 - m;
        if (m == r) s += ssim_gauss(a);
    }
    return s;
}

template <int NQ>
__device__ __forceinline__ void ssim_colpass16(const float (&in)[NQ][16], float (&out)[NQ][16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[q][i] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float w = ssim_t16(i, r);
            if (w != 0.0f) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) out[q][i] = fmaf(w, in[q][r], out[q][i]);
            }
        }
    }
}

// value of lane (L + SH) of the same 16-lane row, 0 when that lane is outside the row
template <int SH>
__device__ __forceinline__ float ssim_row_neighbour(float v) {
    constexpr int ctrl = (SH > 0) ? (0x100 + SH) : (0x110 - SH);   // row_shl:SH / row_shr:-SH
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true));
}

__device__ __forceinline__ float ssim_rowpass16(float v, const float (&wj)[11]) {
    float s = wj[5] * v;
    s = fmaf(wj[6], ssim_row_neighbour<1>(v), s);
    s = fmaf(wj[4], ssim_row_neighbour<-1>(v), s);
    s = fmaf(wj[7], ssim_row_neighbour<2>(v), s);
    s = fmaf(wj[3], ssim_row_neighbour<-2>(v), s);
    s = fmaf(wj[8], ssim_row_neighbour<3>(v), s);
    s = fmaf(wj[2], ssim_row_neighbour<-3>(v), s);
    s = fmaf(wj[9], ssim_row_neighbour<4>(v), s);
    s = fmaf(wj[1], ssim_row_neighbour<-4>(v), s);
    s = fmaf(wj[10], ssim_row_neighbour<5>(v), s);
    s = fmaf(wj[0], ssim_row_neighbour<-5>(v), s);
    return s;
}

// Same contract as ssim_block for one 16x16 block handled by 16 lanes: X [C][256] holds q and receives
// dL/dq; returns the lane's share of -sum_c sw_c * sum(l * cs).
template <int C, bool GRAD>
__device__ __forceinline__ float ssim_block16(float* __restrict__ X, const float* __restrict__ tgt, int sub,
                                              const float (&wj)[11], const float* __restrict__ sw) {
    float part = 0.0f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const float swc = (c == 0) ? sw[0] : ((c == 1) ? sw[1] : sw[2]);
        float* xp = X + c * 256 + sub;
        const float* yp = tgt + c * 256 + sub;
        float in[5][16], v[5][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float xv = xp[i * 16], yv = yp[i * 16];
            in[0][i] = xv; in[1][i] = xv * xv; in[2][i] = xv * yv; in[3][i] = yv; in[4][i] = yv * yv;
        }
        ssim_colpass16<5>(in, v);
        float co[3][16];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float mx = ssim_rowpass16(v[0][i], wj), sx = ssim_rowpass16(v[1][i], wj);
            const float pxy = ssim_rowpass16(v[2][i], wj);
            const float my = ssim_rowpass16(v[3][i], wj), sy = ssim_rowpass16(v[4][i], wj);
            const float num0 = mx * my * 2.0f;
            const float den0 = mx * mx + my * my;
            const float N0 = num0 + SSIM_C1, D0 = den0 + SSIM_C1;
            const float N1 = (pxy * 2.0f - num0) + SSIM_C2;
            const float D1 = ((sx + sy) - den0) + SSIM_C2;
            const float r0 = fast_rcp(D0), r1 = fast_rcp(D1);      // 1 ulp: far inside the 2e-5 loss tolerance
            const float lum = N0 * r0, cs = N1 * r1;
            acc = fmaf(lum, cs, acc);
            if (GRAD) {
                const float dl = (2.0f * my - lum * (2.0f * mx)) * r0;
                const float dc = (cs * (2.0f * mx) - 2.0f * my) * r1;
                co[0][i] = -swc * fmaf(cs, dl, lum * dc);
                co[1][i] = swc * ((lum * cs) * r1);
                co[2][i] = -swc * ((lum + lum) * r1);
            }
        }
        part -= swc * acc;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int q = 0; q < 3; ++q) co[q][i] = ssim_rowpass16(co[q][i], wj);
            }
            float g3[3][16];
            ssim_colpass16<3>(co, g3);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                xp[i * 16] = fmaf(in[3][i], g3[2][i], fmaf(in[0][i] + in[0][i], g3[1][i], g3[0][i]));
        }
    }
    return part;
}

// All-reduce of ten per-lane values over the G lanes of a block (sum, or min with MIN) through the wavefront's
// reduction scratch: every lane stores its ten values as rows, lane `sub` < 10 folds row `sub` over the block's
// lanes in a fixed order and publishes the result in s_out, every lane reads the ten results back.
template <int G, int ROW, bool MIN>
__device__ __forceinline__ void block_allreduce10(float (&v)[10], float* __restrict__ scratch_wave, float* __restrict__ s_out,
                                                  int lane, int grp, int sub) {
#pragma unroll
    for (int i = 0; i < 10; ++i) scratch_wave[i * ROW + lane] = v[i];
    wave_lds_sync();
    if (sub < 10) {
        const float4* row = reinterpret_cast<const float4*>(scratch_wave + sub * ROW + grp * G);
        float4 r = row[0];
#pragma unroll
        for (int i = 1; i < G / 4; ++i) {
            const float4 q = row[i];
            if (MIN) { r.x = fminf(r.x, q.x); r.y = fminf(r.y, q.y); r.z = fminf(r.z, q.z); r.w = fminf(r.w, q.w); }
            else { r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
        }
        s_out[sub] = MIN ? fminf(fminf(r.x, r.y), fminf(r.z, r.w)) : (r.x + r.y) + (r.z + r.w);
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = s_out[i];
    wave_lds_sync();
}

// ---------------------------------------------------------------------------
// fit kernel: n_iters x (forward + backward + prune + TF1 Adam), parameters resident
// ---------------------------------------------------------------------------
template <int D, int C, int K, bool HAS_LW, int HL, bool IC = false>
__device__ __forceinline__ void pixel_loop_train(const BlockRegs<D, C, K>& R, const KernelConsts& kc,
                                                 const float* __restrict__ s_coords, const float* __restrict__ s_tgt,
                                                 const float* __restrict__ s_lw, int N, int G, int sub,
                                                 float* __restrict__ acc, bool sample = false) {
    // sample: the loss weights are a pixel sub-sample (FitArgs::lw_is_sample): weight 0 = pixel not fed
    const int pxl = (N + G - 1) / G;
    const int full = N / G;                        // steps in which every lane of the block has a pixel
    unsigned long long flags[K];
#pragma unroll
    for (int k = 0; k < K; ++k) flags[k] = 0ull;
    // full steps: uniform control flow, so the lane-mask votes stay on the scalar unit.  Unrolled by two by hand: the
    // votes are convergent operations, which keeps the compiler from unrolling a loop of run-time trip count itself.
    constexpr bool RAWL = !HAS_LW;
    const unsigned long long not_sampled = sample ? 0ull : ~0ull;           // wave-uniform: all pixels are fed unless the weights are a sample
    float lraw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) lraw[c] = 0.0f;
    auto step = [&](int i) {
        const int n = i * G + sub;
        float x[D], t[C];
#pragma unroll
        for (int l = 0; l < D; ++l) x[l] = (l < D - HL) ? s_coords[l * N + n] : 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) t[c] = s_tgt[c * N + n];
        const float lw = HAS_LW ? s_lw[n] : 1.0f;
        PixelOut<D, C, K> o;
        const bool fed = !(HAS_LW && sample) || lw != 0.0f;
        unsigned long long fedm = ~0ull;
        if constexpr (HAS_LW) fedm = __ballot(lw != 0.0f) | not_sampled;        // (the vote of the compare itself, no branch in the step)
        pixel<D, C, K, true, HL, false, IC, true, RAWL>(R, kc, x, t, lw, acc, o, nullptr, flags, fed, lraw, fedm);
    };
    int i = 0;
    if (full > 0) { step(0); i = 1; }              // peeled: with -fno-signed-zeros the zero initialisation of acc[] folds away
    if constexpr (D == 2 && C == 3) {
        // three channels in two dimensions: interleaving two steps costs 50-60 VGPRs (d2c3k4 on 64 lanes: 196 -> 258 = one
        // wavefront per SIMD less), so these instantiations take one unguarded step per trip
        for (; i < full; ++i) step(i);
    } else {
        for (; i + 1 < full; i += 2) { step(i); step(i + 1); }
        if (i < full) step(i);
    }
    if (full < pxl) {                              // ragged tail (N not a multiple of G): guarded, flags as partial sums
        const int n = full * G + sub;
        if (n < N) {
            float x[D], t[C];
#pragma unroll
            for (int l = 0; l < D; ++l) x[l] = (l < D - HL) ? s_coords[l * N + n] : 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) t[c] = s_tgt[c * N + n];
            const float lw = HAS_LW ? s_lw[n] : 1.0f;
            PixelOut<D, C, K> o;
            pixel<D, C, K, true, HL, false, IC, false, RAWL>(R, kc, x, t, lw, acc, o, nullptr, nullptr, !(HAS_LW && sample) || lw != 0.0f, lraw);
        }
    }
    if constexpr (RAWL) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[Layout<D, C, K>::S_LOSS] = fmaf(kc.cw[c], lraw[c], acc[Layout<D, C, K>::S_LOSS]);
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) acc[Layout<D, C, K>::S_CNT + k] += ((flags[k] >> lane) & 1ull) ? 1.0f : 0.0f;
}

// HL = number of trailing axes whose index is the same for every pixel n = i*G + sub of a lane: the host
// guarantees G % (block_shape[D-1] * ... * block_shape[D-HL]) == 0.  Terms in those coordinates are
// hoisted out of the pixel loop (HL = 1 for 16x16 blocks with G = 16; HL = 2 for 16x16x4 with G = 64).
// SSIM (ssim_opt, G == 64 and D == 2 only): loss_pixel = 1 - SSIM.  Per iteration: a forward-only sweep
// leaves the quantised reconstruction of the block in LDS, the wavefront turns it into dL/dq
// (ssim_block), and the usual fused sweep runs with that gradient instead of the margin loss.
// QUANT: quantization_mode 2 / 3 (all variables fake-quantised in the graph).
// IC: train_inverse_cov (symmetric A, maha = r^T A r).
// PAIR (few blocks: at most one wavefront per SIMD otherwise; G == 64, two wavefronts per workgroup): the workgroup
// takes ONE block.  Both wavefronts run the pixel loop, on alternate steps of 64 pixels, and reduce their own
// accumulators; the second hands its raw totals over through its reduction scratch and the first alone runs the owner
// phase.  Two workgroup barriers per iteration (totals handed over / parameters written).  A lone wavefront issues one
// VALU instruction per ~5.6 cycles whatever its instruction-level parallelism (profiles/r02/ubench_valu.txt), two per
// SIMD one per ~3.2: the second wavefront is nearly free.
// Parameter-rich triples on the one-block-per-wavefront tiling: the kernel is told to stay within 256 VGPRs (two
// wavefronts per SIMD).  Left alone the compiler takes ~320 (launch bounds of 128 threads allow 512), which makes 2 040
// blocks two ROUNDS of lone wavefronts at 5.6 cycles per instruction; with the bound it parks ~60 loop-invariant values in
// scratch (a handful of reloads per pixel step) and the two wavefronts share a SIMD: 32x32 / K = 8 / RGB 76.9 -> 102.4
// Gpx-it/s, with train_inverse_cov 70 -> 94.  (Round 1 measured the same attribute as a loss, 62 -> 53: the kernel of that
// round spilled inside the pixel loop.)  On the 32-lane tiling of these 64-lane-sized blocks it loses (1 020 wavefronts are lone anyway: 88 -> 60; the
// 32-lane bound SMOE_W2_G32 is for the mid-size batches of SMALL blocks) and it is not applied to the smaller triples, whose
// kernels are at or below 256 registers or run three wavefronts per SIMD.  (The headline kernel bound to FOUR wavefronts per
// SIMD -- 128 VGPRs + 22 parked dwords -- changes nothing with four-wavefront workgroups, 385.3 vs 386.5 Gpx-it/s: 43 KB
// of LDS hold three workgroups per CU whatever the registers; with two-wavefront workgroups, seven per CU, it LOSES, 384.5 ->
// 363.8: that kernel is bound by VALU issue, and a fourth wavefront adds reloads, not issue slots.)
// (Margin-loss graph without mode-2/3 quantisation only: the quantised and SSIM variants spill twice as much under the
// bound and lose -- mode 3: 54 -> 45, mode 2: 70 -> 61, SSIM 3.5 -> 2.5 Gpx-it/s; scripts/cfg3_variants.py.)
// Hoisting kernels only: without hoisting the same triples take 420-510 VGPRs, the bound would park ~200 dwords and the
// kernel crawls (24x24 / K = 8 / RGB: 53.8 -> 16.7 Gpx-it/s); unbound, big_block_lanes sends those shapes to 32 lanes.
template <int D, int C, int K, int G, int HL, bool SSIM, bool QUANT>
constexpr int fit_min_waves() {
    return ((G == 64 || (G == 16 && SMOE_W2_G16) || (G == 32 && SMOE_W2_G32)) && HL >= 1 && !SSIM && !QUANT && Layout<D, C, K>::NSLOT >= ((G == 64) ? SMOE_W2_SLOTS : 90)
            && Layout<D, C, K>::NSLOT <= SMOE_W2_SLOTS_MAX) ? 2 : 1;
}

template <int D, int C, int K, int G, int WAVES, int HL, bool SSIM = false, bool QUANT = false, bool IC = false, bool PAIR = false>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(fit_min_waves<D, C, K, G, HL, SSIM, QUANT>()))) fit_kernel(FitArgs a) {
    using Lt = Layout<D, C, K>;
    using T = Tile<D, C, K, G, WAVES>;
    static_assert(!PAIR || (G == 64 && WAVES == 2 && !SSIM && !QUANT), "PAIR: one block on the two wavefronts of a workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int B = a.B;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G;
    const int sub = lane - grp * G;
    const int blk0 = blockIdx.x * (PAIR ? 1 : T::NB);
    const int lb = PAIR ? 0 : wave * T::BPW + grp;
    // PAIR: the wavefront that only sweeps pixels (alternating the role with the workgroup index, so that a SIMD would not
    // host two owners, measured no difference)
    const bool helper = PAIR && wave != 0;
    const int b_raw = blk0 + lb;
    const bool valid_b = b_raw < B;
    const int b = valid_b ? b_raw : B - 1;

    constexpr int CR = D - HL;                     // only the coordinates read per pixel are staged
    float* s_coords = lds + T::off_coords();
    float* s_par = lds + T::off_par(N, CR) + lb * Lt::LP_STRIDE;
    float* s_mv = lds + T::off_mv(N, CR) + lb * T::MV_STRIDE;
    float* s_scratch = lds + T::off_scratch(N, CR) + wave * (T::CH * T::ROW);
    const bool owner_post = T::OWNER_POST && a.desc_off > 0;
    float* s_desc = lds + a.desc_off + lb * T::DESC_STRIDE;              // owner_post: gradient descriptors of the block's slots
    float* s_tot = s_scratch + grp * T::TOT_STRIDE;                      // OWNER_POST: published raw totals (scratch, after the reduction)
    const float* s_tgt = lds + T::off_tgt(N, CR) + lb * T::tgt_stride(N);
    const float* s_lw = lds + T::off_lw(N, CR) + lb * T::lw_stride(N);
    const bool has_lw = a.loss_w != nullptr;

    SMOE_LDS_CHECK(T::off_ssim(N, has_lw, CR, QUANT) + (SSIM ? T::ssim_tabs(a.bh, a.bw, (D == 3) ? a.bt : 0) + WAVES * T::ssim_wave(N) : 0), 1u);
    if (owner_post) SMOE_LDS_CHECK(a.desc_off + T::NB * T::DESC_STRIDE, 2u);
    stage_inputs<D, C, K, G, WAVES, CR>(a.coords, a.target, a.loss_w, B, N, blk0, lds);
    // fake-quantised graph: the quantised parameter image of the block, its mode-3 range records, all-reduce hand-off
    float* s_q = lds + T::off_qimg(N, has_lw, CR) + lb * T::QI_STRIDE;
    float* s_rng = s_q + Lt::LP_STRIDE;
    float* s_out = s_rng + T::QI_RNG;
    // ssim_opt planes (see Tile::off_ssim)
    float* s_ssim = lds + T::off_ssim(N, has_lw, CR, QUANT);
    const int bh = a.bh, bw = a.bw, bt = (D == 3) ? a.bt : 0;
    const float* s_Tr = s_ssim;
    const float* s_Tc = s_ssim + bh * 11;
    const float* s_Tt = s_Tc + bw * 11;               // 3-d blocks: the taps of the third axis
    float* s_X = s_ssim + T::ssim_tabs(bh, bw, bt) + wave * T::ssim_wave(N) + ((G == 16) ? grp * (C * N) : 0);
    float* s_Wa = s_X + C * N;
    float* s_Wb = s_Wa + 5 * N;
    float wj[11];                                  // G == 16: this lane's row-pass weights T16[sub][sub + a - 5]
    if (SSIM) {
        if (G == 16) {
#pragma unroll
            for (int q = 0; q < 11; ++q) {
                wj[q] = a.ssim_T[11 * 16 + sub * 11 + q];
            }
        } else {
            for (int i = threadIdx.x; i < 11 * (bh + bw + bt); i += T::THREADS) s_ssim[i] = a.ssim_T[i];
        }
    }
    float xc[D];                                   // coordinates of the lane's pixel i = 0 (hoisted axes: all its pixels)
#pragma unroll
    for (int l = 0; l < D; ++l) xc[l] = a.coords[l * N + min(sub, N - 1)];

    // ---- owner set-up: this lane owns packed slots sub, sub+G, ... of its block --------
    // per-slot learning rate (0 = not trained) and l1 regulariser constant stay in registers;
    // the parameter and its Adam slots live in LDS between iterations.
    // fixed-range fake quant of the slot (pis; everything in mode 2): nudged range + step; qt[s] = index of the slot's
    // mode-3 tensor (0 A_diagonal, 1 A_corr, 2 musX, 3 nu_e, 4 gamma_e; -1 none)
    // meta[s] = tensor | kernel << 4 of the slot (-1: not a parameter).  The per-slot range arrays exist in the QUANT
    // instantiations only: kept live across the loop they cost the other kernels ~20 VGPRs (the SSIM kernel lost a
    // wave per SIMD over them); quantize_pis alone needs just the uniform constants of the pis range.
    constexpr int QS = QUANT ? T::SPL : 1;
    float lr[T::SPL], reg[T::SPL], qlo[QS], qhi[QS], qsc[QS], qiv[QS];
    float goff[QS];          // use_diff_center: the kernel-grid centre of a musX slot (the quantised variable is musX - grid), else 0
    int meta[T::SPL], qt[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        qlo[s] = -__builtin_huge_valf(); qhi[s] = __builtin_huge_valf();
        qsc[s] = qiv[s] = 0.0f;
        qt[s] = -1;
        goff[s] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < T::SPL; ++s) {
        const int j = T::slot_of(sub, s);
        lr[s] = reg[s] = 0.0f;
        meta[s] = -1;
        if (j < Lt::NPAR) {
            int tensor, kern; long off;
            decode_slot<D, C, K>(j, b, tensor, off, kern);
            // (the coalesced tile path of the evaluation kernel -- TileIO -- was measured here too: three LDS hand-offs for
            // parameters, m and v cost more than these scattered loads of lines the 16 blocks share: -2 % at 20 iterations
            // per launch, -0.9 % at 100)
            s_par[j] = pick(a.p, tensor)[off];
            s_mv[2 * j] = pick(a.m, tensor)[off];
            s_mv[2 * j + 1] = pick(a.v, tensor)[off];
            // optimizer groups, smoe.py:1102-1104; untrainable variables dropped, 1112-1117
            float r = (tensor == 0) ? a.lr_pis : ((tensor == 2 || tensor == 3) ? a.lr_steer : a.lr_expert);
            if (tensor == 0 && !a.train_pis) r = 0.0f;
            if (tensor == 1 && !a.train_musx) r = 0.0f;
            if (tensor == 4 && !a.kc.train_gammas) r = 0.0f;
            if (tensor == 4 && a.kc.only_y_gamma && (off % C) != 0) r = 0.0f;   // masked slopes get zero gradient
            if (tensor == 3 && a.kc.radial) r = 0.0f;                           // radial_as: A_corr is not trainable (smoe.py:434)
            lr[s] = r;
            // smoe.py:1027,1044; radial_as: u_l1 * sum(diag A) = u_l1 * d * a, and the d tied slots each carry dL/da
            reg[s] = (tensor == 0) ? a.reg_pi : ((tensor == 2) ? (a.kc.radial ? a.reg_u * (float)D : a.reg_u) : 0.0f);
            // fixed-range fake quant of this variable: the gradient passes inside the nudged range only
            const int qg = (tensor == 0) ? 3 : ((tensor == 1) ? 1 : ((tensor == 4) ? 4 : ((tensor == 5) ? 2 : 0)));
            meta[s] = tensor | (kern << 4);
            if constexpr (QUANT) {
                if (tensor == 1 && a.mus_grid != nullptr && a.kc.qmode >= 2) goff[s] = a.mus_grid[off];
                if ((tensor == 0 && a.kc.qpis) || (tensor != 0 && a.kc.qmode == 2)) {
                    qlo[s] = a.kc.q_nmin[qg]; qhi[s] = a.kc.q_nmax[qg]; qsc[s] = a.kc.q_scale[qg]; qiv[s] = a.kc.q_inv[qg];
                }
                if (a.kc.qmode == 3 && tensor != 0 && !(tensor == 1 && !a.kc.q_musx))
                    qt[s] = (tensor == 2) ? 0 : ((tensor == 3) ? 1 : ((tensor == 1) ? 2 : ((tensor == 5) ? 3 : 4)));
            } else {
                (void)qg;
            }
        } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
            const int k = j - Lt::S_CNT;
            s_par[Lt::LP_ACT + k] = ((a.active[b] >> k) & 1u) ? 1.0f : 0.0f;
        } else if (j == Lt::S_LOSS) {
            s_par[Lt::LP_FROZEN] = (a.diverged != nullptr && a.diverged[b] != 0u) ? 1.0f : 0.0f;
        }
    }
    const float loss0 = (a.loss0 != nullptr) ? a.loss0[b] : 0.0f;
    const bool has_loss0 = a.loss0 != nullptr;
    const bool has_reg = (a.reg_pi != 0.0f) || (a.reg_u != 0.0f);
    const bool has_quant = (a.kc.qmode != 0) || (a.kc.qpis != 0);
    float last_loss = 0.0f, last_sse = 0.0f;
    __syncthreads();

    // ---- fake-quantised graph: (re)build the block's quantised image from the raw parameters in s_par.  Runs once here
    // and after every Adam step; the forward / backward then read s_q and never quantise themselves.  Mode 3: the ranges
    // are the min / max over the kernels with qpis > 0 (smoe.py:497-530), found with one block-wide all-reduce; lane
    // t < 5 nudges range t (TF Nudge()) and publishes it; every owner quantises its own slots.
    auto pick5 = [](const float (&v)[10], int base, int t) {
        return (t == 0) ? v[base] : ((t == 1) ? v[base + 1] : ((t == 2) ? v[base + 2] : ((t == 3) ? v[base + 3] : v[base + 4])));
    };
    auto refresh_quantised_image = [&]() {
        if constexpr (QUANT) {
            if (a.kc.qmode == 3) {
                constexpr float INF = __builtin_huge_valf();
                float ex[10];                              // lo[0..4], -hi[5..9] of this lane's slots
#pragma unroll
                for (int i = 0; i < 10; ++i) ex[i] = INF;
                const FqRange rp = fq_fixed(a.kc, 3);
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    const int j = T::slot_of(sub, s);
                    if (j < Lt::NPAR && qt[s] >= 0) {
                        const bool keep = fq_val(s_par[(meta[s] >> 4) * Lt::PK + Lt::O_PI], rp) > 0.0f;      // pis_mask = qpis > 0
                        const float x = s_par[j] - goff[s];
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            const bool hit = keep && (qt[s] == t);
                            ex[t] = hit ? fminf(ex[t], x) : ex[t];
                            ex[5 + t] = hit ? fminf(ex[5 + t], -x) : ex[5 + t];
                        }
                    }
                }
                block_allreduce10<G, T::ROW, true>(ex, s_scratch, s_out, lane, grp, sub);
                {
                    const int t = min(sub, 4);
                    float lo = pick5(ex, 0, t), hi = -pick5(ex, 5, t);
                    if (lo == INF) { lo = 0.0f; hi = 0.0f; }                       // no kernel left
                    if (t == 1) { lo = fminf(lo, 0.0f); hi = fmaxf(hi, 0.0f); }    // structural zeros of the A_corr variable
                    const float lv = (t < 2) ? a.kc.q_levels[0] : ((t == 2) ? a.kc.q_levels[1] : ((t == 3) ? a.kc.q_levels[2] : a.kc.q_levels[4]));
                    const FqRange r = fq_vars(lo, hi, lv, t == 0 || t == 3, t == 0 && a.kc.radial != 0);
                    if (sub < 5) {
                        float* o = s_rng + t * 8;
                        o[0] = r.nmin; o[1] = r.nmax; o[2] = r.scale; o[3] = r.inv;
                        o[4] = r.back; o[5] = r.zero ? 1.0f : 0.0f; o[6] = lo; o[7] = hi;
                    }
                }
                wave_lds_sync();
            }
        }
        if constexpr (!QUANT) {
            // quantize_pis alone (the CLI default): the K quantised pis live in K extra floats of the block's image
#pragma unroll
            for (int s = 0; s < T::SPL; ++s) {
                const int j = T::slot_of(sub, s);
                if (meta[s] >= 0 && (meta[s] & 15) == 0) {          // a pis slot (this branch runs with kc.qpis only)
                    const float x = s_par[j];
                    const float cl = fminf(fmaxf(x, a.kc.q_nmin[3]), a.kc.q_nmax[3]);
                    s_par[Lt::LP_QPI + (meta[s] >> 4)] = floorf((cl - a.kc.q_nmin[3]) * a.kc.q_inv[3] + 0.5f) * a.kc.q_scale[3] + a.kc.q_nmin[3];
                }
            }
            wave_lds_sync();
            return;
        }
        if constexpr (QUANT) {
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j < Lt::NPAR) {
                const float x = s_par[j] - goff[s];
                float q = x;
                if (qsc[s] != 0.0f) {                       // fixed range (pis; mode 2)
                    const float cl = fminf(fmaxf(x, qlo[s]), qhi[s]);
                    q = floorf((cl - qlo[s]) * qiv[s] + 0.5f) * qsc[s] + qlo[s];
                }
                {
                    if (qt[s] >= 0) {
                        const float* o = s_rng + qt[s] * 8;
                        FqRange r;
                        r.nmin = o[0]; r.nmax = o[1]; r.scale = o[2]; r.inv = o[3]; r.back = o[4]; r.zero = o[5] != 0.0f;
                        r.shift = (qt[s] == 0 && a.kc.radial) ? 0.0f : r.back;       // radial_as: unshifted input (fq_vars)
                        q = fq_val(x, r);
                    }
                }
                s_q[j] = q + goff[s];
            } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
                s_q[Lt::LP_ACT + (j - Lt::S_CNT)] = s_par[Lt::LP_ACT + (j - Lt::S_CNT)];
            } else if (j == Lt::S_LOSS) {
                s_q[Lt::LP_FROZEN] = s_par[Lt::LP_FROZEN];
            }
        }
        }
        wave_lds_sync();
    };
    if (has_quant) refresh_quantised_image();
    const float* s_img = (QUANT && has_quant) ? s_q : s_par;      // what the graph is built on (QUANT: the quantised image)
    const bool patch_pis = !QUANT && (a.kc.qpis != 0);               // default kernels: quantised pis from the K extra floats
    if (owner_post && !helper) {
        // one descriptor per packed parameter of the block (byte offsets from the start of the LDS), built by its owner
        // (PAIR: by the owning wavefront only -- the descriptors point into ITS published totals)
        const uint32_t tot_off = (uint32_t)((s_tot - lds) * sizeof(float));
        const uint32_t img_off = (uint32_t)((s_img - lds) * sizeof(float));
        const uint32_t par_off = (uint32_t)((s_par - lds) * sizeof(float));
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j < Lt::NPAR) {
                const SlotDesc d = build_slot_desc<D, C, K, IC>(j, tot_off, img_off, par_off, patch_pis, a.kc.use_det != 0);
                float4* o = reinterpret_cast<float4*>(s_desc + j * T::DESC_DW);
                o[0] = make_float4(d.c_self, d.pub_scale, __uint_as_float(d.T1), __uint_as_float(d.T2));
                o[1] = make_float4(__uint_as_float(d.T3), __uint_as_float(d.TR), __uint_as_float(d.P1), __uint_as_float(d.P2));
                o[2] = make_float4(__uint_as_float(d.P3), __uint_as_float(d.PR), __uint_as_float(d.FLAG), __uint_as_float(d.PIV));
            } else if (j == Lt::S_LOSS) {
                s_par[Lt::LP_ZERO] = 0.0f;
                s_par[Lt::LP_ONE] = 1.0f;
            }
        }
        wave_lds_sync();
    }

    float b1p = a.b1p, b2p = a.b2p;
    const KernelConsts kc = a.kc;
    const float beta1 = a.beta1, beta2 = a.beta2, adam_eps = a.eps, clip = a.clip;
    const float reg_pi = a.reg_pi, reg_u = a.reg_u;

#if SMOE_PHASE_CLOCKS
    float clk[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    unsigned long long clk_last = __builtin_amdgcn_s_memtime();
#endif
    const bool phase_prio = !PAIR && (G < 64 || N >= 8 * G);       // (a compile-time constant on the 16- and 32-lane tilings)
    for (int it = 0; it < a.n_iters; ++it) {
        // wavefront priority of the iteration's first part (parameter image, derived constants, pixel loop): the rotating level
        // in a one-round launch (rotate_priority), the lowest otherwise; the second part runs at the highest (below)
        if (a.prio_rotate) rotate_priority(it, hw_wave_slot());
        else if (phase_prio) __builtin_amdgcn_s_setprio(0);
        float acc[Lt::NSLOT];
#pragma unroll
        for (int j = 0; j < Lt::NSLOT; ++j) acc[j] = 0.0f;
        bool frozen;
        float reg_loss = 0.0f;
        {
            BlockRegs<D, C, K> R;
            R.load(s_img);                                  // the graph sees the fake-quantised variables
            if (patch_pis) {
#pragma unroll
                for (int k = 0; k < K; ++k) R.P[k * Lt::PK + Lt::O_PI] = R.P[Lt::LP_QPI + k];
            }
            R.template derive<IC>(kc);
            frozen = R.frozen();
            if (has_reg) {                                  // smoe.py:1027,1044 (active kernels only)
                const float rp = kc.kcount_norm ? kc.pis_l1_raw / count_pis<D, C, K>(R.P) : reg_pi;   // smoe.py:1022-1023
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (R.act(k)) {
                        reg_loss += rp * R.pi(k);
#pragma unroll
                        for (int l = 0; l < D; ++l) reg_loss += reg_u * R.A(k, l, l);
                    }
                }
            }
            if (HL > 0) hoist_const<D, C, K, HL, IC>(R, xc);
#if SMOE_SGPR_CONSTS
            // one block per wavefront: what the pixel loop reads of the parameters besides the lane's hoisted constants is the
            // same in every lane -- scalar registers for the parameter-rich triples (one scalar operand per VALU instruction)
            if constexpr (G == 64 && !SSIM && !QUANT && !IC && HL >= 1 && K * C >= 24) {
                auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    R.coef[k] = uni(R.coef[k]);
#pragma unroll
                    for (int l = 0; l < D - HL; ++l) {
#pragma unroll
                        for (int m = 0; m <= l; ++m) R.As[k][tri_index(l, m)] = uni(R.As[k][tri_index(l, m)]);
#pragma unroll
                        for (int c = 0; c < C; ++c) R.P[k * Lt::PK + Lt::O_GA + l * C + c] = uni(R.P[k * Lt::PK + Lt::O_GA + l * C + c]);
                    }
                }
            }
#endif
            SMOE_CLK(0);
            if constexpr (SSIM) {
                // the reference's SSIM branch does not use loss_weights (smoe.py:929-1010)
                // Both sweeps, 16-lane tiling: the full pixel steps run unguarded (uniform control flow), a ragged tail guarded.
                // The wavefront-per-block instantiations keep every step guarded (their three-channel versions lose up to a
                // third to the extra register pressure otherwise).
                const int pxl = (N + G - 1) / G;
                const int full = (G == 16) ? N / G : 0;
                auto recon_step = [&](int n) {                  // sweep 1: reconstruction only
                    float x[D], t[C], scratch_acc[Lt::NSLOT];
#pragma unroll
                    for (int l = 0; l < D; ++l) x[l] = (l < D - HL) ? s_coords[l * N + n] : 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) t[c] = s_tgt[c * N + n];
#pragma unroll
                    for (int j = 0; j < Lt::NSLOT; ++j) scratch_acc[j] = 0.0f;
                    PixelOut<D, C, K> o;
                    pixel<D, C, K, false, HL, false, IC>(R, kc, x, t, 1.0f, scratch_acc, o);
#pragma unroll
                    for (int c = 0; c < C; ++c) s_X[c * N + n] = o.q[c];
                };
                for (int i = 0; i < full; ++i) recon_step(i * G + sub);
                for (int i = full; i < pxl; ++i)
                    if (i * G + sub < N) recon_step(i * G + sub);
                wave_lds_sync();
                if constexpr (G == 16) acc[Lt::S_LOSS] = ssim_block16<C, true>(s_X, s_tgt, sub, wj, kc.sw);
                else if constexpr (D == 3) acc[Lt::S_LOSS] = ssim_block3<C, true>(s_X, s_tgt, s_Wa, s_Wb, s_Tr, s_Tc, s_Tt, kc.sw, bh, bw, bt, N, lane);
                else acc[Lt::S_LOSS] = ssim_block<C, true>(s_X, s_tgt, s_Wa, s_Wb, s_Tr, s_Tc, kc.sw, bh, bw, N, lane);
                // sweep 2: forward again + backward with dL/dq; influence flags as scalar lane-mask votes in the full steps
                unsigned long long flags[K];
#pragma unroll
                for (int k = 0; k < K; ++k) flags[k] = 0ull;
                auto grad_step = [&](int n, auto voted) {
                    float x[D], t[C], gq[C];
#pragma unroll
                    for (int l = 0; l < D; ++l) x[l] = (l < D - HL) ? s_coords[l * N + n] : 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { t[c] = s_tgt[c * N + n]; gq[c] = s_X[c * N + n]; }
                    PixelOut<D, C, K> o;
                    pixel<D, C, K, true, HL, true, IC, decltype(voted)::value>(R, kc, x, t, 1.0f, acc, o, gq, flags);
                };
                for (int i = 0; i < full; ++i) grad_step(i * G + sub, std::true_type{});
                for (int i = full; i < pxl; ++i)
                    if (i * G + sub < N) grad_step(i * G + sub, std::false_type{});
#pragma unroll
                for (int k = 0; k < K; ++k) acc[Lt::S_CNT + k] += ((flags[k] >> lane) & 1ull) ? 1.0f : 0.0f;
            } else {
                // PAIR: pixel n = i * 128 + wave * 64 + sub (the hoisted trailing coordinates stay lane constants)
                const int gl = PAIR ? 2 * G : G, sl = PAIR ? wave * G + sub : sub;
                if (has_lw) pixel_loop_train<D, C, K, true, HL, IC>(R, kc, s_coords, s_tgt, s_lw, N, gl, sl, acc, a.lw_is_sample != 0);
                else pixel_loop_train<D, C, K, false, HL, IC>(R, kc, s_coords, s_tgt, s_lw, N, gl, sl, acc);
            }
            if (HL > 0) complete_const<D, C, K, HL, IC>(R, xc, acc);
            SMOE_CLK(1);
        }
        // The reduction and owner phases are chains of LDS round trips with few instructions; the other wavefronts of the SIMD are
        // mostly in their pixel loops and issue continuously.  Served FIRST (s_setprio 3 until the next iteration starts) the
        // chain is not held up behind them: 65 536 blocks 396 -> 406 Gpx-it/s, 12 288: 369 -> 387, 4 096 (32 lanes): 255 -> 260.
        // (Two rotating levels per part, as in the duo kernel, for launches with two wavefronts per SIMD: 4 096 blocks -5 %, 8 192: -3 %.)
        // Only where the pixel loop is the larger part of the iteration (at least eight pixels per lane): a 16x16 block on one
        // wavefront spends most of its iteration IN this part (2 048 blocks: -3 %).
        if (phase_prio) __builtin_amdgcn_s_setprio(3);
        if (!owner_post) {
            BlockRegs<D, C, K> R2;                           // re-read mu, A, pi (not kept live over the pixel loop)
            R2.load(s_img);
            if (patch_pis) {
#pragma unroll
                for (int k = 0; k < K; ++k) R2.P[k * Lt::PK + Lt::O_PI] = R2.P[Lt::LP_QPI + k];
            }
            finish_partials<D, C, K, IC>(R2, kc, acc);
        }

        SMOE_CLK(2);
        float total[T::SPL];
        reduce_slots<D, C, K, G, WAVES, 0>(acc, s_scratch, lane, total);
        SMOE_CLK(3);

        if constexpr (PAIR) {
            // the helper's totals (linear partial sums, like the accumulators) go to the owners through its own scratch
            float* s_x = lds + T::off_scratch(N, CR) + (T::CH * T::ROW);
            if (helper) {
                wave_lds_sync();                           // its transpose reads are done
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    const int j = T::slot_of(sub, s);
                    if (j < Lt::NSLOT) s_x[j] = total[s];
                }
            }
            __syncthreads();
            if (helper) {
                __syncthreads();                           // the owners' Adam step: parameters written
                b1p *= beta1;
                b2p *= beta2;
                SMOE_CLK(4);
                continue;
            }
#pragma unroll
            for (int s = 0; s < T::SPL; ++s) {
                const int j = T::slot_of(sub, s);
                if (j < Lt::NSLOT) total[s] += s_x[j];
            }
        }

        SMOE_CLK(4);
        if (owner_post) {
            // raw sums -> gradients on the owner side: publish the raw totals in the (now free) scratch, then every owner
            // evaluates the descriptor of its slot (eval_slot_desc)
            SlotDesc dsc[T::SPL];
#pragma unroll
            for (int s = 0; s < T::SPL; ++s) {
                int j = T::slot_of(sub, s);
                asm volatile("" : "+v"(j));                  // keep the descriptor address out of the registers live across the loop
                const int jc = (j < Lt::NPAR) ? j : 0;
                const float4* q = reinterpret_cast<const float4*>(s_desc + jc * T::DESC_DW);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                dsc[s].c_self = q0.x; dsc[s].pub_scale = q0.y; dsc[s].T1 = __float_as_uint(q0.z); dsc[s].T2 = __float_as_uint(q0.w);
                dsc[s].T3 = __float_as_uint(q1.x); dsc[s].TR = __float_as_uint(q1.y); dsc[s].P1 = __float_as_uint(q1.z); dsc[s].P2 = __float_as_uint(q1.w);
                dsc[s].P3 = __float_as_uint(q2.x); dsc[s].PR = __float_as_uint(q2.y); dsc[s].FLAG = __float_as_uint(q2.z); dsc[s].PIV = __float_as_uint(q2.w);
                if (j < Lt::NPAR) s_tot[j] = total[s] * dsc[s].pub_scale;
            }
            wave_lds_sync();
#pragma unroll
            for (int s = 0; s < T::SPL; ++s) {
                const int j = T::slot_of(sub, s);
                if (j < Lt::NPAR) total[s] = eval_slot_desc(lds, dsc[s], total[s]);
            }
            if (kc.radial) {
                // radial_as (smoe.py:714-719): every diagonal slot of a kernel receives the trace of dL/dA
                wave_lds_sync();
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    const int j = T::slot_of(sub, s);
                    if (j < Lt::NPAR) s_tot[j] = total[s];
                }
                wave_lds_sync();
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    const int j = T::slot_of(sub, s);
                    if (j < Lt::NPAR) {
                        const int k = j / Lt::PK, o = j - k * Lt::PK;
                        bool diag = false;
#pragma unroll
                        for (int l = 0; l < D; ++l) diag = diag || (o == Lt::O_A + tri_index(l, l));
                        if (diag) {
                            float tr = 0.0f;
#pragma unroll
                            for (int l = 0; l < D; ++l) tr += s_tot[k * Lt::PK + Lt::O_A + tri_index(l, l)];
                            total[s] = tr;
                        }
                    }
                }
            }
            wave_lds_sync();
        }

        SMOE_CLK(5);
        // ---- owner phase: TF1 ApplyAdam (smoe.py:1173-1193), prune (1763-1766), stop test (1565-1570)
        const float bias = __builtin_amdgcn_sqrtf(1.0f - b2p) * fast_rcp(1.0f - b1p);   // alpha = lr * sqrt(1-b2^t)/(1-b1^t): one division per iteration
        // gradients w.r.t. the (quantised) graph variables incl. the l1 terms
        float gq[T::SPL];
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            const int jc = (j < Lt::NPAR) ? j : 0;
            float gsum = total[s];
            if (has_reg && reg[s] != 0.0f) {
                const int k = jc / Lt::PK;
                const float piv = patch_pis ? s_par[Lt::LP_QPI + k] : s_img[k * Lt::PK + Lt::O_PI];
                const bool act = (s_par[Lt::LP_ACT + k] != 0.0f) && (piv > 0.0f);
                float rs = reg[s];
                if (kc.kcount_norm && (jc - k * Lt::PK) == Lt::O_PI) {          // pis_l1 / count(qpis > 0), smoe.py:1022-1027
                    float cnt = 0.0f;
#pragma unroll
                    for (int kk = 0; kk < K; ++kk) cnt += ((patch_pis ? s_par[Lt::LP_QPI + kk] : s_img[kk * Lt::PK + Lt::O_PI]) > 0.0f) ? 1.0f : 0.0f;
                    rs = kc.pis_l1_raw / fmaxf(cnt, 1.0f);
                }
                gsum += act ? rs : 0.0f;
            }
            gq[s] = gsum;
        }
        if constexpr (QUANT) {
            if (kc.qmode == 3) {
                // back through fake_quant_with_min_max_vars (+ reduce_min / reduce_max): inside the nudged range the
                // gradient passes; what falls below / above goes to the extreme elements of the tensor, split over ties
                float rs[10], cn[10];                      // GL[0..4] GA[5..9] ; tie counts at lo / hi
                bool bel[T::SPL], abv[T::SPL], tlo[T::SPL], thi[T::SPL];
#pragma unroll
                for (int i = 0; i < 10; ++i) { rs[i] = 0.0f; cn[i] = 0.0f; }
                const FqRange rp = fq_fixed(kc, 3);
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    const int j = T::slot_of(sub, s);
                    bel[s] = abv[s] = tlo[s] = thi[s] = false;
                    if (j < Lt::NPAR && qt[s] >= 0) {
                        const float* o = s_rng + qt[s] * 8;
                        const bool unshifted = qt[s] == 0 && kc.radial;       // radial_as steering (fq_vars, noshift)
                        const float x = s_par[j] - goff[s], v = unshifted ? x : x - o[4];
                        const bool zero = o[5] != 0.0f;
                        const bool keep = fq_val(s_par[(meta[s] >> 4) * Lt::PK + Lt::O_PI], rp) > 0.0f;
                        bel[s] = !zero && (v < o[0]);
                        abv[s] = !zero && (v > o[1]);
                        tlo[s] = keep && (x == o[6]);
                        thi[s] = keep && (x == o[7]);
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            const bool hit = qt[s] == t;
                            // unshifted: the lower end of the range is the constant 0 (below it: lost), and the added minimum
                            // collects sum(g) while max - min takes sum(g * above) off it: the minimum receives sum(g * !above)
                            rs[t] += (hit && (unshifted ? !abv[s] : bel[s])) ? gq[s] : 0.0f;
                            rs[5 + t] += (hit && abv[s]) ? gq[s] : 0.0f;
                            cn[t] += (hit && tlo[s]) ? 1.0f : 0.0f;
                            cn[5 + t] += (hit && thi[s]) ? 1.0f : 0.0f;
                        }
                    }
                }
                block_allreduce10<G, T::ROW, false>(rs, s_scratch, s_out, lane, grp, sub);
                block_allreduce10<G, T::ROW, false>(cn, s_scratch, s_out, lane, grp, sub);
#pragma unroll
                for (int s = 0; s < T::SPL; ++s) {
                    if (qt[s] >= 0) {
                        float g = (bel[s] || abv[s]) ? 0.0f : gq[s];
                        g += tlo[s] ? pick5(rs, 0, qt[s]) / fmaxf(pick5(cn, 0, qt[s]), 1.0f) : 0.0f;
                        g += thi[s] ? pick5(rs, 5, qt[s]) / fmaxf(pick5(cn, 5, qt[s]), 1.0f) : 0.0f;
                        gq[s] = g;
                    }
                }
            }
        }
        // The old value and both Adam slots of every owned slot are read first, in one batch (one LDS wait for the phase instead
        // of one per slot), the update is branch-free, and the new slots are stored together with the new parameters below.
        float newp[T::SPL], newm[T::SPL], newv[T::SPL], pvs[T::SPL], mvs[T::SPL], vvs[T::SPL];
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            const int jc = (j < Lt::NPAR) ? j : 0;
            pvs[s] = s_par[jc];
            mvs[s] = s_mv[2 * jc];
            vvs[s] = s_mv[2 * jc + 1];
        }
        bool bad = false;
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            const float pv = pvs[s], mv = mvs[s], vv = vvs[s];
            float gsum = gq[s];
            // fixed-range fake quant: straight-through inside the nudged range only
            if constexpr (QUANT) {
                gsum = (pv - goff[s] >= qlo[s] && pv - goff[s] <= qhi[s]) ? gsum : 0.0f;
            } else {
                if (patch_pis && meta[s] >= 0 && (meta[s] & 15) == 0) gsum = (pv >= kc.q_nmin[3] && pv <= kc.q_nmax[3]) ? gsum : 0.0f;
            }
            if (clip > 0.0f) gsum = fminf(fmaxf(gsum, -clip), clip);
            const float alpha = lr[s] * bias;
            const float m2 = mv + (gsum - mv) * (1.0f - beta1);
            const float v2 = vv + (gsum * gsum - vv) * (1.0f - beta2);
            // hardware sqrt / rcp (1 ulp each): the step changes by ~2e-7 relative, an IEEE sqrt + division per slot cost ~14 VALU
            const float p2 = pv - (m2 * alpha) * fast_rcp(__builtin_amdgcn_sqrtf(v2) + adam_eps);
            const bool upd = (j < Lt::NPAR) && (lr[s] != 0.0f) && !frozen;
            newp[s] = upd ? p2 : pv;
            newm[s] = upd ? m2 : mv;
            newv[s] = upd ? v2 : vv;
            const float lossv = (SSIM ? 1.0f + total[s] : total[s]) + reg_loss;         // smoe.py:1010: 1 - ssim
            const bool is_loss = (j == Lt::S_LOSS) && !frozen;
            last_loss = is_loss ? lossv : last_loss;
            bad = bad || (is_loss && ((lossv != lossv) || (has_loss0 && (lossv + 1.0f > (loss0 + 100.0f) * 10.0f))));
            last_sse = ((j == Lt::S_SSE) && !frozen) ? total[s] : last_sse;
        }
        wave_lds_sync();   // every lane has consumed the old flags / params
        SMOE_CLK(6);
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j < Lt::NPAR) {
                s_par[j] = newp[s];
                s_mv[2 * j] = newm[s];
                s_mv[2 * j + 1] = newv[s];
                if constexpr (!QUANT) {
                    // quantize_pis alone (the CLI default): the owner of a prior publishes its fake-quantised value with the new
                    // prior itself (one LDS hand-off less per iteration than a separate refresh_quantised_image pass)
                    if (patch_pis && meta[s] >= 0 && (meta[s] & 15) == 0) {
                        const float cl = fminf(fmaxf(newp[s], kc.q_nmin[3]), kc.q_nmax[3]);
                        s_par[Lt::LP_QPI + (meta[s] >> 4)] = floorf((cl - kc.q_nmin[3]) * kc.q_inv[3] + 0.5f) * kc.q_scale[3] + kc.q_nmin[3];
                    }
                }
            } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
                if (!frozen) s_par[Lt::LP_ACT + (j - Lt::S_CNT)] = (total[s] > 0.0f) ? 1.0f : 0.0f;
            } else if (j == Lt::S_LOSS) {
                if (bad) s_par[Lt::LP_FROZEN] = 1.0f;      // takes effect from the next iteration
            }
        }
        wave_lds_sync();
        if constexpr (QUANT) {
            if (has_quant) refresh_quantised_image();
        }
        if constexpr (PAIR) __syncthreads();
        b1p *= beta1;
        b2p *= beta2;
        SMOE_CLK(7);
    }
#if SMOE_PHASE_CLOCKS
    if (blockIdx.x == 0 && lane == 0 && a.loss_out != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.loss_out[B / 2 + wave * 8 + i] = clk[i];
    }
    return;
#endif

    // ---- write back (pointers are re-read from the kernarg segment: keeping 18 of them
    // live across the iteration loop costs SGPR spills inside it) ------------------------
    const FitArgs* ka = (const FitArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    if (valid_b && !helper) {
        int bo = b;
        asm volatile("" : "+v"(bo));       // not the prologue's slot offsets kept live across the iteration loop (4 VGPRs)
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j < Lt::NPAR) {
                int tensor, kern; long off;
                decode_slot<D, C, K>(j, bo, tensor, off, kern);
                pick(ka->p, tensor)[off] = s_par[j];
                pick(ka->m, tensor)[off] = s_mv[2 * j];
                pick(ka->v, tensor)[off] = s_mv[2 * j + 1];
            } else if (j == Lt::S_LOSS) {
                if (ka->loss_out != nullptr && ka->n_iters > 0) ka->loss_out[b] = last_loss;
                if (ka->diverged != nullptr) ka->diverged[b] = (s_par[Lt::LP_FROZEN] != 0.0f) ? 1u : 0u;
                uint32_t mask = 0u;
#pragma unroll
                for (int k = 0; k < K; ++k) mask |= (s_par[Lt::LP_ACT + k] != 0.0f) ? (1u << k) : 0u;
                ka->active[b] = mask;
            } else if (j == Lt::S_SSE) {
                if (ka->sse_out != nullptr && ka->n_iters > 0) ka->sse_out[b] = last_sse;
            }
        }
    }
}

// Output planes are written once and not read again by the kernel: 16-byte non-temporal stores (no L2 write-allocate
// for lines nobody will hit): all outputs of the grayscale headline batch 99 -> 71 us (4.4 -> 6.1 TB/s).  Measured and NOT
// kept: non-temporal DWORD stores on the three-channel path (170 -> 201 us) and non-temporal loads of the targets
// (three-channel loss-only pass 66 -> 76 us).
typedef float smoe_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float* __restrict__ p, const float4& v) {
#if SMOE_NT_STORES
    smoe_f4 q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<smoe_f4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// ---------------------------------------------------------------------------
// forward (evaluation) kernel
// ---------------------------------------------------------------------------
// HL: hoisting level as in fit_kernel (the launcher passes the same rule); the SSIM kernels run with HL = 0.
// OM: 0 = one kernel for both kinds of launch; 1 = loss-only launches, 2 = launches with per-pixel outputs (the plain
// margin-loss kernels: each kind gets its own register allocation, so the four kept steps of the grouped 16-byte stores
// no longer cost the loss-only pass a wavefront per SIMD and the three-channel kernels can group their stores too).
template <int D, int C, int K, int G, int WAVES, bool SSIM = false, bool QUANT = false, bool IC = false, int HL = 0, int OM = 0>
__global__ void __launch_bounds__(WAVES * 64) forward_kernel(FwdArgs a) {
    using Lt = Layout<D, C, K>;
    using T = Tile<D, C, K, G, WAVES>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int B = a.B;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G;
    const int sub = lane - grp * G;
    const int blk0 = blockIdx.x * T::NB;
    const int lb = wave * T::BPW + grp;
    const int b_raw = blk0 + lb;
    const bool valid_b = b_raw < B;
    const int b = valid_b ? b_raw : B - 1;

    constexpr int CR = D - HL;                     // only the coordinates read per pixel are staged
    float* s_coords = lds + T::off_coords();
    float* s_par = lds + T::off_par(N, CR) + lb * Lt::LP_STRIDE;
    float* s_scratch = lds + T::off_scratch(N, CR) + wave * (T::CH * T::ROW);
    const float* s_tgt = lds + T::off_tgt(N, CR) + lb * T::tgt_stride(N);
    const float* s_lw = lds + T::off_lw(N, CR) + lb * T::lw_stride(N);
    const bool has_lw = a.loss_w != nullptr;

    using IO = TileIO<D, C, K, T::NB, T::THREADS>;
    float vp[IO::MAXE];
    IO::fetch(a.p, blk0, B, vp);                   // in flight together with the staging loads
    // Every target is read once: with at most FWD_PXR pixels per lane (a.regt, set by the launcher) a lane takes its
    // targets / loss weights straight from global memory into registers -- all loads of the workgroup are in flight
    // together with the parameter tile, and without the staged planes the LDS holds 5+ workgroups per CU
    constexpr int PXR = T::FWD_PXR;
    const bool regt = !SSIM && (a.regt != 0);
    float tr[PXR][C], lwr[PXR];
    if (regt) {
#pragma unroll
        for (int i = 0; i < PXR; ++i) {
            const int n = min(i * G + sub, N - 1);
#pragma unroll
            for (int c = 0; c < C; ++c) tr[i][c] = a.target[((size_t)b * C + c) * N + n];
            lwr[i] = has_lw ? a.loss_w[(size_t)b * N + n] : 1.0f;
        }
    }
    SMOE_LDS_CHECK(regt ? T::off_tgt(N, CR)
                        : T::off_ssim(N, has_lw, CR) + (SSIM ? T::ssim_tabs(a.bh, a.bw, (D == 3) ? a.bt : 0) + WAVES * T::ssim_wave(N) : 0), 4u);
    stage_inputs<D, C, K, G, WAVES, CR>(a.coords, a.target, a.loss_w, B, N, blk0, lds, !regt);
    float* s_ssim = lds + T::off_ssim(N, has_lw, CR);
    const int bh = a.bh, bw = a.bw, bt = (D == 3) ? a.bt : 0;
    const float* s_Tr = s_ssim;
    const float* s_Tc = s_ssim + bh * 11;
    const float* s_Tt = s_Tc + bw * 11;               // 3-d blocks: the taps of the third axis
    float* s_X = s_ssim + T::ssim_tabs(bh, bw, bt) + wave * T::ssim_wave(N) + ((G == 16) ? grp * (C * N) : 0);
    float* s_Wa = s_X + C * N;
    float* s_Wb = s_Wa + 5 * N;
    float wj[11];
    if (SSIM) {
        if (G == 16) {
#pragma unroll
            for (int q = 0; q < 11; ++q) {
                wj[q] = a.ssim_T[11 * 16 + sub * 11 + q];
            }
        } else {
            for (int i = threadIdx.x; i < 11 * (bh + bw + bt); i += T::THREADS) s_ssim[i] = a.ssim_T[i];
        }
    }
    // parameters: one coalesced tile load through the reduction scratch, the owners pick their packed slots
    {
        float* s_tile = lds + T::off_scratch(N, CR);
        IO::put(vp, s_tile);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j < Lt::NPAR) {
                s_par[j] = s_tile[tile_index<D, C, K, T::NB>(j, lb)];
            } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
                const int k = j - Lt::S_CNT;
                s_par[Lt::LP_ACT + k] = ((a.active[b] >> k) & 1u) ? 1.0f : 0.0f;
            } else if (j == Lt::S_LOSS) {
                s_par[Lt::LP_FROZEN] = 0.0f;
            }
        }
    }
    __syncthreads();

    BlockRegs<D, C, K> R;
    R.load(s_par);
    if (a.kc.qmode != 0 || a.kc.qpis != 0)
        quantize_packed<D, C, K, QUANT>(R.P, a.kc, (QUANT && a.mus_grid != nullptr) ? a.mus_grid + (size_t)b * (K * D) : nullptr);
    R.template derive<IC>(a.kc);
    float xc[D];                                   // coordinates of the lane's first pixel (hoisted axes: of all its pixels)
#pragma unroll
    for (int l = 0; l < D; ++l) xc[l] = a.coords[l * N + min(sub, N - 1)];
    if (HL > 0) hoist_const<D, C, K, HL, IC>(R, xc);

    float acc[Lt::NSLOT];
#pragma unroll
    for (int j = 0; j < Lt::NSLOT; ++j) acc[j] = 0.0f;

    const int pxl = (N + G - 1) / G;
    const int full = N / G;                        // steps in which every lane of the block has a pixel
    unsigned long long flags[K];                   // influence votes of the full steps (scalar unit), see pixel<>
#pragma unroll
    for (int k = 0; k < K; ++k) flags[k] = 0ull;
    // OUT: the launch wants per-pixel outputs (reconstruction / gate planes / argmax); the loss-only pass of a validation
    // takes the store-free instance of the loop, two pixels per trip
    const bool any_out = (OM == 0) ? ((a.recon != nullptr) || (a.gate_w != nullptr) || (a.argmax != nullptr) || SSIM) : (OM == 2);
    // `keep`: non-null = hand the pixel's outputs (q[C], wt[K], argmax as float) back to the caller instead of storing
    // them (the grouped 16-byte stores of the register path below)
    auto step_t = [&](int n, auto voted, auto want_out, const float (&t)[C], float lw, float* keep = nullptr) {
        constexpr bool OUT = decltype(want_out)::value;
        float x[D];
#pragma unroll
        for (int l = 0; l < D; ++l) x[l] = (l < D - HL) ? s_coords[l * N + n] : 0.0f;
        PixelOut<D, C, K> o;
        pixel<D, C, K, false, HL, false, IC, decltype(voted)::value>(R, a.kc, x, t, lw, acc, o, nullptr, flags);
        if (OUT && keep != nullptr) {
#pragma unroll
            for (int c = 0; c < C; ++c) keep[c] = o.q[c];
            float best = 0.0f;
            int arg = 255;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                keep[C + k] = o.wt[k];
                if (o.wt[k] > best) { best = o.wt[k]; arg = k; }
            }
            keep[C + K] = __int_as_float(arg);
            return;
        }
        if (SSIM) {
#pragma unroll
            for (int c = 0; c < C; ++c) s_X[c * N + n] = o.q[c];
        }
        if (OUT && valid_b) {
            if (a.recon != nullptr) {
#pragma unroll
                for (int c = 0; c < C; ++c) a.recon[((size_t)b * C + c) * N + n] = o.q[c];
            }
            if (a.gate_w != nullptr) {
#pragma unroll
                for (int k = 0; k < K; ++k) a.gate_w[((size_t)b * K + k) * N + n] = o.wt[k];
            }
            if (a.argmax != nullptr) {
                // tf.argmax over the kernels with influence (smoe.py:833): first maximum;
                // a pixel with no influential kernel resolves to the first listed kernel
                // of the block, which is only known after the block reduction -> 255 for now.
                float best = 0.0f;
                int arg = 255;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (o.wt[k] > best) { best = o.wt[k]; arg = k; }
                }
                a.argmax[(size_t)b * N + n] = (uint8_t)arg;
            }
        }
    };
    auto step = [&](int n, auto voted, auto want_out) {              // targets / loss weight of pixel n from the staged planes
        float t[C];
#pragma unroll
        for (int c = 0; c < C; ++c) t[c] = s_tgt[c * N + n];
        step_t(n, voted, want_out, t, has_lw ? s_lw[n] : 1.0f);
    };
    if (regt) {
        // targets in registers: the (at most PXR) steps are unrolled; `full` is uniform, so the votes stay legal
        auto run = [&](auto want_out) {
            constexpr bool OUT = decltype(want_out)::value;
            // Outputs: four consecutive steps of a block's G lanes cover 4 G consecutive pixels of every output plane.
            // The wavefront turns them through its (idle) reduction scratch -- four dword stores per lane in pixel order,
            // one 16-byte read -- so that every lane writes 16 bytes and a block 4 G contiguous floats per instruction
            // instead of four G-float segments (all outputs at 65 536 blocks: 125 -> see DESIGN 3.2).
            // (only while the four kept steps are <= 24 registers: with three channels they cost the loss-only instance of
            // this kernel a wavefront per SIMD -- 66 -> 84 us -- for 3.9 -> 4.0 TB/s on the output path)
            constexpr bool GROUP_OK = (OM == 2) || (C + K + 1) * 4 <= 24;
            const bool grouped = GROUP_OK && OUT && ((N & 3) == 0) && ((G & 3) == 0);
            float* xs = s_scratch + grp * (4 * G);
#pragma unroll
            for (int i0 = 0; i0 < PXR; i0 += 4) {
                if (grouped && i0 + 3 < full) {
                    float keep[4][C + K + 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) step_t((i0 + r) * G + sub, std::true_type{}, want_out, tr[i0 + r], lwr[i0 + r], keep[r]);
                    const int nb = i0 * G + 4 * sub;                       // first of the lane's four consecutive pixels
                    auto turn = [&](int slot) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xs[r * G + sub] = keep[r][slot];
                        wave_lds_sync();
                        const float4 v = *reinterpret_cast<const float4*>(xs + 4 * sub);
                        wave_lds_sync();
                        return v;
                    };
                    if (a.recon != nullptr) {
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float4 v = turn(c);
                            if (valid_b) store_stream(a.recon + ((size_t)b * C + c) * N + nb, v);
                        }
                    }
                    if (a.gate_w != nullptr) {
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const float4 v = turn(C + k);
                            if (valid_b) store_stream(a.gate_w + ((size_t)b * K + k) * N + nb, v);
                        }
                    }
                    if (a.argmax != nullptr) {
                        const float4 v = turn(C + K);
                        const uint32_t w = (uint32_t)__float_as_int(v.x) | ((uint32_t)__float_as_int(v.y) << 8)
                                         | ((uint32_t)__float_as_int(v.z) << 16) | ((uint32_t)__float_as_int(v.w) << 24);
                        if (valid_b) *reinterpret_cast<uint32_t*>(a.argmax + (size_t)b * N + nb) = w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + r;
                        if (i < full) step_t(i * G + sub, std::true_type{}, want_out, tr[i], lwr[i]);
                        else if (i < pxl && i * G + sub < N) step_t(i * G + sub, std::false_type{}, want_out, tr[i], lwr[i]);
                    }
                }
            }
        };
        if (any_out) run(std::true_type{}); else run(std::false_type{});
    } else if (any_out) {
        for (int i = 0; i < full; ++i) step(i * G + sub, std::true_type{}, std::true_type{});
        if (full < pxl && full * G + sub < N) step(full * G + sub, std::false_type{}, std::true_type{});
    } else {
        int i = 0;
        for (; i + 1 < full; i += 2) {
            step(i * G + sub, std::true_type{}, std::false_type{});
            step((i + 1) * G + sub, std::true_type{}, std::false_type{});
        }
        if (i < full) step(i * G + sub, std::true_type{}, std::false_type{});
        if (full < pxl && full * G + sub < N) step(full * G + sub, std::false_type{}, std::false_type{});
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc[Lt::S_CNT + k] += ((flags[k] >> lane) & 1ull) ? 1.0f : 0.0f;

    if constexpr (SSIM) {                              // loss_pixel = 1 - SSIM (smoe.py:1006-1010)
        wave_lds_sync();
        if constexpr (G == 16) acc[Lt::S_LOSS] = ssim_block16<C, false>(s_X, const_cast<float*>(s_tgt), sub, wj, a.kc.sw);
        else if constexpr (D == 3) acc[Lt::S_LOSS] = ssim_block3<C, false>(s_X, s_tgt, s_Wa, s_Wb, s_Tr, s_Tc, s_Tt, a.kc.sw, bh, bw, bt, N, lane);
        else acc[Lt::S_LOSS] = ssim_block<C, false>(s_X, s_tgt, s_Wa, s_Wb, s_Tr, s_Tc, a.kc.sw, bh, bw, N, lane);
    }
    float total[T::SPL];
    reduce_slots<D, C, K, G, WAVES, Layout<D, C, K>::NPAR>(acc, s_scratch, lane, total);

    // publish the influence flags of the block through LDS (needed by every lane below)
#pragma unroll
    for (int s = 0; s < T::SPL; ++s) {
        const int j = T::slot_of(sub, s);
        if (j >= Lt::S_CNT && j < Lt::S_CNT + K) s_scratch[j - Lt::S_CNT + grp * 16] = (total[s] > 0.0f) ? 1.0f : 0.0f;
    }
    wave_lds_sync();
    uint32_t newmask = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) newmask |= (s_scratch[k + grp * 16] != 0.0f) ? (1u << k) : 0u;

    if (valid_b) {
#pragma unroll
        for (int s = 0; s < T::SPL; ++s) {
            const int j = T::slot_of(sub, s);
            if (j == Lt::S_LOSS) {
                float lossv = SSIM ? 1.0f + total[s] : total[s];
                if (a.reg_pi != 0.0f || a.reg_u != 0.0f) {
                    const float rp = a.kc.kcount_norm ? a.kc.pis_l1_raw / count_pis<D, C, K>(R.P) : a.reg_pi;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool act = R.act(k);
                        if (act) {
                            lossv += rp * R.pi(k);
#pragma unroll
                            for (int l = 0; l < D; ++l) lossv += a.reg_u * R.A(k, l, l);
                        }
                    }
                }
                if (a.loss != nullptr) a.loss[b] = lossv;
                if (a.update_active) a.active[b] = newmask;
            } else if (j == Lt::S_SSE) {
                if (a.sse != nullptr) a.sse[b] = total[s];
            }
        }
        if (a.argmax != nullptr) {
            // every lane patches the bytes it wrote itself: four consecutive pixels per lane where the stores were grouped
            const uint32_t first = (newmask != 0u) ? (uint32_t)(__ffs(newmask) - 1) : 0u;
            const bool grouped = ((OM == 2) || (C + K + 1) * 4 <= 24) && regt && ((N & 3) == 0) && ((G & 3) == 0);
            for (int i = 0; i < pxl; ++i) {
                if (grouped && (i | 3) < full) {
                    if ((i & 3) == 0) {
                        uint32_t* p32 = reinterpret_cast<uint32_t*>(a.argmax + (size_t)b * N + i * G + 4 * sub);
                        const uint32_t w0 = *p32;
                        uint32_t w = w0;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (((w >> (8 * u)) & 255u) == 255u) w = (w & ~(255u << (8 * u))) | (first << (8 * u));
                        if (w != w0) *p32 = w;
                    }
                    continue;
                }
                const int n = i * G + sub;
                if (n < N) {
                    uint8_t* p8 = a.argmax + (size_t)b * N + n;
                    if (*p8 == 255) *p8 = (uint8_t)first;
                }
            }
        }
    }
}

// update_kernel_list when the graph is built on fake-quantised variables: the probe test (smoe.py:806) sees
// q(A), q(musX) and pis_mask = qpis > 0.  One thread per block (mode-3 ranges need all its kernels).
template <int D, int C, int K>
__global__ void readmit_quant_kernel(ReadmitArgs a, KernelConsts kc) {
    using Lt = Layout<D, C, K>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float P[Lt::NPAR];
#pragma unroll
    for (int j = 0; j < Lt::NPAR; ++j) {
        int tensor, kern; long off;
        decode_slot<D, C, K>(j, b, tensor, off, kern);
        P[j] = pick(a.p, tensor)[off];
    }
    quantize_packed<D, C, K, true>(P, kc, (a.mus_grid != nullptr) ? a.mus_grid + (size_t)b * (K * D) : nullptr);
    int nprobe = 1;
#pragma unroll
    for (int l = 0; l < D; ++l) nprobe *= 3;
    uint32_t add = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float* p = P + k * Lt::PK;
        bool near = false;
        for (int q = 0; q < nprobe; ++q) {
            float r[D];
            int rem = q;
#pragma unroll
            for (int l = D - 1; l >= 0; --l) {
                const int sel = rem % 3;
                rem /= 3;
                r[l] = a.probes[l * 3 + sel] - p[Lt::O_MU + l];
            }
            float maha = 0.0f;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float zz = 0.0f;
                if (kc.inverse_cov) {                 // r^T A r with the symmetric A (smoe.py:791-793)
#pragma unroll
                    for (int l = 0; l < D; ++l) zz = fmaf(r[l], (l >= m) ? p[Lt::O_A + tri_index(l, m)] : p[Lt::O_A + tri_index(m, l)], zz);
                    maha = fmaf(zz, r[m], maha);
                } else {
#pragma unroll
                    for (int l = m; l < D; ++l) zz = fmaf(r[l], p[Lt::O_A + tri_index(l, m)], zz);
                    maha = fmaf(zz, zz, maha);
                }
            }
            near = near || (maha < 800.0f);
        }
        if (near && p[Lt::O_PI] > 0.0f) add |= 1u << k;
    }
    a.active[b] |= add;
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_readmit_quant(const ReadmitArgs& a, const KernelConsts& kc, hipStream_t st) {
    const int threads = 64;
    hipLaunchKernelGGL((readmit_quant_kernel<D, C, K>), dim3((a.B + threads - 1) / threads), dim3(threads), 0, st, a, kc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers + dispatch table
// ---------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize sticks to the function: set it when a launch needs more than any earlier one
// of this thread did (one process per GPU; the call costs microseconds of host time in front of every launch otherwise).
inline hipError_t allow_lds(const void* kern, size_t bytes) {
    struct Seen { const void* k; size_t b; int dev; };
    static thread_local Seen seen[64];
    static thread_local int nseen = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < nseen; ++i)
        if (seen[i].k == kern && seen[i].dev == dev) {
            if (seen[i].b >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) seen[i].b = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && nseen < 64) seen[nseen++] = Seen{kern, bytes, dev};
    return e;
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fit(const FitArgs& a, int hoist, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    auto kern = fit_kernel<D, C, K, G, WAVES, 0>;
    int hl = 0;
    if (hoist >= 1) { kern = fit_kernel<D, C, K, G, WAVES, 1>; hl = 1; }
    if (D == 3 && hoist >= 2) { kern = fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1)>; hl = 2; }
    int nb = T::NB;
    if constexpr (G == 64 && WAVES == 2) {
        if (a.pair) {                                  // one block on both wavefronts of the workgroup (see fit_kernel)
            kern = fit_kernel<D, C, K, G, WAVES, 0, false, false, false, true>;
            if (hl == 1) kern = fit_kernel<D, C, K, G, WAVES, 1, false, false, false, true>;
            if (hl == 2) kern = fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1), false, false, false, true>;
            nb = 1;
        }
    }
    const size_t shm = T::bytes(a.N, a.loss_w != nullptr, D - hl, false);
    FitArgs aa = a;
    size_t shm_all = shm;
    aa.desc_off = 0;
    if (T::wants_owner_post(a.N)) { aa.desc_off = (int)(shm / sizeof(float)); shm_all += sizeof(float) * (size_t)T::NB * T::DESC_STRIDE; }
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm_all);
    if (e != hipSuccess) return e;
    const int grid = (a.B + nb - 1) / nb;
    aa.lds_floats = (int)(shm_all / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm_all, st, aa);
    return hipGetLastError();
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fwd(const FwdArgs& a, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    const bool outs = (a.recon != nullptr) || (a.gate_w != nullptr) || (a.argmax != nullptr);
    auto kern = outs ? forward_kernel<D, C, K, G, WAVES, false, false, false, 0, 2> : forward_kernel<D, C, K, G, WAVES, false, false, false, 0, 1>;
    int hl = 0;
    if (a.hoist >= 1) {
        kern = outs ? forward_kernel<D, C, K, G, WAVES, false, false, false, 1, 2> : forward_kernel<D, C, K, G, WAVES, false, false, false, 1, 1>;
        hl = 1;
    }
    if (D == 3 && a.hoist >= 2) {
        kern = outs ? forward_kernel<D, C, K, G, WAVES, false, false, false, (D == 3 ? 2 : 1), 2>
                    : forward_kernel<D, C, K, G, WAVES, false, false, false, (D == 3 ? 2 : 1), 1>;
        hl = 2;
    }
    FwdArgs aa = a;
    aa.regt = ((a.N + G - 1) / G <= T::FWD_PXR) ? 1 : 0;
    const size_t shm = aa.regt ? sizeof(float) * (size_t)T::off_tgt(a.N, D - hl) : T::bytes(a.N, a.loss_w != nullptr, D - hl);
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
    if (e != hipSuccess) return e;
    const int grid = (a.B + T::NB - 1) / T::NB;
    aa.lds_floats = (int)(shm / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm, st, aa);
    return hipGetLastError();
}

// quantization_mode 2 / 3 launches
template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fit_quant(const FitArgs& a, int hoist, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    const bool ic = a.kc.inverse_cov != 0;
    auto kern = ic ? fit_kernel<D, C, K, G, WAVES, 0, false, true, true> : fit_kernel<D, C, K, G, WAVES, 0, false, true>;
    int hl = 0;
    if (hoist >= 1) { kern = ic ? fit_kernel<D, C, K, G, WAVES, 1, false, true, true> : fit_kernel<D, C, K, G, WAVES, 1, false, true>; hl = 1; }
    if (D == 3 && hoist >= 2) {
        kern = ic ? fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1), false, true, true> : fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1), false, true>;
        hl = 2;
    }
    const size_t shm = T::bytes(a.N, a.loss_w != nullptr, D - hl, true);
    FitArgs aa = a;
    size_t shm_all = shm;
    aa.desc_off = 0;
    if (T::wants_owner_post(a.N)) { aa.desc_off = (int)(shm / sizeof(float)); shm_all += sizeof(float) * (size_t)T::NB * T::DESC_STRIDE; }
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm_all);
    if (e != hipSuccess) return e;
    const int grid = (a.B + T::NB - 1) / T::NB;
    aa.lds_floats = (int)(shm_all / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm_all, st, aa);
    return hipGetLastError();
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fwd_quant(const FwdArgs& a, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    const bool ic = a.kc.inverse_cov != 0;
    auto kern = ic ? forward_kernel<D, C, K, G, WAVES, false, true, true> : forward_kernel<D, C, K, G, WAVES, false, true>;
    int hl = 0;
    if (a.hoist >= 1) { kern = ic ? forward_kernel<D, C, K, G, WAVES, false, true, true, 1> : forward_kernel<D, C, K, G, WAVES, false, true, false, 1>; hl = 1; }
    FwdArgs aa = a;
    aa.regt = ((a.N + G - 1) / G <= T::FWD_PXR) ? 1 : 0;
    const size_t shm = aa.regt ? sizeof(float) * (size_t)T::off_tgt(a.N, D - hl) : T::bytes(a.N, a.loss_w != nullptr, D - hl);
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
    if (e != hipSuccess) return e;
    const int grid = (a.B + T::NB - 1) / T::NB;
    aa.lds_floats = (int)(shm / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm, st, aa);
    return hipGetLastError();
}

// train_inverse_cov launches for the margin loss without mode-2/3 quantisation (the SSIM and QUANT launchers pick
// their own IC instantiations)
template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fit_ic(const FitArgs& a, int hoist, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    auto kern = fit_kernel<D, C, K, G, WAVES, 0, false, false, true>;
    int hl = 0;
    if (hoist >= 1) { kern = fit_kernel<D, C, K, G, WAVES, 1, false, false, true>; hl = 1; }
    if (D == 3 && hoist >= 2) { kern = fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1), false, false, true>; hl = 2; }
    const size_t shm = T::bytes(a.N, a.loss_w != nullptr, D - hl, false);
    FitArgs aa = a;
    size_t shm_all = shm;
    aa.desc_off = 0;
    if (T::wants_owner_post(a.N)) { aa.desc_off = (int)(shm / sizeof(float)); shm_all += sizeof(float) * (size_t)T::NB * T::DESC_STRIDE; }
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm_all);
    if (e != hipSuccess) return e;
    const int grid = (a.B + T::NB - 1) / T::NB;
    aa.lds_floats = (int)(shm_all / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm_all, st, aa);
    return hipGetLastError();
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fwd_ic(const FwdArgs& a, hipStream_t st) {
    using T = Tile<D, C, K, G, WAVES>;
    auto kern = forward_kernel<D, C, K, G, WAVES, false, false, true>;
    int hl = 0;
    if (a.hoist >= 1) { kern = forward_kernel<D, C, K, G, WAVES, false, false, true, 1>; hl = 1; }
    if (D == 3 && a.hoist >= 2) { kern = forward_kernel<D, C, K, G, WAVES, false, false, true, (D == 3 ? 2 : 1)>; hl = 2; }
    FwdArgs aa = a;
    aa.regt = ((a.N + G - 1) / G <= T::FWD_PXR) ? 1 : 0;
    const size_t shm = aa.regt ? sizeof(float) * (size_t)T::off_tgt(a.N, D - hl) : T::bytes(a.N, a.loss_w != nullptr, D - hl);
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
    if (e != hipSuccess) return e;
    const int grid = (a.B + T::NB - 1) / T::NB;
    aa.lds_floats = (int)(shm / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm, st, aa);
    return hipGetLastError();
}

// ssim_opt launches: 2-d blocks on the 16- and 64-lane tilings, 3-d blocks on the one-block-per-wavefront tiling
template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fit_ssim(const FitArgs& a, int hoist, hipStream_t st) {
    if constexpr (D == 2 || (D == 3 && G == 64)) {
        using T = Tile<D, C, K, G, WAVES>;
        if (G == 16 && (a.bh != 16 || a.bw != 16 || hoist < 1)) return hipErrorNotSupported;   // register path: 16x16 only
        const bool ic = a.kc.inverse_cov != 0;
        const bool q = a.kc.qmode >= 2;             // all variables fake-quantised: the QUANT instantiation of the SSIM kernel
        constexpr int H0 = (G == 16) ? 1 : 0;
        auto kern = q ? (ic ? fit_kernel<D, C, K, G, WAVES, H0, true, true, true> : fit_kernel<D, C, K, G, WAVES, H0, true, true>)
                      : (ic ? fit_kernel<D, C, K, G, WAVES, H0, true, false, true> : fit_kernel<D, C, K, G, WAVES, H0, true>);
        int hl = H0;
        if (hoist >= 1) {
            kern = q ? (ic ? fit_kernel<D, C, K, G, WAVES, 1, true, true, true> : fit_kernel<D, C, K, G, WAVES, 1, true, true>)
                     : (ic ? fit_kernel<D, C, K, G, WAVES, 1, true, false, true> : fit_kernel<D, C, K, G, WAVES, 1, true>);
            hl = 1;
        }
        const size_t shm = T::bytes_ssim(a.N, a.loss_w != nullptr, D - hl, a.bh, a.bw, q, (D == 3) ? a.bt : 0);
    FitArgs aa = a;
    size_t shm_all = shm;
    aa.desc_off = 0;
    if (T::wants_owner_post(a.N)) { aa.desc_off = (int)(shm / sizeof(float)); shm_all += sizeof(float) * (size_t)T::NB * T::DESC_STRIDE; }
        hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm_all);
        if (e != hipSuccess) return e;
        const int grid = (a.B + T::NB - 1) / T::NB;
        aa.lds_floats = (int)(shm_all / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm_all, st, aa);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int D, int C, int K, int G, int WAVES>
hipError_t launch_fwd_ssim(const FwdArgs& a, hipStream_t st) {
    if constexpr (D == 2 || (D == 3 && G == 64)) {
        using T = Tile<D, C, K, G, WAVES>;
        if (G == 16 && (a.bh != 16 || a.bw != 16)) return hipErrorNotSupported;
        const size_t shm = T::bytes_ssim(a.N, a.loss_w != nullptr, D, a.bh, a.bw, false, (D == 3) ? a.bt : 0);
        const bool q = a.kc.qmode >= 2;
        auto kern = q ? (a.kc.inverse_cov ? forward_kernel<D, C, K, G, WAVES, true, true, true> : forward_kernel<D, C, K, G, WAVES, true, true>)
                      : (a.kc.inverse_cov ? forward_kernel<D, C, K, G, WAVES, true, false, true> : forward_kernel<D, C, K, G, WAVES, true>);
        hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
        if (e != hipSuccess) return e;
        const int grid = (a.B + T::NB - 1) / T::NB;
        FwdArgs aa = a;
        aa.regt = 0;
        aa.lds_floats = (int)(shm / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T::THREADS), shm, st, aa);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int D, int C, int K, int G, int WAVES>
size_t lds_bytes(int N, bool has_lw, bool hq) { return Tile<D, C, K, G, WAVES>::bytes(N, has_lw, D, hq); }

template <int D, int C, int K, int G, int WAVES>
size_t lds_bytes_ssim(int N, bool has_lw, int bh, int bw, int bt, bool hq) {
    if (!(D == 2 || (D == 3 && G == 64)) || (G == 16 && (bh != 16 || bw != 16))) return (size_t)-1;
    return Tile<D, C, K, G, WAVES>::bytes_ssim(N, has_lw, D, bh, bw, hq, (D == 3) ? bt : 0);
}

template <int D, int C, int K, int G, int WAVES>
int fit_occupancy(int N, bool has_lw, int hoist, bool pair) {
    using T = Tile<D, C, K, G, WAVES>;
    int nb = 0;
    auto kern = fit_kernel<D, C, K, G, WAVES, 0>;
    int hl = 0;
    if (hoist >= 1) { kern = fit_kernel<D, C, K, G, WAVES, 1>; hl = 1; }
    if (D == 3 && hoist >= 2) { kern = fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1)>; hl = 2; }
    if constexpr (G == 64 && WAVES == 2) {
        if (pair) {
            kern = fit_kernel<D, C, K, G, WAVES, 0, false, false, false, true>;
            if (hl == 1) kern = fit_kernel<D, C, K, G, WAVES, 1, false, false, false, true>;
            if (hl == 2) kern = fit_kernel<D, C, K, G, WAVES, (D == 3 ? 2 : 1), false, false, false, true>;
        }
    }
    size_t shm = T::bytes(N, has_lw, D - hl);
    if (T::wants_owner_post(N)) shm += sizeof(float) * (size_t)T::NB * T::DESC_STRIDE;
    if (allow_lds(reinterpret_cast<const void*>(kern), shm) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, T::THREADS, shm) != hipSuccess) return -1;
    return nb * WAVES;       // resident wavefronts per CU
}

#define SMOE_STR_(x) #x
#define SMOE_STR(x) SMOE_STR_(x)
#define SMOE_VARIANT(D, C, K, G, W) \
    { D, C, K, G, W, "fit_d" SMOE_STR(D) "c" SMOE_STR(C) "k" SMOE_STR(K) "_g" SMOE_STR(G) "w" SMOE_STR(W), &launch_fit<D, C, K, G, W>, &launch_fwd<D, C, K, G, W>, &lds_bytes<D, C, K, G, W>, &fit_occupancy<D, C, K, G, W>, \
      &launch_fit_ssim<D, C, K, G, W>, &launch_fwd_ssim<D, C, K, G, W>, &lds_bytes_ssim<D, C, K, G, W>, \
      &launch_readmit_quant<D, C, K, G, W>, &launch_fit_quant<D, C, K, G, W>, &launch_fwd_quant<D, C, K, G, W>, \
      &launch_fit_ic<D, C, K, G, W>, &launch_fwd_ic<D, C, K, G, W>, \
      team_fit_ptr<D, C, K, G, W>(), team_lds_ptr<D, C, K, G, W>(), team_occ_ptr<D, C, K, G, W>(), \
      duo_fit_ptr<D, C, K, G>(), duo_lds_ptr<D, C, K, G>(), duo_occ_ptr<D, C, K, G>() }

// Reduced instantiation for the (dim, channels, kernels) triples outside the BASELINE shapes: the margin loss with and
// without train_inverse_cov (quantize_pis included: it lives in the default kernels); ssim_opt and quantization_mode
// 2 / 3 are refused for these triples (smoe_capi.hip) -- each costs a further set of kernels per triple.
#define SMOE_VARIANT_BASIC(D, C, K, G, W) \
    { D, C, K, G, W, "fit_d" SMOE_STR(D) "c" SMOE_STR(C) "k" SMOE_STR(K) "_g" SMOE_STR(G) "w" SMOE_STR(W), &launch_fit<D, C, K, G, W>, &launch_fwd<D, C, K, G, W>, &lds_bytes<D, C, K, G, W>, &fit_occupancy<D, C, K, G, W>, \
      nullptr, nullptr, nullptr, &launch_readmit_quant<D, C, K, G, W>, nullptr, nullptr, \
      &launch_fit_ic<D, C, K, G, W>, &launch_fwd_ic<D, C, K, G, W>, \
      team_fit_ptr<D, C, K, G, W>(), team_lds_ptr<D, C, K, G, W>(), team_occ_ptr<D, C, K, G, W>(), \
      duo_fit_ptr<D, C, K, G>(), duo_lds_ptr<D, C, K, G>(), duo_occ_ptr<D, C, K, G>() }

}  // namespace smoe
#endif
