// smoe_shared.hip -- shared-kernel image mode: ONE global set of K kernels, the
// image is cut into batches (smoe.py:18-35), every batch evaluates only the kernels on its
// kernel list (smoe.py:738-753), gradients of all batches of a pass are accumulated
// (smoe.py:1148-1150) and ONE Adam step follows (smoe.py:1788).
//
// One 256-thread workgroup per batch, PXL pixels per lane held in registers.  The batch's active
// kernels are compacted into an LDS list and staged (with derived quantities) through LDS in chunks;
// three sweeps over the list per pass: (A) gate normaliser S, (B) masked gate / experts / blend /
// influence flags, (C) reverse pass with per-kernel raw sums reduced across the wavefront and written
// to the batch's own rows of a partial buffer; a gather step sums the rows per kernel in batch order
// (fixed order, bit-deterministic; fp64 atomics only when the rows would not fit in memory).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smoe_device.h"
#include "smoe_ssim.hip.h"
#include "smoe_fq.hip.h"

namespace smoe {

namespace {

constexpr int SH_THREADS = 256;
constexpr int SH_KC = 64;                 // kernels staged per LDS chunk
constexpr float SQ = 0.84932180028801904272f;       // sqrt(0.5*log2(e)), see smoe_block.hip.h
constexpr float INV_SQ = 1.17740022503374817543f;

constexpr int tri(int l, int m) { return l * (l + 1) / 2 + m; }

template <int D, int C>
struct SL {                                // staged (derived) record of one kernel in LDS
    static constexpr int TRI = D * (D + 1) / 2;
    static constexpr int O_AS = 0;         // A' = SQ*A lower triangle
    static constexpr int O_CZ = TRI;       // c = A'^T mu
    static constexpr int O_COEF = TRI + D;
    static constexpr int O_NU = O_COEF + 1;
    static constexpr int O_GA = O_NU + C;  // gamma[l][c]
    static constexpr int SP = O_GA + D * C;
    // raw accumulator record (per kernel): su | suz[D] | sxz[TRI] | swg[C] | swgx[D][C]
    static constexpr int R_SU = 0;
    static constexpr int R_SUZ = 1;
    static constexpr int R_SXZ = 1 + D;
    static constexpr int R_SWG = R_SXZ + TRI;
    static constexpr int R_SWGX = R_SWG + C;
    static constexpr int PK = R_SWGX + D * C;
};

// Sum over the 64 lanes, returned to every lane.  DPP only (no LDS round trips: __shfl_xor is a ds_bpermute per step):
// an inclusive scan inside each 16-lane row (row_shr 1, 2, 4, 8; lanes without a source read 0), then lane 15 of rows
// 0 / 2 is broadcast into rows 1 / 3 and lane 31 into rows 2 + 3, which leaves the total in lane 63.  Fixed order.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_src(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, ROWS == 0xf));
}
// inclusive prefix sum of an int over the 64 lanes, same DPP steps (row_bcast adds the preceding rows' totals to whole rows)
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_srci(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, ROWS == 0xf); }
__device__ __forceinline__ int wave_scan(int v) {
    v += dpp_srci<0x111, 0xf>(v);
    v += dpp_srci<0x112, 0xf>(v);
    v += dpp_srci<0x114, 0xf>(v);
    v += dpp_srci<0x118, 0xf>(v);
    v += dpp_srci<0x142, 0xa>(v);
    v += dpp_srci<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_src<0x111, 0xf>(v);
    v += dpp_src<0x112, 0xf>(v);
    v += dpp_src<0x114, 0xf>(v);
    v += dpp_src<0x118, 0xf>(v);
    v += dpp_src<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3
    v += dpp_src<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Global accesses of the one-launch fit (shared_fit_kernel) that carry data BETWEEN workgroups inside the launch: the gfx950 L2
// caches of the eight XCDs are not coherent with each other, so these loads / stores are device-scope relaxed atomics (sc1: the
// store writes through, the load does not take a stale line) and the grid barrier needs no L2 write-back / invalidate -- 256
// workgroups each running `buffer_wbl2` per barrier cost ~45 us per barrier.  COH = false: plain accesses (every other kernel).
template <bool COH, class T>
__device__ __forceinline__ T gload(const T* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, class T>
__device__ __forceinline__ void gstore(T* p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Fixed-range fake quant of the variables inside the graph (quantize_pis; quantization_mode 2; smoe.py:474-496):
// group g: 0 A, 1 musX, 2 nu_e, 3 pis, 4 gamma_e.
__device__ __forceinline__ bool fq_on(const KernelConsts& kc, int g) { return (g == 3) ? (kc.qpis != 0) : (kc.qmode == 2); }
__device__ __forceinline__ float fqv(float x, const KernelConsts& kc, int g) {
    if (!fq_on(kc, g)) return x;
    const float cl = fminf(fmaxf(x, kc.q_nmin[g]), kc.q_nmax[g]);
    return floorf((cl - kc.q_nmin[g]) * kc.q_inv[g] + 0.5f) * kc.q_scale[g] + kc.q_nmin[g];
}
__device__ __forceinline__ bool fq_pass(float x, const KernelConsts& kc, int g) {
    return !fq_on(kc, g) || (x >= kc.q_nmin[g] && x <= kc.q_nmax[g]);
}
// Variable of mode-3 tensor t (0 A_diagonal, 1 A_corr, 2 musX, 3 nu_e, 4 gamma_e) as the graph sees it: fixed ranges
// (mode 2) by group, or the image-wide min / max range record of shared_ranges_kernel (mode 3, smoe.py:497-530)
__device__ __forceinline__ float fqt(float x, const KernelConsts& kc, const float* rng, int t) {
    if (kc.qmode == 3) {
        if (t == 2 && !kc.q_musx) return x;                      // musX is quantised only when trained (smoe.py:506)
        const float* o = rng + t * 8;
        FqRange r;
        r.nmin = o[0]; r.nmax = o[1]; r.scale = o[2]; r.inv = o[3]; r.back = o[4]; r.zero = o[5] != 0.0f;
        r.shift = (t == 0 && kc.radial) ? 0.0f : r.back;         // radial_as: unshifted input (fq_vars, smoe.py:498-504)
        return fq_val(x, r);
    }
    return fqv(x, kc, (t <= 1) ? 0 : ((t == 2) ? 1 : ((t == 3) ? 2 : 4)));
}
// Centre l of kernel k as the graph reads it.  use_diff_center with quantization_mode 2 / 3 (G = the kernel-grid centres
// [K][D], smoe.py:390-394,746-747): the quantised variable is the OFFSET musX - grid, the graph reads fq(offset) + grid.
template <int D, bool COH = false>
__device__ __forceinline__ float mu_off(const float* __restrict__ musX, const float* __restrict__ G, const KernelConsts& kc, int k, int l) {
    const float x = gload<COH>(&musX[(size_t)k * D + l]);
    return (G != nullptr && kc.qmode >= 2) ? x - G[(size_t)k * D + l] : x;
}
template <int D, bool COH = false>
__device__ __forceinline__ float mu_graph(const float* __restrict__ musX, const float* __restrict__ G, const KernelConsts& kc,
                                          const float* rng, int k, int l) {
    const float q = fqt(mu_off<D, COH>(musX, G, kc, k, l), kc, rng, 2);
    return (G != nullptr && kc.qmode >= 2) ? q + G[(size_t)k * D + l] : q;
}
// pis_l1 normaliser (smoe.py:1022-1027): start_pis, or the image-wide count of kernels with qpis > 0
__device__ __forceinline__ float reg_pi_of(float reg_pi, const KernelConsts& kc, const float* rng) {
    return kc.kcount_norm ? kc.pis_l1_raw / fmaxf(rng[40], 1.0f) : reg_pi;
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// one pass over all batches
// ---------------------------------------------------------------------------------------------
// SSIM (ssim_opt; 2-d batches, 3-d batches with the 11x11x11 window): loss_pixel = 1 - SSIM of the batch (smoe.py:980-1011); the quantised reconstruction and
// the target of the batch go to LDS planes, the whole workgroup runs the SSIM stage of smoe_ssim.hip.h and reads dL/dq
// back for the reverse sweep.
// IC: train_inverse_cov (compile-time, it sits in the per-pixel gate).
// (the body of the pass: one call per batch and workgroup -- by shared_pass_kernel once, by shared_fit_kernel once per iteration)
template <int D, int C, int PXL, bool TRAIN, bool SSIM = false, bool IC = false, bool COH = false>
__device__ __forceinline__ void shared_pass_body(const SharedArgs& a, const int b, float* __restrict__ lds) {
    using L = SL<D, C>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // b: batch index inside this launch (buffers are launch-local)
    const int Nb = a.Nb;
    const int K = a.K;

    // LDS carve-up
    int* s_list = reinterpret_cast<int*>(lds);                       // [K] compacted active kernel ids
    float* s_par = lds + a.K;                                        // [SH_KC][SP]
    float* s_acc = s_par + SH_KC * L::SP;                            // [4 waves][SH_KC][PK] (no atomics: fixed summation order)
    int* s_flag = reinterpret_cast<int*>(s_acc + 4 * SH_KC * L::PK); // [K] influence flags (by list position)
    float* s_red = reinterpret_cast<float*>(s_flag + a.K);           // [8] block reductions
    int* s_cnt = reinterpret_cast<int*>(s_red + 8);                  // [8]
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_cnt + 8);       // [KW]

    // ---- 0. compact the batch's kernel list: listed & pis > 0 (smoe.py:480,738) ----------------
    // one thread per KERNEL (256 per round): its list bit, and -- only if listed -- its prior, all loads of a round in flight
    // together (one thread per bitmap word walked its up to 32 listed kernels with one dependent global load each: the dense
    // lists of the first passes cost tens of microseconds); ascending kernel ids: position = kernels kept by the wavefronts
    // before + lanes before (ballot)
    const uint32_t* bits = a.lists + (size_t)b * a.KW;
    if (tid == 0) s_cnt[0] = 0;
    __syncthreads();
    for (int kbase = 0; kbase < K; kbase += SH_THREADS) {
        const int k = kbase + tid;
        bool keep = false;
        // (the list word and the prior are loaded together: the prior behind the bit test was a second memory round trip)
        const int kc_ = (k < K) ? k : K - 1;
        const uint32_t word = gload<COH>(&bits[kc_ >> 5]);
        const float prior = gload<COH>(&a.p.pis[kc_]);
        if (k < K && ((word >> (k & 31)) & 1u)) keep = fqv(prior, a.kc, 3) > 0.0f;
        const unsigned long long m = __ballot(keep);
        if (TRAIN && a.trained != nullptr && lane == 0) {                   // the kernels this pass trains here (gather step)
            const int w0 = (kbase + wave * 64) >> 5;
            if (w0 < a.KW) gstore<COH>(&a.trained[(size_t)(a.b0 + b) * a.KW + w0], (uint32_t)m);
            if (w0 + 1 < a.KW) gstore<COH>(&a.trained[(size_t)(a.b0 + b) * a.KW + w0 + 1], (uint32_t)(m >> 32));
        }
        if (lane == 0) s_cnt[1 + wave] = __popcll(m);
        __syncthreads();
        int off = s_cnt[0];
        for (int ww = 0; ww < wave; ++ww) off += s_cnt[1 + ww];
        if (keep) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = k;
        __syncthreads();
        if (tid == 0) s_cnt[0] += s_cnt[1] + s_cnt[2] + s_cnt[3] + s_cnt[4];
        __syncthreads();
    }
    const int Kact = s_cnt[0];
    for (int i = tid; i < Kact; i += SH_THREADS) s_flag[i] = 0;
    if (TRAIN && a.batch_epoch != nullptr && tid == 0) gstore<COH>(&a.batch_epoch[a.b0 + b], a.epoch);

    // ---- 1. this lane's pixels: global coordinates (smoe.py:2412) and targets --------------------
    int bo[D];                       // batch origin per axis (sliding_window order: last axis fastest)
    {
        int rem = a.b0 + b;           // global batch index -> position in the image
#pragma unroll
        for (int l = D - 1; l >= 0; --l) {
            bo[l] = (rem % a.grid[l]) * a.batch_shape[l];
            rem /= a.grid[l];
        }
    }
    float x[PXL][D], t[PXL][C];
    bool pv[PXL];
#pragma unroll
    for (int p = 0; p < PXL; ++p) {
        const int n = p * SH_THREADS + tid;
        pv[p] = n < Nb;
        int rem = pv[p] ? n : 0;
#pragma unroll
        for (int l = D - 1; l >= 0; --l) {
            const int idx = rem % a.batch_shape[l];
            rem /= a.batch_shape[l];
            x[p][l] = a.axis_coords[a.axis_off[l] + bo[l] + idx];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) t[p][c] = a.target[((size_t)b * C + c) * Nb + (pv[p] ? n : 0)];
    }

    // stage chunk [c0, c0+n) of the active list into LDS (derived quantities, smoe.py:732-733,809-819);
    // a list that fits one chunk (the common case after pruning) is staged once for all three sweeps
    constexpr bool ic = IC;
    int staged_c0 = -1;
    auto stage = [&](int c0, int n) {
        if (c0 == staged_c0) return;
        staged_c0 = c0;
        __syncthreads();
        if (tid < n) {
            const int k = s_list[c0 + tid];
            float* r = s_par + tid * L::SP;
            float det = 1.0f;
            float A[D][D];
#pragma unroll
            for (int l = 0; l < D; ++l)
#pragma unroll
                for (int m = 0; m <= l; ++m) {
                    A[l][m] = fqt((l == m) ? gload<COH>(&a.p.A_diagonal[((size_t)k * D + l) * D + m]) : gload<COH>(&a.p.A_corr[((size_t)k * D + l) * D + m]), a.kc, a.qrng, (l == m) ? 0 : 1);
                    if (l == m) det *= A[l][m];
                    // train_inverse_cov: the coefficients c_lm of r^T A' r over l >= m, A' = SQ^2 A (smoe.py:734-735,791-793)
                    r[L::O_AS + tri(l, m)] = ic ? ((l == m) ? SQ * SQ : 2.0f * SQ * SQ) * A[l][m] : SQ * A[l][m];
                }
#pragma unroll
            for (int m = 0; m < D; ++m) {
                float cz = 0.0f;
                if (ic) {
                    cz = mu_graph<D, COH>(a.p.musX, a.mus_grid, a.kc, a.qrng, k, m);   // the centre itself: r = x - mu per pixel
                } else {
#pragma unroll
                    for (int l = m; l < D; ++l) cz = fmaf(mu_graph<D, COH>(a.p.musX, a.mus_grid, a.kc, a.qrng, k, l), SQ * A[l][m], cz);
                }
                r[L::O_CZ + m] = cz;
            }
            const float nq = a.kc.use_det ? det / a.kc.n_dis : 1.0f;
            r[L::O_COEF] = nq * fqv(gload<COH>(&a.p.pis[k]), a.kc, 3);
#pragma unroll
            for (int c = 0; c < C; ++c) r[L::O_NU + c] = fqt(gload<COH>(&a.p.nu_e[(size_t)k * C + c]), a.kc, a.qrng, 3);
#pragma unroll
            for (int i = 0; i < D * C; ++i)
                r[L::O_GA + i] = (a.kc.train_gammas && !(a.kc.only_y_gamma && (i % C) != 0)) ? fqt(gload<COH>(&a.p.gamma_e[(size_t)k * D * C + i]), a.kc, a.qrng, 4) : 0.0f;
        }
        __syncthreads();
    };

    // g_k(x) * 1 and z' for one pixel
    auto gate = [&](const float* r, const float (&xx)[D], float (&z)[D]) -> float {
        float maha = 0.0f;
        if (ic) {                                  // z := r = x - mu ; maha' = sum_{l>=m} c_lm r_l r_m
#pragma unroll
            for (int l = 0; l < D; ++l) {
                z[l] = xx[l] - r[L::O_CZ + l];
                float tq = 0.0f;
#pragma unroll
                for (int m = 0; m <= l; ++m) tq = fmaf(r[L::O_AS + tri(l, m)], z[m], tq);
                maha = fmaf(tq, z[l], maha);
            }
            return r[L::O_COEF] * fexp2(-maha);
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float zz = -r[L::O_CZ + m];
#pragma unroll
            for (int l = D - 1; l >= m; --l) zz = fmaf(xx[l], r[L::O_AS + tri(l, m)], zz);
            z[m] = zz;
            maha = (m == 0) ? zz * zz : fmaf(zz, zz, maha);
        }
        return r[L::O_COEF] * fexp2(-maha);
    };

    // ---- 2. sweep A: gate normaliser (smoe.py:819-821) ---------------------------------------------
    float S[PXL];
#pragma unroll
    for (int p = 0; p < PXL; ++p) S[p] = 0.0f;
    for (int c0 = 0; c0 < Kact; c0 += SH_KC) {
        const int n = min(SH_KC, Kact - c0);
        stage(c0, n);
        for (int kk = 0; kk < n; ++kk) {
            const float* r = s_par + kk * L::SP;
#pragma unroll
            for (int p = 0; p < PXL; ++p) {
                float z[D];
                S[p] += gate(r, x[p], z);
            }
        }
    }
    float inv[PXL];
#pragma unroll
    for (int p = 0; p < PXL; ++p) inv[p] = frcp(fmaxf(S[p], 10e-12f));

    // ---- 3. sweep B: masked gate, experts, blend, influence, argmax (smoe.py:823-848) ---------------
    float y[PXL][C];
    float best[PXL];
    int arg[PXL];
#pragma unroll
    for (int p = 0; p < PXL; ++p) {
        best[p] = 0.0f;
        arg[p] = -1;
#pragma unroll
        for (int c = 0; c < C; ++c) y[p][c] = 0.0f;
    }
    for (int c0 = 0; c0 < Kact; c0 += SH_KC) {
        const int n = min(SH_KC, Kact - c0);
        stage(c0, n);
        for (int kk = 0; kk < n; ++kk) {
            const float* r = s_par + kk * L::SP;
            bool any = false;
#pragma unroll
            for (int p = 0; p < PXL; ++p) {
                float z[D];
                const float w = gate(r, x[p], z) * inv[p];
                const float wt = (pv[p] && w > a.kc.tau) ? w : 0.0f;
                any = any || (wt > 0.0f);
                if (wt > best[p]) { best[p] = wt; arg[p] = s_list[c0 + kk]; }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float ee = r[L::O_NU + c];
#pragma unroll
                    for (int l = 0; l < D; ++l) ee = fmaf(r[L::O_GA + l * C + c], x[p][l], ee);
                    y[p][c] = fmaf(wt, ee, y[p][c]);
                }
            }
            if (__ballot(any) != 0ull && lane == 0) s_flag[c0 + kk] = 1;       // smoe.py:829
        }
    }

    // ---- 3b. halo (overlap_of_batches > 0): the window's extra pixels take part in the influence test
    // only (smoe.py:829 runs on the whole window, the loss is cropped, smoe.py:909-923); a window pixel
    // outside the image has all-zero coordinates (np.pad of the joint domain, smoe.py:21,28)
    // maha' (in exp2 units) and prod(diag A) of kernel k at a halo pixel, straight from global memory
    auto halo_maha = [&](int k, const float (&xh)[D], float& maha, float& det) {
        float Aq[D][D], rr[D];
#pragma unroll
        for (int l = 0; l < D; ++l) {
            rr[l] = xh[l] - mu_graph<D, COH>(a.p.musX, a.mus_grid, a.kc, a.qrng, k, l);
#pragma unroll
            for (int m = 0; m <= l; ++m) {
                Aq[l][m] = fqt((l == m) ? gload<COH>(&a.p.A_diagonal[((size_t)k * D + l) * D + m]) : gload<COH>(&a.p.A_corr[((size_t)k * D + l) * D + m]), a.kc, a.qrng, (l == m) ? 0 : 1);
                if (l == m) det *= Aq[l][m];
            }
        }
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float zz = 0.0f;
            if (ic) {
#pragma unroll
                for (int l = 0; l < D; ++l) zz = fmaf(rr[l], (SQ * SQ) * ((l >= m) ? Aq[l][m] : Aq[m][l]), zz);
                maha = fmaf(zz, rr[m], maha);
            } else {
#pragma unroll
                for (int l = m; l < D; ++l) zz = fmaf(rr[l], SQ * Aq[l][m], zz);
                maha = fmaf(zz, zz, maha);
            }
        }
    };
    if (a.overlap > 0) {
        int ext[D], next = 1;
#pragma unroll
        for (int l = 0; l < D; ++l) { ext[l] = a.batch_shape[l] + 2 * a.overlap; next *= ext[l]; }
        for (int e = tid; e < next; e += SH_THREADS) {
            int rem = e;
            bool interior = true, inside_img = true;
            float xh[D];
            int gi[D];
#pragma unroll
            for (int l = D - 1; l >= 0; --l) {
                const int el = rem % ext[l];
                rem /= ext[l];
                interior = interior && (el >= a.overlap) && (el < a.overlap + a.batch_shape[l]);
                gi[l] = bo[l] + el - a.overlap;
                inside_img = inside_img && (gi[l] >= 0) && (gi[l] < a.image_shape[l]);
            }
            if (interior) continue;
#pragma unroll
            for (int l = 0; l < D; ++l) xh[l] = inside_img ? a.axis_coords[a.axis_off[l] + gi[l]] : 0.0f;
            // normaliser and influence for this pixel, reading the kernel records straight from global
            // memory (halo pixels are few; the LDS chunk may hold another part of the list)
            float Sh = 0.0f;
            for (int i = 0; i < Kact; ++i) {
                const int k = s_list[i];
                float maha = 0.0f, det = 1.0f;
                halo_maha(k, xh, maha, det);
                const float nq = a.kc.use_det ? det / a.kc.n_dis : 1.0f;
                Sh += nq * fqv(gload<COH>(&a.p.pis[k]), a.kc, 3) * fexp2(-maha);
            }
            const float invh = frcp(fmaxf(Sh, 10e-12f));
            for (int i = 0; i < Kact; ++i) {
                const int k = s_list[i];
                float maha = 0.0f, det = 1.0f;
                halo_maha(k, xh, maha, det);
                const float nq = a.kc.use_det ? det / a.kc.n_dis : 1.0f;
                if (nq * fqv(gload<COH>(&a.p.pis[k]), a.kc, 3) * fexp2(-maha) * invh > a.kc.tau) s_flag[i] = 1;
            }
        }
        __syncthreads();
    }

    // ---- 4. clip + fake quant, loss, dL/dy (smoe.py:857,899,905-937) ----------------------------------
    float Gc[PXL][C], dot[PXL];
    float loss_part = 0.0f, sse_part = 0.0f;
    if constexpr (SSIM) {
        float* s_ss = lds + a.ssim_off;
        const int bh = a.batch_shape[0], bw = a.batch_shape[1], bt = (D == 3) ? a.batch_shape[2] : 0;
        const float* s_Tr = s_ss;
        const float* s_Tc = s_ss + 11 * bh;
        const float* s_Tt = s_Tc + 11 * bw;               // 3-d batches: the taps of the third axis
        float* s_X = s_ss + ((11 * (bh + bw + bt) + 3) & ~3);
        float* s_Y = s_X + C * Nb;
        float* s_Wa = s_Y + C * Nb;
        float* s_Wb = s_Wa + 5 * Nb;                      // 2-d: three planes, 3-d: five
        for (int i = tid; i < 11 * (bh + bw + bt); i += SH_THREADS) s_ss[i] = a.ssim_T[i];
        bool ste[PXL][C];
#pragma unroll
        for (int p = 0; p < PXL; ++p) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float yc = __builtin_amdgcn_fmed3f(y[p][c], 0.0f, a.kc.nudged_max);
                const float q = floorf(fmaf(yc, a.kc.inv_scale, 0.5f)) * a.kc.scale;
                const float diff = q - t[p][c];
                ste[p][c] = pv[p] && (yc == y[p][c]);
                if (pv[p]) {
                    sse_part = fmaf(diff, diff, sse_part);
                    if (a.recon != nullptr) a.recon[((size_t)b * C + c) * Nb + p * SH_THREADS + tid] = q;
                    s_X[c * Nb + p * SH_THREADS + tid] = q;
                    s_Y[c * Nb + p * SH_THREADS + tid] = t[p][c];
                }
            }
        }
        __syncthreads();
        if constexpr (D == 3) loss_part = ssim_block3<C, TRAIN, SH_THREADS>(s_X, s_Y, s_Wa, s_Wb, s_Tr, s_Tc, s_Tt, a.kc.sw, bh, bw, bt, Nb, tid);
        else loss_part = ssim_block<C, TRAIN, SH_THREADS>(s_X, s_Y, s_Wa, s_Wb, s_Tr, s_Tc, a.kc.sw, bh, bw, Nb, tid);
#pragma unroll
        for (int p = 0; p < PXL; ++p) {
            dot[p] = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                Gc[p][c] = (TRAIN && ste[p][c]) ? s_X[c * Nb + p * SH_THREADS + tid] : 0.0f;
                dot[p] = fmaf(Gc[p][c], y[p][c], dot[p]);
            }
            dot[p] = (S[p] > 10e-12f) ? dot[p] : 0.0f;
        }
    } else {
#pragma unroll
    for (int p = 0; p < PXL; ++p) {
        dot[p] = 0.0f;
        // per-pixel loss weight (the graph's loss_weights placeholder, smoe.py:550,932,1674-1677); global batch index
        const float lw = (a.loss_w != nullptr && pv[p]) ? a.loss_w[(size_t)(a.b0 + b) * Nb + p * SH_THREADS + tid] : 1.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float yc = __builtin_amdgcn_fmed3f(y[p][c], 0.0f, a.kc.nudged_max);
            const float q = floorf(fmaf(yc, a.kc.inv_scale, 0.5f)) * a.kc.scale;
            const float diff = q - t[p][c];
            const float ad = fabsf(diff) - a.kc.epsm;
            const float cwl = a.kc.cw[c] * lw;
            if (pv[p]) {
                sse_part = fmaf(diff, diff, sse_part);
                loss_part = fmaf(cwl, ad * ad, loss_part);
                if (a.recon != nullptr) a.recon[((size_t)b * C + c) * Nb + p * SH_THREADS + tid] = q;
            }
            const float sg = __builtin_amdgcn_fmed3f(diff * 1.2676506e30f, -1.0f, 1.0f);
            const float gm = (cwl + cwl) * (ad * sg);
            Gc[p][c] = (pv[p] && yc == y[p][c]) ? gm : 0.0f;
            dot[p] = fmaf(Gc[p][c], y[p][c], dot[p]);
        }
        dot[p] = (S[p] > 10e-12f) ? dot[p] : 0.0f;
    }
    }

    // ---- 5. sweep C: reverse pass, raw sums per kernel (smoe.py:1148-1150) ---------
    if (TRAIN) {
        for (int c0 = 0; c0 < Kact; c0 += SH_KC) {
            const int n = min(SH_KC, Kact - c0);
            stage(c0, n);
            for (int kk = 0; kk < n; ++kk) {
                const float* r = s_par + kk * L::SP;
                float acc[L::PK];
#pragma unroll
                for (int j = 0; j < L::PK; ++j) acc[j] = 0.0f;
#pragma unroll
                for (int p = 0; p < PXL; ++p) {
                    float z[D];
                    const float w = gate(r, x[p], z) * inv[p];
                    const float wt = (pv[p] && w > a.kc.tau) ? w : 0.0f;
                    float eg = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        float ee = r[L::O_NU + c];
#pragma unroll
                        for (int l = 0; l < D; ++l) ee = fmaf(r[L::O_GA + l * C + c], x[p][l], ee);
                        eg = fmaf(ee, Gc[p][c], eg);
                        const float wg = wt * Gc[p][c];
                        acc[L::R_SWG + c] += wg;
#pragma unroll
                        for (int l = 0; l < D; ++l) acc[L::R_SWGX + l * C + c] = fmaf(wg, x[p][l], acc[L::R_SWGX + l * C + c]);
                    }
                    const float u = pv[p] ? fmaf(wt, eg, -(w * dot[p])) : 0.0f;
                    acc[L::R_SU] += u;
                    if (ic) {                              // raw sums of u r_l and u r_l r_m
#pragma unroll
                        for (int l = 0; l < D; ++l) {
                            const float ur = u * z[l];
                            acc[L::R_SUZ + l] += ur;
#pragma unroll
                            for (int m = 0; m <= l; ++m) acc[L::R_SXZ + tri(l, m)] = fmaf(ur, z[m], acc[L::R_SXZ + tri(l, m)]);
                        }
                    } else {
#pragma unroll
                    for (int m = 0; m < D; ++m) {
                        const float uz = u * z[m];
                        acc[L::R_SUZ + m] += uz;
#pragma unroll
                        for (int l = m; l < D; ++l) acc[L::R_SXZ + tri(l, m)] = fmaf(x[p][l], uz, acc[L::R_SXZ + tri(l, m)]);
                    }
                    }
                }
#pragma unroll
                for (int j = 0; j < L::PK; ++j) {
                    const float v = wave_sum(acc[j]);
                    if (lane == 0) s_acc[(wave * SH_KC + kk) * L::PK + j] = v;
                }
            }
            __syncthreads();
            for (int i = tid; i < n * L::PK; i += SH_THREADS) {
                const int kk = i / L::PK;
                const int j = i - kk * L::PK;
                const float v = (s_acc[i] + s_acc[SH_KC * L::PK + i]) + (s_acc[2 * SH_KC * L::PK + i] + s_acc[3 * SH_KC * L::PK + i]);
                // the batch's own row of the partial buffer (summed per kernel in batch order by the gather step: fixed order,
                // bit-deterministic); without the buffer: fp64 atomics, whose order varies from run to run
                if (a.part != nullptr) gstore<COH>(&a.part[((size_t)(a.b0 + b) * K + s_list[c0 + kk]) * L::PK + j], v);
                else atomicAdd(&a.racc[(size_t)s_list[c0 + kk] * L::PK + j], (double)v);
            }
            if (a.part == nullptr && a.nact != nullptr)
                for (int i = tid; i < n; i += SH_THREADS) atomicAdd(&a.nact[s_list[c0 + i]], 1.0);
        }
    }

    // ---- 6. per-batch scalars, kernel-list prune, argmax fix-up ------------------------------------
    loss_part = wave_sum(loss_part);
    sse_part = wave_sum(sse_part);
    __syncthreads();
    if (lane == 0) { s_red[wave] = loss_part; s_red[4 + wave] = sse_part; }
    for (int i = tid; i < a.KW; i += SH_THREADS) s_bits[i] = 0u;
    __syncthreads();
    for (int i = tid; i < Kact; i += SH_THREADS)
        if (s_flag[i]) atomicOr(&s_bits[s_list[i] >> 5], 1u << (s_list[i] & 31));
    __syncthreads();
    if (tid == 0) {
        float lossv = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (SSIM) lossv += 1.0f;                                      // smoe.py:1010: 1 - ssim
        if (a.reg_pi != 0.0f || a.reg_u != 0.0f) {                    // smoe.py:1027,1044 over the batch's active kernels
            const float rp = reg_pi_of(a.reg_pi, a.kc, a.qrng);
            for (int i = 0; i < Kact; ++i) {
                const int k = s_list[i];
                lossv += rp * fqv(gload<COH>(&a.p.pis[k]), a.kc, 3);
                for (int l = 0; l < D; ++l) lossv += a.reg_u * fqt(gload<COH>(&a.p.A_diagonal[((size_t)k * D + l) * D + l]), a.kc, a.qrng, 0);
            }
        }
        if (a.loss != nullptr) a.loss[b] = lossv;
        if (a.sse != nullptr) a.sse[b] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
    if (a.argmax != nullptr) {
        int first = 0;                                                // smoe.py:1713-1716
        for (int w = 0; w < a.KW; ++w)
            if (s_bits[w]) { first = w * 32 + (__ffs(s_bits[w]) - 1); break; }
#pragma unroll
        for (int p = 0; p < PXL; ++p)
            if (pv[p]) a.argmax[(size_t)b * Nb + p * SH_THREADS + tid] = (arg[p] >= 0) ? arg[p] : first;
    }
    if (a.update_lists)                                               // smoe.py:1763-1766
        for (int i = tid; i < a.KW; i += SH_THREADS) gstore<COH>(&a.lists[(size_t)b * a.KW + i], s_bits[i]);
}

template <int D, int C, int PXL, bool TRAIN, bool SSIM = false, bool IC = false>
__global__ void __launch_bounds__(SH_THREADS) shared_pass_kernel(SharedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    shared_pass_body<D, C, PXL, TRAIN, SSIM, IC>(a, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------
// one Adam step on the accumulated gradients (train_op, smoe.py:1788,1173-1193); clears the
// accumulators for the next pass (zero_op, smoe.py:1613)
// ---------------------------------------------------------------------------------------------
// Everything one kernel contributes to the step: its variables raw and as the graph sees them, and the gradients
// w.r.t. the (fake-quantised) graph variables incl. the l1 terms, before the backward of the fake-quant ops.
template <int D, int C>
struct KernelStep {
    float pi_raw, pi, mu_raw[D], mu[D], Araw[D][D], A[D][D], nu_raw[C], ga_raw[D * C];
    float g_pi, g_mu[D], g_A[D][D], g_nu[C], g_ga[D * C];
};

// Sum of the batches' rows of kernel k over the current pass, by the 64 lanes of a wavefront: lane l takes the batches
// l, l + 64, ... in ascending order (fp64), then a butterfly over the lanes -- a fixed order.  Every lane returns the totals.
template <int PK, bool COH = false>
__device__ __forceinline__ void gather_kernel_sums(const SharedGatherArgs& g, int k, int lane, double (&s)[PK], double& cnt) {
#pragma unroll
    for (int j = 0; j < PK; ++j) s[j] = 0.0;
    cnt = 0.0;
    const int w = k >> 5;
    const uint32_t bit = 1u << (k & 31);
    // four batches per trip, every load unconditional (a row that was not written this pass is valid memory whose content
    // is dropped by the select): the loads of a trip are independent, one memory latency instead of three per batch
    constexpr int UB = 4;
    for (int b0 = lane; b0 < g.NB_total; b0 += 64 * UB) {
        uint32_t ep[UB], tw[UB];
        float row[UB][PK];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = min(b0 + 64 * u, g.NB_total - 1);
            ep[u] = gload<COH>(&g.batch_epoch[b]);
            tw[u] = gload<COH>(&g.trained[(size_t)b * g.KW + w]);
            const float* r = g.part + ((size_t)b * g.K + k) * PK;
#pragma unroll
            for (int j = 0; j < PK; ++j) row[u][j] = gload<COH>(&r[j]);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool on = (b0 + 64 * u < g.NB_total) && (ep[u] == g.epoch) && ((tw[u] & bit) != 0u);
#pragma unroll
            for (int j = 0; j < PK; ++j) s[j] += on ? (double)row[u][j] : 0.0;
            cnt += on ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int j = 0; j < PK; ++j) s[j] += __shfl_xor(s[j], off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
}

template <int PK>
__global__ void __launch_bounds__(64) shared_gather_kernel(SharedGatherArgs g) {
    const int k = blockIdx.x, lane = threadIdx.x;
    double s[PK], cnt;
    gather_kernel_sums<PK>(g, k, lane, s, cnt);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < PK; ++j) g.racc[(size_t)k * PK + j] = s[j];
        g.nact[k] = cnt;
    }
}

template <int D, int C, bool COH = false>
__device__ __forceinline__ void kernel_step_from(const SharedAdamArgs& a, int k, const float (&r)[SL<D, C>::PK], float nact, KernelStep<D, C>& S);

template <int D, int C>
__device__ __forceinline__ void kernel_step(const SharedAdamArgs& a, int k, bool clear, KernelStep<D, C>& S) {
    using L = SL<D, C>;
    double* rk = a.racc + (size_t)k * L::PK;
    float r[L::PK];
#pragma unroll
    for (int j = 0; j < L::PK; ++j) { r[j] = (float)rk[j]; if (clear) rk[j] = 0.0; }
    const float nact = a.nact ? (float)a.nact[k] : 0.0f;
    if (a.nact && clear) a.nact[k] = 0.0;
    kernel_step_from<D, C, false>(a, k, r, nact, S);
}

template <int D, int C, bool COH>
__device__ __forceinline__ void kernel_step_from(const SharedAdamArgs& a, int k, const float (&r)[SL<D, C>::PK], float nact, KernelStep<D, C>& S) {
    using L = SL<D, C>;
    S.pi_raw = gload<COH>(&a.p.pis[k]);
    S.pi = fqv(S.pi_raw, a.kc, 3);
#pragma unroll
    for (int l = 0; l < D; ++l) {
        S.mu_raw[l] = mu_off<D, COH>(a.p.musX, a.mus_grid, a.kc, k, l);        // the quantised variable (use_diff_center: the offset)
        S.mu[l] = mu_graph<D, COH>(a.p.musX, a.mus_grid, a.kc, a.qrng, k, l);
#pragma unroll
        for (int m = 0; m < D; ++m) {
            S.Araw[l][m] = (l == m) ? gload<COH>(&a.p.A_diagonal[((size_t)k * D + l) * D + m]) : ((l > m) ? gload<COH>(&a.p.A_corr[((size_t)k * D + l) * D + m]) : 0.0f);
            S.A[l][m] = (l >= m) ? fqt(S.Araw[l][m], a.kc, a.qrng, (l == m) ? 0 : 1) : 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) S.nu_raw[c] = gload<COH>(&a.p.nu_e[(size_t)k * C + c]);
#pragma unroll
    for (int i = 0; i < D * C; ++i) S.ga_raw[i] = gload<COH>(&a.p.gamma_e[(size_t)k * D * C + i]);
    const float su = r[L::R_SU];
    const bool ic = a.kc.inverse_cov != 0;        // train_inverse_cov: suz holds sum u r_l, sxz holds sum u r_l r_m
    float suz[D];
#pragma unroll
    for (int m = 0; m < D; ++m) suz[m] = ic ? r[L::R_SUZ + m] : r[L::R_SUZ + m] * INV_SQ;
    S.g_pi = (S.pi > 0.0f ? su / S.pi : 0.0f) + nact * reg_pi_of(a.reg_pi, a.kc, a.qrng);
#pragma unroll
    for (int l = 0; l < D; ++l) {
        float g = 0.0f;
        if (ic) {                                 // d/dmu_l = sum_m A_lm sur_m with the symmetric A
#pragma unroll
            for (int m = 0; m < D; ++m) g = fmaf((l >= m) ? S.A[l][m] : S.A[m][l], suz[m], g);
        } else {
#pragma unroll
            for (int m = 0; m <= l; ++m) g = fmaf(S.A[l][m], suz[m], g);
        }
        S.g_mu[l] = g;
    }
#pragma unroll
    for (int l = 0; l < D; ++l)
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float g = 0.0f;
            if (m <= l) {
                g = ic ? ((l == m) ? -0.5f * r[L::R_SXZ + tri(l, m)] : -r[L::R_SXZ + tri(l, m)])
                       : fmaf(S.mu[l], suz[m], -(r[L::R_SXZ + tri(l, m)] * INV_SQ));
                if (l == m) {
                    if (a.use_det) g += su / S.A[l][l];
                    g += nact * a.reg_u;
                }
            }
            S.g_A[l][m] = g;
        }
#pragma unroll
    for (int c = 0; c < C; ++c) S.g_nu[c] = r[L::R_SWG + c];
#pragma unroll
    for (int i = 0; i < D * C; ++i)              // untrained / masked slopes are constants of the graph (smoe.py:1112-1117)
        S.g_ga[i] = (a.train_gammas && !(a.only_y_gamma && (i % C) != 0)) ? r[L::R_SWGX + i] : 0.0f;
}

// TF1 ApplyAdam on every variable of kernel k with the routed gradients in S (optimizer groups smoe.py:1102-1104)
template <int D, int C, bool COH = false>
__device__ __forceinline__ void kernel_apply(const SharedAdamArgs& a, int k, KernelStep<D, C>& S) {
    // One table of the kernel's variables, then ALL loads (variable, m, v), then the updates, then the stores: the loads are
    // independent and issued together -- as one adam_apply after the other every variable paid its own memory round trip
    // (device-scope loads of the one-launch fit: ~2 us each, 9 - 25 of them in a row).
    constexpr int TRI = D * (D + 1) / 2;
    constexpr int NV = 1 + D + TRI + C + D * C;
    float* var[NV]; float* pm[NV]; float* pv[NV];
    float g[NV], lr[NV];
    int n = 0;
    var[n] = &a.p.pis[k]; pm[n] = &a.m.pis[k]; pv[n] = &a.v.pis[k]; g[n] = S.g_pi; lr[n] = a.train_pis ? a.lr_pis : 0.0f; ++n;
#pragma unroll
    for (int l = 0; l < D; ++l) {
        const size_t o = (size_t)k * D + l;
        var[n] = &a.p.musX[o]; pm[n] = &a.m.musX[o]; pv[n] = &a.v.musX[o]; g[n] = S.g_mu[l]; lr[n] = a.train_musx ? a.lr_expert : 0.0f; ++n;
    }
    if (a.kc.radial) {        // radial_as (smoe.py:714-719): one value per kernel -> its gradient is the trace; A_corr untrained
        float tr = 0.0f;
#pragma unroll
        for (int l = 0; l < D; ++l) tr += S.g_A[l][l];
#pragma unroll
        for (int l = 0; l < D; ++l) S.g_A[l][l] = tr;
    }
#pragma unroll
    for (int l = 0; l < D; ++l)
#pragma unroll
        for (int m = 0; m <= l; ++m) {
            const size_t o = ((size_t)k * D + l) * D + m;
            if (l == m) { var[n] = &a.p.A_diagonal[o]; pm[n] = &a.m.A_diagonal[o]; pv[n] = &a.v.A_diagonal[o]; lr[n] = a.lr_steer; }
            else { var[n] = &a.p.A_corr[o]; pm[n] = &a.m.A_corr[o]; pv[n] = &a.v.A_corr[o]; lr[n] = a.kc.radial ? 0.0f : a.lr_steer; }
            g[n] = S.g_A[l][m]; ++n;
        }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t o = (size_t)k * C + c;
        var[n] = &a.p.nu_e[o]; pm[n] = &a.m.nu_e[o]; pv[n] = &a.v.nu_e[o]; g[n] = S.g_nu[c]; lr[n] = a.lr_expert; ++n;
    }
#pragma unroll
    for (int i = 0; i < D * C; ++i) {
        const size_t o = (size_t)k * D * C + i;
        var[n] = &a.p.gamma_e[o]; pm[n] = &a.m.gamma_e[o]; pv[n] = &a.v.gamma_e[o]; g[n] = S.g_ga[i];
        lr[n] = (a.train_gammas && !(a.only_y_gamma && (i % C) != 0)) ? a.lr_expert : 0.0f; ++n;
    }
    float x0[NV], m0[NV], v0[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { x0[i] = gload<COH>(var[i]); m0[i] = gload<COH>(pm[i]); v0[i] = gload<COH>(pv[i]); }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lr[i] == 0.0f) continue;                       // untrained: variable and slots stay (adam_apply)
        float gi = g[i];
        if (a.clip > 0.0f) gi = fminf(fmaxf(gi, -a.clip), a.clip);
        const float alpha = lr[i] * sqrtf(1.0f - a.b2p) / (1.0f - a.b1p);
        const float m2 = m0[i] + (gi - m0[i]) * (1.0f - a.beta1);
        const float v2 = v0[i] + (gi * gi - v0[i]) * (1.0f - a.beta2);
        gstore<COH>(pm[i], m2);
        gstore<COH>(pv[i], v2);
        gstore<COH>(var[i], x0[i] - (m2 * alpha) / (sqrtf(v2) + a.eps));
    }
}

// quantization_mode 0 / 1 / 2 and quantize_pis: the fake-quant backward is a per-element mask (straight through inside
// the nudged range), kernels are independent
// One wavefront per kernel.  a.gather.part != null: the wavefront first sums the batches' rows of its kernel (fixed order);
// lane 0 then takes the step.
template <int D, int C, bool COH = false>
__device__ __forceinline__ void shared_adam_body(const SharedAdamArgs& a, const int k, const int lane) {
    using L = SL<D, C>;
    KernelStep<D, C> S;
    if (a.gather.part != nullptr) {
        double s[L::PK], cnt;
        gather_kernel_sums<L::PK, COH>(a.gather, k, lane, s, cnt);
        if (lane != 0) return;
        float r[L::PK];
#pragma unroll
        for (int j = 0; j < L::PK; ++j) { r[j] = (float)s[j]; a.racc[(size_t)k * L::PK + j] = 0.0; }
        if (a.nact) a.nact[k] = 0.0;
        kernel_step_from<D, C, COH>(a, k, r, (float)cnt, S);
    } else {
        if (lane != 0) return;
        kernel_step<D, C>(a, k, true, S);
    }
    S.g_pi = fq_pass(S.pi_raw, a.kc, 3) ? S.g_pi : 0.0f;
#pragma unroll
    for (int l = 0; l < D; ++l) {
        S.g_mu[l] = fq_pass(S.mu_raw[l], a.kc, 1) ? S.g_mu[l] : 0.0f;
#pragma unroll
        for (int m = 0; m <= l; ++m) S.g_A[l][m] = fq_pass(S.Araw[l][m], a.kc, 0) ? S.g_A[l][m] : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) S.g_nu[c] = fq_pass(S.nu_raw[c], a.kc, 2) ? S.g_nu[c] : 0.0f;
#pragma unroll
    for (int i = 0; i < D * C; ++i) S.g_ga[i] = fq_pass(S.ga_raw[i], a.kc, 4) ? S.g_ga[i] : 0.0f;
    kernel_apply<D, C, COH>(a, k, S);
}

template <int D, int C>
__global__ void __launch_bounds__(64) shared_adam_kernel(SharedAdamArgs a) {
    shared_adam_body<D, C>(a, (int)blockIdx.x, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// the whole fit of smoe_shared_fit as ONE launch (single GPU): every batch keeps its workgroup for all n_iters iterations;
// pass -> grid barrier -> gather + Adam step (one wavefront per kernel, spread over the workgroups) -> grid barrier.
// The same device functions in the same order as the two launches per iteration it replaces: bit-identical results.  What one
// workgroup writes for another (the batches' rows, the new parameters) goes through device-scope relaxed atomics (gload / gstore
// <true>), so the barrier is an arrival counter and a generation word without any cache maintenance.  The grid
// must be co-resident (the host checks the occupancy and launches cooperatively); the barrier's spin is bounded -- a workgroup
// that waits longer than ~2 s raises the abort word, every workgroup leaves at its next barrier and the host reports the failure
// instead of a hung device.
// ---------------------------------------------------------------------------------------------
// bar[2]: abort word; bar[16 + w]: arrival flag of workgroup w = the number of the last barrier it has reached.  A workgroup
// arrives with ONE store of its own flag (no read-modify-write on a shared counter: 256 of those in a row were ~5 us) and its
// first wavefront polls all flags, four per lane and load (16-byte device-scope loads), until every workgroup has reached
// barrier `gen`.
__device__ __forceinline__ bool grid_barrier(uint32_t* __restrict__ bar, const int nwg, const int b, const uint32_t gen) {
    // every store of this thread has been written through (the data that crosses workgroups goes through gstore<true>: sc1)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&bar[16 + b], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        uint32_t polls = 0u;
        for (;;) {
            bool ok = true;
            for (int i0 = 4 * lane; i0 < nwg; i0 += 256) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                u4 v;
                const uint32_t* src = bar + 16 + i0;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
                ok = ok && ((int)(v.x - gen) >= 0) && (i0 + 1 >= nwg || (int)(v.y - gen) >= 0)
                        && (i0 + 2 >= nwg || (int)(v.z - gen) >= 0) && (i0 + 3 >= nwg || (int)(v.w - gen) >= 0);
            }
            if (__ballot(!ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1u << 21) || ((polls & 255u) == 0u && __hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) __hip_atomic_store(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return __hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

// SMOE_PHASE_CLOCKS (diagnostic build, scripts/phase_clocks_shared.py): thread 0 of workgroup 0 sums the s_memtime ticks of the
// four phases of an iteration into bar[8 .. 11] (pass, barrier, gather + step, barrier)
#ifndef SMOE_PHASE_CLOCKS
#define SMOE_PHASE_CLOCKS 0
#endif
#if SMOE_PHASE_CLOCKS
#define SMOE_SCLK(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); sclk[i] += (uint32_t)(_t - sclk_last); sclk_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMOE_SCLK(i) do { } while (0)
#endif

template <int D, int C, int PXL, bool IC>
__global__ void __launch_bounds__(SH_THREADS) shared_fit_kernel(SharedFitArgs f) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t epoch = f.pass.epoch;
    float b1p = f.adam.b1p, b2p = f.adam.b2p;
    uint32_t gen = 0u;                       // barriers passed (the flags start at zero)
#if SMOE_PHASE_CLOCKS
    uint32_t sclk[4] = {0u, 0u, 0u, 0u};
    unsigned long long sclk_last = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0; it < f.n_iters; ++it) {
        {
            SharedArgs a = f.pass;
            a.epoch = epoch;
            shared_pass_body<D, C, PXL, true, false, IC, true>(a, b, lds);
        }
        SMOE_SCLK(0);
        if (!grid_barrier(f.bar, nwg, b, ++gen)) return;
        SMOE_SCLK(1);
        {
            SharedAdamArgs ad = f.adam;
            ad.b1p = b1p; ad.b2p = b2p; ad.gather.epoch = epoch;
            for (int k = wave * nwg + b; k < ad.K; k += (SH_THREADS / 64) * nwg) shared_adam_body<D, C, true>(ad, k, lane);
        }
        SMOE_SCLK(2);
        if (!grid_barrier(f.bar, nwg, b, ++gen)) return;
        SMOE_SCLK(3);
        b1p *= f.adam.beta1;
        b2p *= f.adam.beta2;
        epoch += 1u;
        if (epoch == 0u) epoch = 1u;
    }
#if SMOE_PHASE_CLOCKS
    if (b == 0 && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) f.bar[8 + i] = sclk[i];
#endif
}

// ---------------------------------------------------------------------------------------------
// image-wide records of the fake-quantised graph (SharedRangesArgs): ONE workgroup walks the kernels
// ---------------------------------------------------------------------------------------------
static constexpr int RG_THREADS = 1024;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}

// sums (or minima) of NV per-thread values over the workgroup -> out[NV] in LDS, fixed order
template <int NV, bool MIN>
__device__ __forceinline__ void workgroup_reduce(float (&v)[NV], float* s_part /* [waves][NV] */, float* s_out /* [NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float t = MIN ? wave_min(v[i]) : wave_sum(v[i]);
        if (lane == 0) s_part[wave * NV + i] = t;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float t = s_part[threadIdx.x];
        for (int w = 1; w < waves; ++w) t = MIN ? fminf(t, s_part[w * NV + threadIdx.x]) : t + s_part[w * NV + threadIdx.x];
        s_out[threadIdx.x] = t;
    }
    __syncthreads();
}

template <int D, int C>
__device__ __forceinline__ void ranges_of_image(const smoe_params& p, const float* __restrict__ G, int K, const KernelConsts& kc, float* qrng, float* s_part, float* s_out) {
    constexpr float INF = __builtin_huge_valf();
    float ex[11];                                   // lo[0..4], -hi[5..9] of this thread's kernels; [10]: -count
#pragma unroll
    for (int i = 0; i < 10; ++i) ex[i] = INF;
    float cnt = 0.0f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        if (!(fqv(p.pis[k], kc, 3) > 0.0f)) continue;               // pis_mask = qpis > 0, not the kernel lists
        cnt += 1.0f;
        auto see = [&](int t, float x) { ex[t] = fminf(ex[t], x); ex[5 + t] = fminf(ex[5 + t], -x); };
#pragma unroll
        for (int l = 0; l < D; ++l) {
            see(2, mu_off<D>(p.musX, G, kc, k, l));
#pragma unroll
            for (int m = 0; m <= l; ++m)
                see((l == m) ? 0 : 1, (l == m) ? p.A_diagonal[((size_t)k * D + l) * D + m] : p.A_corr[((size_t)k * D + l) * D + m]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) see(3, p.nu_e[(size_t)k * C + c]);
#pragma unroll
        for (int i = 0; i < D * C; ++i) see(4, p.gamma_e[(size_t)k * D * C + i]);
    }
    float cv[1] = {cnt};
    workgroup_reduce<1, false>(cv, s_part, s_out);
    const float total = s_out[0];
    __syncthreads();
    float mv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) mv[i] = ex[i];
    workgroup_reduce<10, true>(mv, s_part, s_out);
    if (threadIdx.x < 5) {
        const int t = threadIdx.x;
        float lo = s_out[t], hi = -s_out[5 + t];
        if (lo == INF) { lo = 0.0f; hi = 0.0f; }                       // no kernel left
        if (t == 1) { lo = fminf(lo, 0.0f); hi = fmaxf(hi, 0.0f); }    // diagonal / upper entries of the A_corr variable (zero)
        const float lv = (t < 2) ? kc.q_levels[0] : ((t == 2) ? kc.q_levels[1] : ((t == 3) ? kc.q_levels[2] : kc.q_levels[4]));
        const FqRange r = fq_vars(lo, hi, lv, t == 0 || t == 3, t == 0 && kc.radial != 0);
        float* o = qrng + t * 8;
        o[0] = r.nmin; o[1] = r.nmax; o[2] = r.scale; o[3] = r.inv;
        o[4] = r.back; o[5] = r.zero ? 1.0f : 0.0f; o[6] = lo; o[7] = hi;
    }
    if (threadIdx.x == 5) qrng[40] = total;
    __threadfence();
    __syncthreads();
}

template <int D, int C>
__global__ void __launch_bounds__(RG_THREADS) shared_ranges_kernel(SharedRangesArgs a) {
    __shared__ float s_part[(RG_THREADS / 64) * 20];
    __shared__ float s_out[20];
    ranges_of_image<D, C>(a.p, a.mus_grid, a.K, a.kc, a.qrng, s_part, s_out);
}

// quantization_mode 3: fake_quant_with_min_max_vars sends the gradient of what falls outside the nudged range to its
// min / max input, reduce_min / reduce_max hand it to the extreme elements of the tensor (split over ties) -- an
// image-wide exchange between the kernels.  ONE workgroup: pass 1 sums the outlying gradients and counts the ties per
// tensor (fixed order), pass 2 routes and applies Adam.  a.qrng holds the records of the parameters BEFORE the step.
template <int D, int C>
__global__ void __launch_bounds__(RG_THREADS) shared_adam_routed_kernel(SharedAdamArgs a) {
    __shared__ float s_part[(RG_THREADS / 64) * 20];
    __shared__ float s_tot[20];
    auto visit = [&](KernelStep<D, C>& S, auto&& f) {      // f(tensor, raw value, gradient&) over the mode-3 variables of a kernel
#pragma unroll
        for (int l = 0; l < D; ++l) {
            if (a.kc.q_musx) f(2, S.mu_raw[l], S.g_mu[l]);
#pragma unroll
            for (int m = 0; m <= l; ++m) f((l == m) ? 0 : 1, S.Araw[l][m], S.g_A[l][m]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) f(3, S.nu_raw[c], S.g_nu[c]);
#pragma unroll
        for (int i = 0; i < D * C; ++i) f(4, S.ga_raw[i], S.g_ga[i]);
    };
    float st[20];                                   // GL[0..4], GA[5..9], ties at lo [10..14], at hi [15..19]
#pragma unroll
    for (int i = 0; i < 20; ++i) st[i] = 0.0f;
    for (int k = threadIdx.x; k < a.K; k += blockDim.x) {
        KernelStep<D, C> S;
        kernel_step<D, C>(a, k, false, S);
        const bool keep = S.pi > 0.0f;
        visit(S, [&](int t, float x, float& g) {
            const float* o = a.qrng + t * 8;
            const bool unshifted = t == 0 && a.kc.radial;        // radial_as steering: see fq_vars / the block kernels
            const float v = unshifted ? x : x - o[4];
            const bool zero = o[5] != 0.0f;
#pragma unroll
            for (int tt = 0; tt < 5; ++tt) {
                const bool hit = tt == t;
                st[tt] += (hit && (unshifted ? !(!zero && v > o[1]) : (!zero && v < o[0]))) ? g : 0.0f;
                st[5 + tt] += (hit && !zero && v > o[1]) ? g : 0.0f;
                st[10 + tt] += (hit && keep && x == o[6]) ? 1.0f : 0.0f;
                st[15 + tt] += (hit && keep && x == o[7]) ? 1.0f : 0.0f;
            }
        });
    }
    workgroup_reduce<20, false>(st, s_part, s_tot);
    for (int k = threadIdx.x; k < a.K; k += blockDim.x) {
        KernelStep<D, C> S;
        kernel_step<D, C>(a, k, true, S);
        const bool keep = S.pi > 0.0f;
        S.g_pi = fq_pass(S.pi_raw, a.kc, 3) ? S.g_pi : 0.0f;
        visit(S, [&](int t, float x, float& g) {
            const float* o = a.qrng + t * 8;
            const float v = (t == 0 && a.kc.radial) ? x : x - o[4];
            const bool zero = o[5] != 0.0f;
            float r = (!zero && (v < o[0] || v > o[1])) ? 0.0f : g;
            r += (keep && x == o[6]) ? s_tot[t] / fmaxf(s_tot[10 + t], 1.0f) : 0.0f;
            r += (keep && x == o[7]) ? s_tot[5 + t] / fmaxf(s_tot[15 + t], 1.0f) : 0.0f;
            g = r;
        });
        kernel_apply<D, C>(a, k, S);
    }
}

// update_kernel_list (smoe.py:2287-2365): probes = {min,max,mid}^d of the batch's coordinates
template <int D>
__global__ void shared_readmit_kernel(SharedReadmitArgs a) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)a.NB * a.K) return;
    const int b = (int)(t / a.K);
    const int k = (int)(t - (long)b * a.K);
    if (!(fqv(a.p.pis[k], a.kc, 3) > 0.0f)) return;
    float A[D][D];
#pragma unroll
    for (int l = 0; l < D; ++l)
#pragma unroll
        for (int m = 0; m < D; ++m)
            A[l][m] = (l == m) ? fqt(a.p.A_diagonal[((size_t)k * D + l) * D + m], a.kc, a.qrng, 0)
                               : ((l > m) ? fqt(a.p.A_corr[((size_t)k * D + l) * D + m], a.kc, a.qrng, 1) : 0.0f);
    int nprobe = 1;
#pragma unroll
    for (int l = 0; l < D; ++l) nprobe *= 3;
    bool near = false;
    for (int q = 0; q < nprobe; ++q) {
        float r[D];
        int rem = q;
#pragma unroll
        for (int l = D - 1; l >= 0; --l) {
            const int sel = rem % 3;
            rem /= 3;
            r[l] = a.probes[((size_t)b * D + l) * 3 + sel] - mu_graph<D>(a.p.musX, a.mus_grid, a.kc, a.qrng, k, l);
        }
        float maha = 0.0f;
#pragma unroll
        for (int m = 0; m < D; ++m) {
            float zz = 0.0f;
            if (a.kc.inverse_cov) {
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
    if (near) atomicOr(&a.lists[(size_t)b * a.KW + (k >> 5)], 1u << (k & 31));
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
size_t shared_lds_bytes(int D, int C, int K, int KW) {
    const int TRI = D * (D + 1) / 2;
    const int SP = TRI + D + 1 + C + D * C;
    const int PK = 1 + D + TRI + C + D * C;
    return sizeof(float) * (((size_t)K + SH_KC * SP + 4 * SH_KC * PK + K + 8 + 8 + KW + 3) & ~(size_t)3);
}

// ssim_opt: tap tables + X, Y [C][Nb] + Wa [5][Nb] + Wb [3][Nb] (3-d batches: a third table, Wb [5][Nb]) behind the regular carve-up
size_t shared_ssim_lds_bytes(int C, int Nb, int bh, int bw, int bt) {
    return sizeof(float) * ((size_t)((11 * (bh + bw + bt) + 3) & ~3) + (size_t)(2 * C + (bt ? 10 : 8)) * Nb);
}

template <int D, int C, int PXL>
static hipError_t launch_pass_t(const SharedArgs& a, bool train, hipStream_t st) {
    size_t shm = shared_lds_bytes(D, C, a.K, a.KW);
    const bool ic = a.kc.inverse_cov != 0;
    auto kern = ic ? (train ? shared_pass_kernel<D, C, PXL, true, false, true> : shared_pass_kernel<D, C, PXL, false, false, true>)
                   : (train ? shared_pass_kernel<D, C, PXL, true> : shared_pass_kernel<D, C, PXL, false>);
    if (a.ssim) {
        kern = ic ? (train ? shared_pass_kernel<D, C, PXL, true, true, true> : shared_pass_kernel<D, C, PXL, false, true, true>)
                  : (train ? shared_pass_kernel<D, C, PXL, true, true> : shared_pass_kernel<D, C, PXL, false, true>);
        shm += shared_ssim_lds_bytes(C, a.Nb, a.batch_shape[0], a.batch_shape[1], (D == 3) ? a.batch_shape[2] : 0);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.NB), dim3(SH_THREADS), shm, st, a);
    return hipGetLastError();
}

template <int D, int C>
static hipError_t launch_pass_dc(const SharedArgs& a, bool train, hipStream_t st) {
    const int pxl = (a.Nb + SH_THREADS - 1) / SH_THREADS;
    if (pxl <= 1) return launch_pass_t<D, C, 1>(a, train, st);
    if (pxl <= 2) return launch_pass_t<D, C, 2>(a, train, st);
    if (pxl <= 4) return launch_pass_t<D, C, 4>(a, train, st);
    if (C == 1 && pxl <= 8) return launch_pass_t<D, 1, 8>(a, train, st);
    return hipErrorInvalidValue;
}

bool shared_supported(int D, int C, int Nb) {
    if (!((D == 2 || D == 3) && (C == 1 || C == 3))) return false;
    const int pxl = (Nb + SH_THREADS - 1) / SH_THREADS;
    return pxl <= (C == 1 ? 8 : 4);
}

hipError_t launch_shared_pass(const SharedArgs& a, int D, int C, bool train, hipStream_t st) {
    if (D == 2 && C == 1) return launch_pass_dc<2, 1>(a, train, st);
    if (D == 2 && C == 3) return launch_pass_dc<2, 3>(a, train, st);
    if (D == 3 && C == 1) return launch_pass_dc<3, 1>(a, train, st);
    if (D == 3 && C == 3) return launch_pass_dc<3, 3>(a, train, st);
    return hipErrorInvalidValue;
}

template <int D, int C, int PXL>
static hipError_t launch_fit_t(const SharedFitArgs& f, int num_cus, hipStream_t st) {
    const size_t shm = shared_lds_bytes(D, C, f.pass.K, f.pass.KW);
    auto kern = (f.pass.kc.inverse_cov != 0) ? shared_fit_kernel<D, C, PXL, true> : shared_fit_kernel<D, C, PXL, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SH_THREADS, shm);
    if (e != hipSuccess) return e;
    if ((long)per_cu * num_cus < (long)f.pass.NB) return hipErrorCooperativeLaunchTooLarge;     // the barrier needs every batch resident
    SharedFitArgs args = f;
    void* argv[] = {&args};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(f.pass.NB), dim3(SH_THREADS), argv, (unsigned int)shm, st);
}

template <int D, int C>
static hipError_t launch_fit_dc(const SharedFitArgs& f, int num_cus, hipStream_t st) {
    const int pxl = (f.pass.Nb + SH_THREADS - 1) / SH_THREADS;
    if (pxl <= 1) return launch_fit_t<D, C, 1>(f, num_cus, st);
    if (pxl <= 2) return launch_fit_t<D, C, 2>(f, num_cus, st);
    if (pxl <= 4) return launch_fit_t<D, C, 4>(f, num_cus, st);
    if (C == 1 && pxl <= 8) return launch_fit_t<D, 1, 8>(f, num_cus, st);
    return hipErrorInvalidValue;
}

hipError_t launch_shared_fit(const SharedFitArgs& f, int D, int C, int num_cus, hipStream_t st) {
    if (f.pass.ssim || f.adam.kc.qmode == 3 || f.pass.part == nullptr) return hipErrorInvalidValue;
    if (D == 2 && C == 1) return launch_fit_dc<2, 1>(f, num_cus, st);
    if (D == 2 && C == 3) return launch_fit_dc<2, 3>(f, num_cus, st);
    if (D == 3 && C == 1) return launch_fit_dc<3, 1>(f, num_cus, st);
    if (D == 3 && C == 3) return launch_fit_dc<3, 3>(f, num_cus, st);
    return hipErrorInvalidValue;
}

hipError_t launch_shared_adam(const SharedAdamArgs& a, int D, int C, hipStream_t st) {
    if (a.kc.qmode == 3) {                            // one workgroup: the fake-quant backward couples the kernels
        if (D == 2 && C == 1) hipLaunchKernelGGL((shared_adam_routed_kernel<2, 1>), dim3(1), dim3(RG_THREADS), 0, st, a);
        else if (D == 2 && C == 3) hipLaunchKernelGGL((shared_adam_routed_kernel<2, 3>), dim3(1), dim3(RG_THREADS), 0, st, a);
        else if (D == 3 && C == 1) hipLaunchKernelGGL((shared_adam_routed_kernel<3, 1>), dim3(1), dim3(RG_THREADS), 0, st, a);
        else if (D == 3 && C == 3) hipLaunchKernelGGL((shared_adam_routed_kernel<3, 3>), dim3(1), dim3(RG_THREADS), 0, st, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    const int threads = 64;
    const int grid = a.K;                              // one wavefront per kernel
    if (D == 2 && C == 1) hipLaunchKernelGGL((shared_adam_kernel<2, 1>), dim3(grid), dim3(threads), 0, st, a);
    else if (D == 2 && C == 3) hipLaunchKernelGGL((shared_adam_kernel<2, 3>), dim3(grid), dim3(threads), 0, st, a);
    else if (D == 3 && C == 1) hipLaunchKernelGGL((shared_adam_kernel<3, 1>), dim3(grid), dim3(threads), 0, st, a);
    else if (D == 3 && C == 3) hipLaunchKernelGGL((shared_adam_kernel<3, 3>), dim3(grid), dim3(threads), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_shared_gather(const SharedGatherArgs& g, hipStream_t st) {
    // PK = 1 + D + D (D + 1) / 2 + C + D C for (D, C) in {2, 3} x {1, 3}
    if (g.PK == SL<2, 1>::PK) hipLaunchKernelGGL((shared_gather_kernel<SL<2, 1>::PK>), dim3(g.K), dim3(64), 0, st, g);
    else if (g.PK == SL<2, 3>::PK) hipLaunchKernelGGL((shared_gather_kernel<SL<2, 3>::PK>), dim3(g.K), dim3(64), 0, st, g);
    else if (g.PK == SL<3, 1>::PK) hipLaunchKernelGGL((shared_gather_kernel<SL<3, 1>::PK>), dim3(g.K), dim3(64), 0, st, g);
    else if (g.PK == SL<3, 3>::PK) hipLaunchKernelGGL((shared_gather_kernel<SL<3, 3>::PK>), dim3(g.K), dim3(64), 0, st, g);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_shared_ranges(const SharedRangesArgs& a, int D, int C, hipStream_t st) {
    if (D == 2 && C == 1) hipLaunchKernelGGL((shared_ranges_kernel<2, 1>), dim3(1), dim3(RG_THREADS), 0, st, a);
    else if (D == 2 && C == 3) hipLaunchKernelGGL((shared_ranges_kernel<2, 3>), dim3(1), dim3(RG_THREADS), 0, st, a);
    else if (D == 3 && C == 1) hipLaunchKernelGGL((shared_ranges_kernel<3, 1>), dim3(1), dim3(RG_THREADS), 0, st, a);
    else if (D == 3 && C == 3) hipLaunchKernelGGL((shared_ranges_kernel<3, 3>), dim3(1), dim3(RG_THREADS), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_shared_readmit(const SharedReadmitArgs& a, int D, hipStream_t st) {
    const int threads = 256;
    const long total = (long)a.NB * a.K;
    const int grid = (int)((total + threads - 1) / threads);
    if (D == 2) hipLaunchKernelGGL(shared_readmit_kernel<2>, dim3(grid), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL(shared_readmit_kernel<3>, dim3(grid), dim3(threads), 0, st, a);
    return hipGetLastError();
}

}  // namespace smoe
