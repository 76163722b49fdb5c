// smoe_duo.hip.h -- fit kernel for batches of at most a few blocks per SIMD: ONE block on the two wavefronts of a workgroup,
// both wavefronts doing the same work ("duo" tiling).
//
// Why (profiles/r03/phase_clocks_before.txt): with one block per SIMD the iteration is a chain of LDS
// hand-offs on one wavefront -- two transpose-reduction rounds per wavefront, the helper's totals handed over, descriptors,
// parameters and Adam slots read back from LDS, written, published -- and the second wavefront of fit_kernel<PAIR> idles through
// the owner phase.  Here
//   * both wavefronts sweep alternate 64-pixel steps and write their partial sums into ONE joint scratch (row = slot, 128 columns);
//   * after ONE workgroup barrier the rows are summed by lane pairs (each lane half a row, one DPP exchange): the slots of the
//     first half of the kernels by wavefront 0, of the second half by wavefront 1 -- a single hand-off reduces the block AND
//     merges the two wavefronts;
//   * the lane that holds a slot's total OWNS the slot and keeps its parameter, its Adam slots and its gradient descriptor in
//     REGISTERS for the whole launch (nothing is read back per iteration); raw totals are published inside the wavefront for
//     the gradients that involve the neighbouring slots of the same kernel (eval_slot_desc), new parameters go to the OTHER
//     parameter buffer (second barrier; the other wavefront may still be reading the old values).
// Two workgroup barriers and about five LDS round trips per iteration instead of two barriers and about twelve round trips.
// Graph coverage as fit_kernel<PAIR>: the margin-loss graph with quantization_mode 0 / 1 (quantize_pis included), loss weights,
// l1 terms, clipping, trainable flags, only_y_gamma, kernel_count_as_norm_l1, pixel sub-samples; not radial_as /
// train_inverse_cov / ssim_opt / quantization_mode 2, 3.  Triples with at most 128 slots.
#ifndef SMOE_DUO_HIP_H
#define SMOE_DUO_HIP_H

#include "smoe_block.hip.h"
#include "smoe_team.hip.h"

namespace smoe {

template <int D, int C, int K>
struct DuoTile {
    using Lt = Layout<D, C, K>;
    static constexpr int ROWW = 128 + 4;                              // a row of the joint scratch: 64 partials per wavefront + pad
    // Which wavefront sums (and then owns) which slot: the kernels are split in two halves, ALL slots of a kernel go to one
    // wavefront -- the gradient of a slot involves raw totals of the same kernel only (eval_slot_desc), so the totals are
    // exchanged inside the wavefront (no workgroup barrier).  Wavefront 0: kernels [0, KH), the loss slot, the influence
    // counters of its kernels; wavefront 1: kernels [KH, K), the SSE slot, the influence counters of its kernels.
    static constexpr int KH = (K + 1) / 2;
    static constexpr int ROWS0 = KH * Lt::PK + 1 + KH;
    static constexpr int ROWS1 = (K - KH) * Lt::PK + 1 + (K - KH);
    static constexpr int SPL = ((ROWS0 > ROWS1 ? ROWS0 : ROWS1) + 31) / 32;      // slots per owning lane (rounds of 32 rows)
    static constexpr bool OK = SPL <= 2;
    __host__ __device__ static constexpr int slot_of(int wave, int rr) {        // NSLOT = no slot
        const int nk = (wave == 0) ? KH * Lt::PK : (K - KH) * Lt::PK;
        const int k0 = (wave == 0) ? 0 : KH, kn = (wave == 0) ? KH : K - KH;
        if (rr < nk) return k0 * Lt::PK + rr;
        if (rr == nk) return (wave == 0) ? Lt::S_LOSS : Lt::S_SSE;
        if (rr - nk - 1 < kn) return Lt::S_CNT + k0 + (rr - nk - 1);
        return Lt::NSLOT;
    }
    static constexpr int TOT_STRIDE = round_up(Lt::NSLOT, 4);
    __host__ __device__ static int off_par(int N, int CR) { return round_up(CR * N, 4); }                 // two parameter buffers
    static constexpr int DER_STRIDE = round_up(Lt::NPAR, 4);          // derived constants of the block, packed like the parameters
    __host__ __device__ static int off_der(int N, int CR) { return off_par(N, CR) + 2 * Lt::LP_STRIDE; }
    __host__ __device__ static int off_tot(int N, int CR) { return off_der(N, CR) + DER_STRIDE; }
    __host__ __device__ static int off_scr(int N, int CR) { return off_tot(N, CR) + TOT_STRIDE; }
    __host__ __device__ static int off_tgt(int N, int CR) { return off_scr(N, CR) + Lt::NSLOT * ROWW; }
    __host__ __device__ static int off_lw(int N, int CR) { return off_tgt(N, CR) + C * N; }
    __host__ __device__ static size_t bytes(int N, bool has_lw, int CR) {
        return sizeof(float) * (size_t)round_up(off_lw(N, CR) + (has_lw ? N : 0), 4);
    }
};


template <int D, int C, int K, int HL>
__global__ void __launch_bounds__(128) fit_duo_kernel(FitArgs a) {
    using Lt = Layout<D, C, K>;
    using DT = DuoTile<D, C, K>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;                          // one block per workgroup
    constexpr int CR = D - HL;
    const bool has_lw = a.loss_w != nullptr;

    float* s_coords = lds;
    float* s_par0 = lds + DT::off_par(N, CR);          // parameter buffer 0; buffer 1 = + LP_STRIDE
    float* s_der = lds + DT::off_der(N, CR);
    float* s_tot = lds + DT::off_tot(N, CR);
    float* s_scr = lds + DT::off_scr(N, CR);
    float* s_tgt = lds + DT::off_tgt(N, CR);
    float* s_lw = lds + DT::off_lw(N, CR);
    SMOE_LDS_CHECK(DT::off_lw(N, CR) + (has_lw ? N : 0), 16u);
    // the wavefront's 64 columns of the joint scratch as an M0 base for ds_write_addtid_b32 (0: scratch beyond the first 64 KB)
    const uint32_t scr_end = lds_address(s_scr) + (uint32_t)(Lt::NSLOT * DT::ROWW * sizeof(float));
    const uint32_t scr_m0 = (scr_end <= 65536u) ? __builtin_amdgcn_readfirstlane(lds_address(s_scr) + (uint32_t)(wave * 64 * sizeof(float))) : 0u;

    // ---- staging ------------------------------------------------------------------------------------------------------------
    for (int i = threadIdx.x; i < CR * N; i += 128) s_coords[i] = a.coords[i];
    for (int i = threadIdx.x; i < C * N; i += 128) s_tgt[i] = a.target[(size_t)b * C * N + i];
    if (has_lw)
        for (int i = threadIdx.x; i < N; i += 128) s_lw[i] = a.loss_w[(size_t)b * N + i];
    float xc[D];                                       // coordinates of the lane's pixel i = 0 (hoisted axes: of all its pixels)
#pragma unroll
    for (int l = 0; l < D; ++l) xc[l] = a.coords[l * N + min(lane, N - 1)];

    const KernelConsts kc = a.kc;
    const bool patch_pis = kc.qpis != 0;
    const bool has_reg = (a.reg_pi != 0.0f) || (a.reg_u != 0.0f);
    const bool has_loss0 = a.loss0 != nullptr;
    const float beta1 = a.beta1, beta2 = a.beta2, adam_eps = a.eps, clip = a.clip;
    const float reg_pi = a.reg_pi, reg_u = a.reg_u;
    constexpr uint32_t PB1 = (uint32_t)(Lt::LP_STRIDE * sizeof(float));
    const uint32_t par_bytes = (uint32_t)((s_par0 - lds) * sizeof(float));
    const uint32_t tot_bytes = (uint32_t)((s_tot - lds) * sizeof(float));
    auto quant_pi = [&](float x) {
        const float cl = fminf(fmaxf(x, kc.q_nmin[3]), kc.q_nmax[3]);
        return floorf((cl - kc.q_nmin[3]) * kc.q_inv[3] + 0.5f) * kc.q_scale[3] + kc.q_nmin[3];
    };

    // Derived constants (what BlockRegs::derive computes from the parameters, same operation order): published by the slot
    // owners after every step instead of being re-derived by all 128 lanes.  Own slot only: A' = SQ A, nu, gamma (masked);
    // with the neighbouring slots of the same kernel (same wavefront): c_m = sum_{l >= m} mu_l A'_lm, coef.
    auto derived_own = [&](int kind, float v, bool masked) {
        return (kind == 2 || kind == 3) ? SMOE_SQ * v : ((kind == 4 && masked) ? 0.0f : v);
    };
    // c_m = sum_{l >= m} mu_l A'_lm and coef = act ? pi prod diag(A) / sqrt((2 pi)^d) : 0 from ONE batch of LDS reads with per-lane
    // operand offsets (xo / yo pairs, prior, list flag; unused operands point at the constant cells), both forms evaluated by every
    // lane and selected: a divergent branch per slot kind cost two more LDS round trips per iteration
    auto derive_cross = [&](const float* __restrict__ pp, const int (&xo)[D], const int (&yo)[D], int po, int fo, bool is_pi) {
        float xs[D], ys[D];
#pragma unroll
        for (int l = 0; l < D; ++l) { xs[l] = pp[xo[l]]; ys[l] = pp[yo[l]]; }
        const float piv = pp[po], flg = pp[fo];
        float c = 0.0f, det = 1.0f;
#pragma unroll
        for (int l = 0; l < D; ++l) {
            c = fmaf(xs[l], SMOE_SQ * ys[l], c);
            det *= xs[l];
        }
        const float nq = kc.use_det ? det * kc.inv_n_dis : 1.0f;
        const bool act = (flg != 0.0f) & (piv > 0.0f);
        return is_pi ? (act ? nq * piv : 0.0f) : c;
    };

    // ---- owner set-up: lane pair (2r, 2r+1) of wavefront w sums the row of slot slot_of(w, r + 32 q) in round q; the even lane owns it ---
    const int r = lane >> 1, half = lane & 1;
    const bool owner_lane = half == 0;
    float pv[DT::SPL], mv[DT::SPL], vv[DT::SPL], lr[DT::SPL], regc[DT::SPL];
    int tens[DT::SPL];           // kind | kernel << 4; kind = tensor 0..5 of a parameter, 8 loss, 9 SSE, 10 influence counter, 15 none
    int wro[DT::SPL], rdo[DT::SPL];      // float offsets inside a parameter buffer: what the lane writes / the old value it reads
    int cxo[DT::SPL][D], cyo[DT::SPL][D], cpo[DT::SPL], cfo[DT::SPL];    // operands of derive_cross (centre / prior slots)
    SlotDesc dsc[DT::SPL];
#pragma unroll
    for (int q = 0; q < DT::SPL; ++q) {
        const int j = DT::slot_of(wave, r + 32 * q);
        pv[q] = mv[q] = vv[q] = lr[q] = regc[q] = 0.0f;
        tens[q] = 15;
        wro[q] = Lt::LP_ZERO; rdo[q] = Lt::LP_ZERO;
#pragma unroll
        for (int l = 0; l < D; ++l) { cxo[q][l] = Lt::LP_ZERO; cyo[q][l] = Lt::LP_ZERO; }
        cpo[q] = Lt::LP_ONE; cfo[q] = Lt::LP_ONE;
        {   // a descriptor whose every operand is the zero cell: lanes that own no parameter evaluate it to 0
            const uint32_t Z = par_bytes + 4u * Lt::LP_ZERO;
            dsc[q] = SlotDesc{0.0f, 1.0f, Z, Z, Z, Z, Z, Z, Z, par_bytes + 4u * Lt::LP_ONE, Z, Z};
        }
        if (!owner_lane) continue;
        if (j < Lt::NPAR) {
            int tensor, kern; long off;
            decode_slot<D, C, K>(j, b, tensor, off, kern);
            pv[q] = pick(a.p, tensor)[off];
            mv[q] = pick(a.m, tensor)[off];
            vv[q] = pick(a.v, tensor)[off];
            s_par0[j] = pv[q];
            if (tensor == 0) s_par0[Lt::LP_QPI + kern] = patch_pis ? quant_pi(pv[q]) : pv[q];
            int chan = 0;
            if (tensor == 4) chan = (int)(off % C);
            float rate = (tensor == 0) ? a.lr_pis : ((tensor == 2 || tensor == 3) ? a.lr_steer : a.lr_expert);   // smoe.py:1102-1104
            if (tensor == 0 && !a.train_pis) rate = 0.0f;
            if (tensor == 1 && !a.train_musx) rate = 0.0f;
            if (tensor == 4 && (!kc.train_gammas || (kc.only_y_gamma && chan != 0))) rate = 0.0f;
            lr[q] = rate;
            regc[q] = (tensor == 0) ? reg_pi : ((tensor == 2) ? reg_u : 0.0f);                                   // smoe.py:1027,1044
            const bool masked = tensor == 4 && (!kc.train_gammas || (kc.only_y_gamma && chan != 0));             // smoe.py:841-848,725-729
            tens[q] = tensor | (kern << 4) | (masked ? 4096 : 0);
            if (tensor >= 2) s_der[j] = derived_own(tensor, pv[q], masked);
            wro[q] = j;
            if (tensor == 0) {                         // coef: x_l = A_ll (their product), prior as the graph reads it, list flag
#pragma unroll
                for (int l = 0; l < D; ++l) { cxo[q][l] = kern * Lt::PK + Lt::O_A + tri_index(l, l); cyo[q][l] = Lt::LP_ZERO; }
                cpo[q] = patch_pis ? Lt::LP_QPI + kern : kern * Lt::PK + Lt::O_PI;
                cfo[q] = Lt::LP_ACT + kern;
            } else if (tensor == 1) {                  // c_m = sum_{l >= m} mu_l A'_lm
                const int m = (j - kern * Lt::PK) - Lt::O_MU;
#pragma unroll
                for (int l = 0; l < D; ++l) {
                    cxo[q][l] = (l >= m) ? kern * Lt::PK + Lt::O_MU + l : Lt::LP_ZERO;
                    cyo[q][l] = (l >= m) ? kern * Lt::PK + Lt::O_A + l * (l + 1) / 2 + m : Lt::LP_ZERO;
                }
            }
            dsc[q] = build_slot_desc<D, C, K, false>(j, tot_bytes, par_bytes, par_bytes, patch_pis, kc.use_det != 0);
        } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
            s_par0[Lt::LP_ACT + (j - Lt::S_CNT)] = ((a.active[b] >> (j - Lt::S_CNT)) & 1u) ? 1.0f : 0.0f;
            tens[q] = 10 | ((j - Lt::S_CNT) << 4);
            wro[q] = rdo[q] = Lt::LP_ACT + (j - Lt::S_CNT);
        } else if (j == Lt::S_SSE) {
            tens[q] = 9;
        } else if (j == Lt::S_LOSS) {
            tens[q] = 8;
            wro[q] = Lt::LP_FROZEN;
            s_par0[Lt::LP_FROZEN] = (a.diverged != nullptr && a.diverged[b] != 0u) ? 1.0f : 0.0f;
            s_par0[Lt::LP_ZERO] = 0.0f; s_par0[Lt::LP_ONE] = 1.0f;
            s_par0[Lt::LP_STRIDE + Lt::LP_ZERO] = 0.0f; s_par0[Lt::LP_STRIDE + Lt::LP_ONE] = 1.0f;
        }
    }
    const float loss0 = has_loss0 ? a.loss0[b] : 0.0f;
    float b1p = a.b1p, b2p = a.b2p;
    float last_loss = 0.0f, last_sse = 0.0f;
    int cur = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DT::SPL; ++q) {
        const int j = DT::slot_of(wave, r + 32 * q);
        const float dvx = derive_cross(s_par0, cxo[q], cyo[q], cpo[q], cfo[q], (tens[q] & 15) == 0);
        if (owner_lane && j < Lt::NPAR && (tens[q] & 15) < 2) s_der[j] = dvx;
    }
    __syncthreads();
#if SMOE_PHASE_CLOCKS
    float clk[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    unsigned long long clk_last = __builtin_amdgcn_s_memtime();
#endif

    for (int it = 0; it < a.n_iters; ++it) {
        // One-round launches (smoe_block.hip.h, rotate_priority): two wavefronts share a SIMD here.  Both parts of the iteration
        // rotate between two levels, the second part (partial sums, row sums, owner phase: LDS round trips, few instructions)
        // always above the pixel part of the other wavefront: ONE image 110 -> 116 Gpx-it/s (plain rotation: 108 -> 110).  On
        // fit_kernel's one-block-per-wavefront form the same scheme loses (2 048 blocks -2 %, three per SIMD -35 %).
        const uint32_t r2 = ((uint32_t)it + hw_wave_slot()) & 1u;
        if (a.prio_rotate) { if (r2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        const float* par = s_par0 + cur * Lt::LP_STRIDE;
        float* parn = s_par0 + (cur ^ 1) * Lt::LP_STRIDE;
        const uint32_t pb = cur ? PB1 : 0u;
        // ---- forward + backward over the wavefront's pixel steps (n = i * 128 + wave * 64 + lane) -------------------------------
        {
            float acc[Lt::NSLOT];
#pragma unroll
            for (int j = 0; j < Lt::NSLOT; ++j) acc[j] = 0.0f;
            {
                BlockRegs<D, C, K> R;
                {
                    float dv[DT::DER_STRIDE];
                    const float4* src = reinterpret_cast<const float4*>(s_der);
#pragma unroll
                    for (int i = 0; i < DT::DER_STRIDE / 4; ++i) {
                        const float4 v = src[i];
                        dv[4 * i + 0] = v.x; dv[4 * i + 1] = v.y; dv[4 * i + 2] = v.z; dv[4 * i + 3] = v.w;
                    }
#pragma unroll
                    for (int i = 0; i < Lt::LP_STRIDE; ++i) R.P[i] = 0.0f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        R.coef[k] = dv[k * Lt::PK + Lt::O_PI];
#pragma unroll
                        for (int m = 0; m < D; ++m) R.cz[k][m] = dv[k * Lt::PK + Lt::O_MU + m];
#pragma unroll
                        for (int t = 0; t < Lt::TRI; ++t) R.As[k][t] = dv[k * Lt::PK + Lt::O_A + t];
#pragma unroll
                        for (int i = 0; i < C + D * C; ++i) R.P[k * Lt::PK + Lt::O_NU + i] = dv[k * Lt::PK + Lt::O_NU + i];
                    }
                }
                if (HL > 0) hoist_const<D, C, K, HL, false>(R, xc);
                SMOE_CLK(0);
                if (has_lw) pixel_loop_train<D, C, K, true, HL, false>(R, kc, s_coords, s_tgt, s_lw, N, 128, wave * 64 + lane, acc, a.lw_is_sample != 0);
                else pixel_loop_train<D, C, K, false, HL, false>(R, kc, s_coords, s_tgt, s_lw, N, 128, wave * 64 + lane, acc);
                if (HL > 0) complete_const<D, C, K, HL, false>(R, xc, acc);
                SMOE_CLK(1);
            }
            if (a.prio_rotate) { if (r2) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
            // ---- joint scratch: row j holds the 128 partial sums of slot j ---------------------------------------------------------
            if (scr_m0 != 0u) {
                addtid_store_rows<DT::ROWW * (int)sizeof(float), Lt::NSLOT>(acc, scr_m0);
            } else {
#pragma unroll
                for (int j = 0; j < Lt::NSLOT; ++j) s_scr[j * DT::ROWW + threadIdx.x] = acc[j];
            }
        }
        SMOE_CLK(2);
        __syncthreads();
        SMOE_CLK(3);
        // ---- one hand-off: each lane sums half a row, the pair exchanges (quad_perm [1,0,3,2]) ------------------------------------
        float T[DT::SPL];
#pragma unroll
        for (int q = 0; q < DT::SPL; ++q) {
            const int j = DT::slot_of(wave, r + 32 * q);
            const int jr = (j < Lt::NSLOT) ? j : (Lt::NSLOT - 1);                 // rows past the last slot: clamp (their sums are not used)
            // packed adds (v_pk_add_f32: two sums per instruction): this kernel is bound by the instruction count of its two
            // wavefronts, not by flops
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float4* row = reinterpret_cast<const float4*>(s_scr + jr * DT::ROWW + half * 64);
            float4 rv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = row[i];
            // all sixteen reads in flight before the first add (the accumulators are dead here: the registers are there); left to
            // itself the scheduler recycles four registers and waits for every read in turn
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 34, 0);
            f2 lo = {rv[0].x, rv[0].y}, hi = {rv[0].z, rv[0].w};
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                lo += f2{rv[i].x, rv[i].y};
                hi += f2{rv[i].z, rv[i].w};
            }
            lo += hi;
            float t = lo.x + lo.y;
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xf, 0xf, true));
            T[q] = t;
            if (owner_lane && j < Lt::NPAR) s_tot[j] = t * dsc[q].pub_scale;        // suz = suz' / SQ for the centre slots
        }
        SMOE_CLK(4);
        wave_lds_sync();              // the totals a slot's gradient involves belong to the same kernel = the same wavefront
        SMOE_CLK(5);
        // ---- owner phase: gradient, l1 terms, clip, TF1 ApplyAdam (smoe.py:1173-1193), prune (1763-1766), stop test (1565-1570) -----
        // Branch-free over the slot kinds (parameter / loss / SSE / influence counter): every owning lane runs the same code on
        // its own operands, so that all LDS reads of the phase go out together and one wait covers them (a divergent branch per
        // kind serialised three read-wait-compute-write sequences: 1 400 of the 5 100 ticks of an iteration).
        const float bias = __builtin_amdgcn_sqrtf(1.0f - b2p) * fast_rcp(1.0f - b1p);
        const bool frozen = par[Lt::LP_FROZEN] != 0.0f;
        float reg_loss = 0.0f, cntpi = 1.0f;
        if (has_reg) {                                                              // wave-uniform
            cntpi = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) cntpi += ((patch_pis ? par[Lt::LP_QPI + kk] : par[kk * Lt::PK + Lt::O_PI]) > 0.0f) ? 1.0f : 0.0f;
            cntpi = fmaxf(cntpi, 1.0f);
            const float rp = kc.kcount_norm ? kc.pis_l1_raw / cntpi : reg_pi;       // smoe.py:1022-1027
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const float piv = patch_pis ? par[Lt::LP_QPI + kk] : par[kk * Lt::PK + Lt::O_PI];
                const bool act = (par[Lt::LP_ACT + kk] != 0.0f) && (piv > 0.0f);
                float term = rp * piv;
#pragma unroll
                for (int l = 0; l < D; ++l) term += reg_u * par[kk * Lt::PK + Lt::O_A + tri_index(l, l)];
                reg_loss += act ? term : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < DT::SPL; ++q) {
            const int kind = tens[q] & 15, k = (tens[q] >> 4) & 255;  // 0..5: parameter of that tensor; 8 loss, 9 SSE, 10 influence counter, 15 none
            const bool is_par = kind < 8;
            const float oldv = par[rdo[q]];                           // influence counters: the old list flag (zero cell otherwise)
            float gsum = team_eval_desc(lds, dsc[q], T[q], pb);
            if (has_reg) {                                            // smoe.py:1027,1044 (active kernels only)
                const int kc_ = is_par ? k : 0;
                const float piv = patch_pis ? par[Lt::LP_QPI + kc_] : par[kc_ * Lt::PK + Lt::O_PI];
                const bool act = (par[Lt::LP_ACT + kc_] != 0.0f) && (piv > 0.0f);
                const float rs = (kc.kcount_norm && kind == 0) ? kc.pis_l1_raw / cntpi : regc[q];
                gsum += act ? rs : 0.0f;
            }
            if (patch_pis) gsum = (kind != 0 || (pv[q] >= kc.q_nmin[3] && pv[q] <= kc.q_nmax[3])) ? gsum : 0.0f;   // straight-through range
            if (clip > 0.0f) gsum = fminf(fmaxf(gsum, -clip), clip);
            const float alpha = lr[q] * bias;
            const float m2 = mv[q] + (gsum - mv[q]) * (1.0f - beta1);
            const float v2 = vv[q] + (gsum * gsum - vv[q]) * (1.0f - beta2);
            const float p2 = pv[q] - (m2 * alpha) * fast_rcp(__builtin_amdgcn_sqrtf(v2) + adam_eps);
            const bool upd = (lr[q] != 0.0f) && !frozen;              // lr is 0 for everything that is not a trained parameter
            pv[q] = upd ? p2 : pv[q];
            mv[q] = upd ? m2 : mv[q];
            vv[q] = upd ? v2 : vv[q];
            const float lossv = T[q] + reg_loss;
            const bool bad = (kind == 8) && !frozen && ((lossv != lossv) || (has_loss0 && (lossv + 1.0f > (loss0 + 100.0f) * 10.0f)));
            last_loss = (kind == 8 && !frozen) ? lossv : last_loss;
            last_sse = (kind == 9 && !frozen) ? T[q] : last_sse;
            const float wv = is_par ? pv[q]
                           : ((kind == 8) ? ((frozen || bad) ? 1.0f : 0.0f)                 // takes effect from the next iteration
                           : (frozen ? oldv : ((T[q] > 0.0f) ? 1.0f : 0.0f)));              // smoe.py:829,1763-1766
            if (owner_lane && (is_par || kind == 8 || kind == 10)) parn[wro[q]] = wv;
            if (owner_lane && kind == 0) parn[Lt::LP_QPI + k] = patch_pis ? quant_pi(pv[q]) : pv[q];
            if (owner_lane && is_par && kind >= 2) s_der[wro[q]] = derived_own(kind, pv[q], (tens[q] & 4096) != 0);
        }
        wave_lds_sync();              // the new values of the same kernel's slots (same wavefront) for c = A'^T mu and coef
#pragma unroll
        for (int q = 0; q < DT::SPL; ++q) {
            const int kind = tens[q] & 15;
            const float dvx = derive_cross(parn, cxo[q], cyo[q], cpo[q], cfo[q], kind == 0);
            if (owner_lane && kind < 2) s_der[wro[q]] = dvx;
        }
        SMOE_CLK(6);
        __syncthreads();
        cur ^= 1;
        b1p *= beta1;
        b2p *= beta2;
        SMOE_CLK(7);
    }
#if SMOE_PHASE_CLOCKS
    if (blockIdx.x == 0 && lane == 0 && a.loss_out != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.loss_out[a.B / 2 + wave * 8 + i] = clk[i];
    }
    return;
#endif

    // ---- write back: the owners' registers ----------------------------------------------------------------------------------------
    const float* par = s_par0 + cur * Lt::LP_STRIDE;
#pragma unroll
    for (int q = 0; q < DT::SPL; ++q) {
        const int j = DT::slot_of(wave, r + 32 * q);
        if (!owner_lane || j >= Lt::NSLOT) continue;
        if (j < Lt::NPAR) {
            int tensor, kern; long off;
            decode_slot<D, C, K>(j, b, tensor, off, kern);
            pick(a.p, tensor)[off] = pv[q];
            pick(a.m, tensor)[off] = mv[q];
            pick(a.v, tensor)[off] = vv[q];
        } else if (j == Lt::S_LOSS) {
            if (a.loss_out != nullptr && a.n_iters > 0) a.loss_out[b] = last_loss;
            if (a.diverged != nullptr) a.diverged[b] = (par[Lt::LP_FROZEN] != 0.0f) ? 1u : 0u;
            uint32_t mask = 0u;
#pragma unroll
            for (int k = 0; k < K; ++k) mask |= (par[Lt::LP_ACT + k] != 0.0f) ? (1u << k) : 0u;
            a.active[b] = mask;
        } else if (j == Lt::S_SSE) {
            if (a.sse_out != nullptr && a.n_iters > 0) a.sse_out[b] = last_sse;
        }
    }
}

template <int D, int C, int K>
size_t duo_lds_bytes(int N, bool has_lw, int hoist) {
    if (!DuoTile<D, C, K>::OK) return (size_t)-1;
    return DuoTile<D, C, K>::bytes(N, has_lw, D - ((hoist >= 1) ? 1 : 0));
}

template <int D, int C, int K>
hipError_t launch_fit_duo(const FitArgs& a, int hoist, hipStream_t st) {
    if constexpr (DuoTile<D, C, K>::OK) {
        using DT = DuoTile<D, C, K>;
        auto kern = (hoist >= 1) ? fit_duo_kernel<D, C, K, 1> : fit_duo_kernel<D, C, K, 0>;
        const size_t shm = DT::bytes(a.N, a.loss_w != nullptr, D - ((hoist >= 1) ? 1 : 0));
        hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
        if (e != hipSuccess) return e;
        FitArgs aa = a;
        aa.lds_floats = (int)(shm / sizeof(float));
        hipLaunchKernelGGL(kern, dim3(a.B), dim3(128), shm, st, aa);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int D, int C, int K>
int duo_occupancy(int N, bool has_lw, int hoist) {
    if constexpr (DuoTile<D, C, K>::OK) {
        auto kern = (hoist >= 1) ? fit_duo_kernel<D, C, K, 1> : fit_duo_kernel<D, C, K, 0>;
        const size_t shm = DuoTile<D, C, K>::bytes(N, has_lw, D - ((hoist >= 1) ? 1 : 0));
        int nb = 0;
        if (allow_lds(reinterpret_cast<const void*>(kern), shm) != hipSuccess) return -1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 128, shm) != hipSuccess) return -1;
        return nb * 2;
    } else {
        return -1;
    }
}

// table entries of the 64-lane variants; other tilings: null
template <int D, int C, int K, int G>
constexpr auto duo_fit_ptr() -> hipError_t (*)(const FitArgs&, int, hipStream_t) {
    if constexpr (G == 64) return &launch_fit_duo<D, C, K>;
    else return nullptr;
}
template <int D, int C, int K, int G>
constexpr auto duo_lds_ptr() -> size_t (*)(int, bool, int) {
    if constexpr (G == 64) return &duo_lds_bytes<D, C, K>;
    else return nullptr;
}
template <int D, int C, int K, int G>
constexpr auto duo_occ_ptr() -> int (*)(int, bool, int) {
    if constexpr (G == 64) return &duo_occupancy<D, C, K>;
    else return nullptr;
}

}  // namespace smoe
#endif
