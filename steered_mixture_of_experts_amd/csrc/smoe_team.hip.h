// smoe_team.hip.h -- the fit kernel for batches that leave the chip under-filled ("team" tiling).
//
// The regular kernels (smoe_block.hip.h) give a block to 16 / 32 / 64 lanes of ONE wavefront, which also runs everything an
// iteration does outside the pixel loop (load + derive the block constants, cross-lane reduction, post-transform, Adam).
// That work is cheapest per block on the 16-lane tiling (four blocks share every instruction: ~100 wavefront-instructions
// per block-iteration against ~650 on the wavefront-per-block tiling), but 16 lanes per block need >= 16 000 blocks to put
// four wavefronts on every SIMD: ONE 512x512 image (1 024 blocks) is 256 wavefronts for 1 024 SIMDs, and a lone wavefront
// issues one instruction per ~5.6 cycles (profiles/r02/ubench_valu.txt).
//
// Here a workgroup of W wavefronts (W = 2, 4, 8) takes FOUR blocks on the 16-lane layout (lane = (block grp, column sub)),
// and the wavefronts split the PIXEL ROWS of those blocks: wavefront w sweeps the steps i = w, w + W, ... (pixel n =
// i * 16 + sub).  Per iteration:
//   P0  every wavefront loads the DERIVED constants of its lanes' blocks (A' = SQ A, c = A'^T mu, coef, nu, gamma: published
//       by the slot owners, one ds_read_b128 per four values) and hoists the lane-constant coordinate;
//   P1  its share of the pixel steps (the same `pixel<>` routine, scalar influence votes), accumulators in registers;
//   P2  the 16-lane transpose reduction inside the wavefront (reduce_slots_lds); the per-wavefront partial totals go to
//       the wavefront's own scratch;                                                         -- barrier --
//   P3  slot owners: round q of the slot set belongs to wavefront q mod W; lane (grp, sub) of it owns slot 16 q + sub of
//       block grp, sums the W partial totals in wavefront order and publishes the raw total;   -- barrier --
//   P4  owners turn raw totals into gradients (eval_slot_desc, as on the wavefront-per-block tiling), add the l1 terms,
//       clip, apply TF1 Adam, and write the new parameter, the kernel list, the divergence flag and the derived constants
//       that depend on their own slot only into the OTHER parameter buffer (double-buffered: lanes of other wavefronts
//       are still reading the old values);                                                    -- barrier --
//   P5  the owners of centres and priors derive c = A'^T mu and coef from the new values of the neighbouring slots.
//                                                                                              -- barrier --
// Every wavefront does the same amount of pixel work; the owner phases are one slot per lane with all 64 lanes busy.
// Graph coverage: the margin-loss graph with quantization_mode 0 / 1 (quantize_pis included), loss weights, l1 terms,
// clipping, trainable flags, only_y_gamma, kernel_count_as_norm_l1; not radial_as / train_inverse_cov / ssim_opt /
// quantization_mode 2, 3 (those run the regular kernels).  Block shapes whose last axis divides 16 (HL >= 1).
#ifndef SMOE_TEAM_HIP_H
#define SMOE_TEAM_HIP_H

#include "smoe_block.hip.h"

#define SMOE_TEAM_MAXW 8      // wavefronts per workgroup: 2, 4 or 8 (launch bound of the kernel)

namespace smoe {

template <int D, int C, int K>
struct TeamTile {
    using Lt = Layout<D, C, K>;
    static constexpr int NB = 4;                          // blocks per workgroup (64 lanes / 16 lanes per block)
    static constexpr int MAXW = SMOE_TEAM_MAXW;
    static constexpr int ROW = 64 + 4;
    static constexpr int RPR = 16;                        // rows per reduction round on the 16-lane layout
    static constexpr int SPL = (Lt::NSLOT + RPR - 1) / RPR;   // rounds = partial totals a lane holds after the reduction
    static constexpr int SCR = RPR * ROW;                 // transpose scratch of one wavefront; afterwards its partial totals
    static_assert(SPL * 64 <= SCR, "a wavefront's partial totals are handed over through its reduction scratch");
    static constexpr int DER_STRIDE = round_up(Lt::NPAR, 4);
    static constexpr int TOT_STRIDE = round_up(Lt::NSLOT, 4);
    static constexpr int MV_STRIDE = round_up(2 * Lt::NPAR, 4);
    static constexpr int DESC_DW = 12;
    static constexpr int PARBUF = NB * Lt::LP_STRIDE;     // one parameter buffer (all four blocks)
    __host__ __device__ static int off_par(int N, int CR) { return round_up(CR * N, 4); }
    __host__ __device__ static int off_mv(int N, int CR) { return off_par(N, CR) + 2 * PARBUF; }
    __host__ __device__ static int off_der(int N, int CR) { return off_mv(N, CR) + NB * MV_STRIDE; }
    __host__ __device__ static int off_tot(int N, int CR) { return off_der(N, CR) + NB * DER_STRIDE; }      // scaled totals
    __host__ __device__ static int off_raw(int N, int CR) { return off_tot(N, CR) + NB * TOT_STRIDE; }      // raw totals
    __host__ __device__ static int off_desc(int N, int CR) { return off_raw(N, CR) + NB * TOT_STRIDE; }
    __host__ __device__ static int off_scr(int N, int CR) { return off_desc(N, CR) + NB * Lt::NPAR * DESC_DW; }
    __host__ __device__ static int off_tgt(int N, int CR, int nw) { return off_scr(N, CR) + nw * SCR; }
    __host__ __device__ static int off_lw(int N, int CR, int nw) { return off_tgt(N, CR, nw) + NB * C * N; }
    __host__ __device__ static size_t bytes(int N, bool has_lw, int CR, int nw) {
        return sizeof(float) * (size_t)(off_lw(N, CR, nw) + (has_lw ? NB * N : 0));
    }
};

// tensor of packed offset o inside a kernel's record: 0 pis 1 musX 2 A_diagonal 3 A_corr 4 gamma_e 5 nu_e; chan = channel
// of a gamma_e entry
template <int D, int C, int K>
__device__ __forceinline__ int team_slot_tensor(int o, int& chan) {
    using Lt = Layout<D, C, K>;
    const int t = o - Lt::O_A;
    const int l = (t >= 3) ? 2 : ((t >= 1) ? 1 : 0);
    const int m = t - l * (l + 1) / 2;
    chan = (o >= Lt::O_GA) ? (o - Lt::O_GA) % C : 0;
    return (o == Lt::O_PI) ? 0 : ((o < Lt::O_A) ? 1 : ((o < Lt::O_NU) ? ((l == m) ? 2 : 3) : ((o < Lt::O_GA) ? 5 : 4)));
}

// eval_slot_desc with the parameter operands taken from the CURRENT parameter buffer (pb = its byte offset from buffer 0)
__device__ __forceinline__ float team_eval_desc(const float* __restrict__ lds, const SlotDesc& d, float tot, uint32_t pb) {
    const float t1 = lds_f32(lds, d.T1), t2 = lds_f32(lds, d.T2), t3 = lds_f32(lds, d.T3), tr = lds_f32(lds, d.TR);
    const float p1 = lds_f32(lds, d.P1 + pb), p2 = lds_f32(lds, d.P2 + pb), p3 = lds_f32(lds, d.P3 + pb), pr = lds_f32(lds, d.PR + pb);
    const float flag = lds_f32(lds, d.FLAG + pb), piv = lds_f32(lds, d.PIV + pb);      // both loaded before the test: no dependent read
    const bool act = (flag != 0.0f) & (piv > 0.0f);                                    // smoe.py:480,738
    float g = d.c_self * tot;
    g = fmaf(p1, t1, g);
    g = fmaf(p2, t2, g);
    g = fmaf(p3, t3, g);
    return fmaf(act ? fast_rcp(pr) : 0.0f, tr, g);
}

template <int D, int C, int K, int HL, int TW>
__global__ void __launch_bounds__(SMOE_TEAM_MAXW * 64) fit_team_kernel(FitArgs a) {
    using Lt = Layout<D, C, K>;
    using TT = TeamTile<D, C, K>;
    static_assert(HL >= 1, "team tiling: the last block axis divides the 16 lanes of a block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N, B = a.B;
    const int nw = (int)(blockDim.x >> 6);
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4;
    const int sub = lane & 15;
    const int blk0 = blockIdx.x * TT::NB;
    const int b_raw = blk0 + grp;
    const bool valid_b = b_raw < B;
    const int b = valid_b ? b_raw : B - 1;
    constexpr int CR = D - HL;

    float* s_coords = lds;
    float* s_par0 = lds + TT::off_par(N, CR) + grp * Lt::LP_STRIDE;        // buffer 0 of this lane's block; buffer 1 = + PARBUF
    float* s_mv = lds + TT::off_mv(N, CR) + grp * TT::MV_STRIDE;
    float* s_der = lds + TT::off_der(N, CR) + grp * TT::DER_STRIDE;
    float* s_tot = lds + TT::off_tot(N, CR) + grp * TT::TOT_STRIDE;
    float* s_raw = lds + TT::off_raw(N, CR) + grp * TT::TOT_STRIDE;
    float* s_desc = lds + TT::off_desc(N, CR) + grp * (Lt::NPAR * TT::DESC_DW);
    float* s_scr_all = lds + TT::off_scr(N, CR);
    float* s_scr = s_scr_all + wave * TT::SCR;
    const float* s_tgt = lds + TT::off_tgt(N, CR, nw) + grp * (C * N);
    const float* s_lw = lds + TT::off_lw(N, CR, nw) + grp * N;
    const bool has_lw = a.loss_w != nullptr;

    SMOE_LDS_CHECK(TT::off_lw(N, CR, nw) + (has_lw ? TT::NB * N : 0), 8u);
    // ---- staging: coordinates, targets and loss weights of the four blocks (one contiguous run each) --------------------
    {
        const int nt = (int)blockDim.x;
        for (int i = threadIdx.x; i < CR * N; i += nt) s_coords[i] = a.coords[i];
        auto copy_planes = [&](const float* __restrict__ src, float* __restrict__ dst, int per) {
            const size_t g0 = (size_t)blk0 * per, gend = (size_t)B * per;
            if ((per & 3) == 0) {
                const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
                const size_t last4 = (gend >> 2) - 1;
                for (int i = threadIdx.x; i < TT::NB * (per >> 2); i += nt)
                    reinterpret_cast<float4*>(dst)[i] = src4[min((g0 >> 2) + (size_t)i, last4)];
            } else {
                for (int i = threadIdx.x; i < TT::NB * per; i += nt) dst[i] = src[min(g0 + (size_t)i, gend - 1)];
            }
        };
        copy_planes(a.target, lds + TT::off_tgt(N, CR, nw), C * N);
        if (has_lw) copy_planes(a.loss_w, lds + TT::off_lw(N, CR, nw), N);
    }
    float xc[D];                                   // coordinates of the lane's pixel i = 0 (hoisted axis: of all its pixels)
#pragma unroll
    for (int l = 0; l < D; ++l) xc[l] = a.coords[l * N + min(sub, N - 1)];

    const KernelConsts kc = a.kc;
    const bool patch_pis = kc.qpis != 0;                     // quantize_pis: the graph reads the fake-quantised priors
    const bool has_reg = (a.reg_pi != 0.0f) || (a.reg_u != 0.0f);
    const bool has_loss0 = a.loss0 != nullptr;
    const float beta1 = a.beta1, beta2 = a.beta2, adam_eps = a.eps, clip = a.clip;
    const float reg_pi = a.reg_pi, reg_u = a.reg_u;
    const uint32_t par_bytes = (uint32_t)((s_par0 - lds) * sizeof(float));
    constexpr uint32_t PB1 = (uint32_t)(TT::PARBUF * sizeof(float));      // byte distance of parameter buffer 1

    auto quant_pi = [&](float x) {                           // fake_quant_with_min_max_args of a prior (smoe.py:474-480)
        const float cl = fminf(fmaxf(x, kc.q_nmin[3]), kc.q_nmax[3]);
        return floorf((cl - kc.q_nmin[3]) * kc.q_inv[3] + 0.5f) * kc.q_scale[3] + kc.q_nmin[3];
    };
    // what slot j's own value contributes to the derived constants: A' = SQ A; nu, gamma (masked as derive() does)
    auto derived_own = [&](int o, float v) {
        int chan;
        const int tensor = team_slot_tensor<D, C, K>(o, chan);
        float d = v;
        if (tensor == 2 || tensor == 3) d = SMOE_SQ * v;
        if (tensor == 4 && (!kc.train_gammas || (kc.only_y_gamma && chan != 0))) d = 0.0f;    // smoe.py:841-848, 725-729
        return d;
    };
    // P5: c_m = sum_{l >= m} mu_l A'_lm ; coef = act ? pi * prod diag(A) / sqrt((2 pi)^d) : 0  (same operation order as
    // BlockRegs::derive, so that the forward is bit-identical to the regular kernels')
    auto derive_cross = [&](int j, const float* __restrict__ par) {
        const int k = j / Lt::PK, o = j - k * Lt::PK;
        const float* p = par + k * Lt::PK;
        if (o == Lt::O_PI) {
            float det = 1.0f;
#pragma unroll
            for (int l = 0; l < D; ++l) det *= p[Lt::O_A + tri_index(l, l)];
            const float nq = kc.use_det ? det * kc.inv_n_dis : 1.0f;
            const float piv = patch_pis ? par[Lt::LP_QPI + k] : p[Lt::O_PI];
            const bool act = (par[Lt::LP_ACT + k] != 0.0f) && (piv > 0.0f);
            s_der[j] = act ? nq * piv : 0.0f;
        } else if (o < Lt::O_A) {
            const int m = o - Lt::O_MU;
            float c = 0.0f;
#pragma unroll
            for (int l = 0; l < D; ++l)
                if (l >= m) c = fmaf(p[Lt::O_MU + l], SMOE_SQ * p[Lt::O_A + l * (l + 1) / 2 + m], c);
            s_der[j] = c;
        }
    };

    // ---- owner set-up: round q of the slot set belongs to wavefront q mod nw; lane (grp, sub) owns slot 16 q + sub ------
    for (int q = wave; q < TT::SPL; q += nw) {
        const int j = q * 16 + sub;
        if (j < Lt::NPAR) {
            int tensor, kern; long off;
            decode_slot<D, C, K>(j, b, tensor, off, kern);
            const float v = pick(a.p, tensor)[off];
            s_par0[j] = v;
            s_mv[2 * j] = pick(a.m, tensor)[off];
            s_mv[2 * j + 1] = pick(a.v, tensor)[off];
            const int o = j - kern * Lt::PK;
            if (tensor != 0 && tensor != 1) s_der[j] = derived_own(o, v);
            if (tensor == 0) s_par0[Lt::LP_QPI + kern] = patch_pis ? quant_pi(v) : v;
            const SlotDesc d = build_slot_desc<D, C, K, false>(j, (uint32_t)((s_tot - lds) * sizeof(float)), par_bytes, par_bytes,
                                                               patch_pis, kc.use_det != 0);
            float4* od = reinterpret_cast<float4*>(s_desc + j * TT::DESC_DW);
            od[0] = make_float4(d.c_self, d.pub_scale, __uint_as_float(d.T1), __uint_as_float(d.T2));
            od[1] = make_float4(__uint_as_float(d.T3), __uint_as_float(d.TR), __uint_as_float(d.P1), __uint_as_float(d.P2));
            od[2] = make_float4(__uint_as_float(d.P3), __uint_as_float(d.PR), __uint_as_float(d.FLAG), __uint_as_float(d.PIV));
        } else if (j >= Lt::S_CNT && j < Lt::S_CNT + K) {
            s_par0[Lt::LP_ACT + (j - Lt::S_CNT)] = ((a.active[b] >> (j - Lt::S_CNT)) & 1u) ? 1.0f : 0.0f;
        } else if (j == Lt::S_LOSS) {
            s_par0[Lt::LP_FROZEN] = (a.diverged != nullptr && a.diverged[b] != 0u) ? 1.0f : 0.0f;
            s_par0[Lt::LP_ZERO] = 0.0f; s_par0[Lt::LP_ONE] = 1.0f;                              // constant cells of both buffers
            s_par0[TT::PARBUF + Lt::LP_ZERO] = 0.0f; s_par0[TT::PARBUF + Lt::LP_ONE] = 1.0f;
        }
    }
    const float loss0 = has_loss0 ? a.loss0[b] : 0.0f;
    __syncthreads();
    for (int q = wave; q < TT::SPL; q += nw) {
        const int j = q * 16 + sub;
        if (j < Lt::NPAR) derive_cross(j, s_par0);
    }
    __syncthreads();

    float b1p = a.b1p, b2p = a.b2p;
    float last_loss = 0.0f, last_sse = 0.0f;
    int cur = 0;                                             // parameter buffer the graph of this iteration is built on

    for (int it = 0; it < a.n_iters; ++it) {
        // ---- P0 / P1: the wavefront's share of the pixel steps of its lanes' blocks ---------------------------------------
        float acc[Lt::NSLOT];
#pragma unroll
        for (int j = 0; j < Lt::NSLOT; ++j) acc[j] = 0.0f;
        {
            BlockRegs<D, C, K> R;
            float dv[TT::DER_STRIDE];
            const float4* src = reinterpret_cast<const float4*>(s_der);
#pragma unroll
            for (int i = 0; i < TT::DER_STRIDE / 4; ++i) {
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
            hoist_const<D, C, K, HL, false>(R, xc);
            // pixel n = i * (16 nw) + wave * 16 + sub: the wavefronts take the 16-pixel steps of a block round robin
            if (has_lw) pixel_loop_train<D, C, K, true, HL, false>(R, kc, s_coords, s_tgt, s_lw, N, 16 * nw, wave * 16 + sub, acc, a.lw_is_sample != 0);
            else pixel_loop_train<D, C, K, false, HL, false>(R, kc, s_coords, s_tgt, s_lw, N, 16 * nw, wave * 16 + sub, acc);
            complete_const<D, C, K, HL, false>(R, xc, acc);
        }
        // ---- P2: 16-lane transpose reduction inside the wavefront; partial totals stay in its scratch ---------------------
        {
            float total[TT::SPL];
            reduce_slots_lds<D, C, K, 16, TW, 0>(acc, s_scr, lane, total);
#pragma unroll
            for (int q = 0; q < TT::SPL; ++q) s_scr[q * 64 + lane] = total[q];
        }
        __syncthreads();
        // ---- P3: raw totals of the owned slots = sum of the wavefronts' partial totals, in wavefront order ----------------
        for (int q = wave; q < TT::SPL; q += nw) {
            const int j = q * 16 + sub;
            float T = 0.0f;
#pragma unroll
            for (int w = 0; w < TT::MAXW; ++w) {
                const float v = s_scr_all[min(w, nw - 1) * TT::SCR + q * 64 + lane];
                T += (w < nw) ? v : 0.0f;
            }
            if (j < Lt::NSLOT) {
                s_raw[j] = T;
                if (j < Lt::NPAR) {
                    const int o = j % Lt::PK;
                    s_tot[j] = (o >= Lt::O_MU && o < Lt::O_A) ? T * SMOE_INV_SQ : T;     // suz = suz' / SQ (eval_slot_desc)
                }
            }
        }
        __syncthreads();
        // ---- P4: gradient, l1 terms, clip, TF1 ApplyAdam (smoe.py:1173-1193), prune (1763-1766), stop test (1565-1570) -----
        const float bias = __builtin_amdgcn_sqrtf(1.0f - b2p) * fast_rcp(1.0f - b1p);
        const float* par = s_par0 + cur * TT::PARBUF;
        float* parn = s_par0 + (cur ^ 1) * TT::PARBUF;
        const uint32_t pb = cur ? PB1 : 0u;
        for (int q = wave; q < TT::SPL; q += nw) {
            const int j = q * 16 + sub;
            if (j >= Lt::NSLOT) continue;
            const float T = s_raw[j];
            const bool frozen = par[Lt::LP_FROZEN] != 0.0f;
            if (j < Lt::NPAR) {
                const int k = j / Lt::PK, o = j - k * Lt::PK;
                int chan;
                const int tensor = team_slot_tensor<D, C, K>(o, chan);
                const float4* dq = reinterpret_cast<const float4*>(s_desc + j * TT::DESC_DW);
                const float4 q0 = dq[0], q1 = dq[1], q2 = dq[2];
                SlotDesc d;
                d.c_self = q0.x; d.pub_scale = q0.y; d.T1 = __float_as_uint(q0.z); d.T2 = __float_as_uint(q0.w);
                d.T3 = __float_as_uint(q1.x); d.TR = __float_as_uint(q1.y); d.P1 = __float_as_uint(q1.z); d.P2 = __float_as_uint(q1.w);
                d.P3 = __float_as_uint(q2.x); d.PR = __float_as_uint(q2.y); d.FLAG = __float_as_uint(q2.z); d.PIV = __float_as_uint(q2.w);
                float gsum = team_eval_desc(lds, d, T, pb);
                // optimizer groups, smoe.py:1102-1104; untrainable variables dropped, 1112-1117
                float lr = (tensor == 0) ? a.lr_pis : ((tensor == 2 || tensor == 3) ? a.lr_steer : a.lr_expert);
                if (tensor == 0 && !a.train_pis) lr = 0.0f;
                if (tensor == 1 && !a.train_musx) lr = 0.0f;
                if (tensor == 4 && (!kc.train_gammas || (kc.only_y_gamma && chan != 0))) lr = 0.0f;
                if (has_reg && (tensor == 0 || tensor == 2)) {                      // smoe.py:1027,1044 (active kernels only)
                    const float piv = patch_pis ? par[Lt::LP_QPI + k] : par[k * Lt::PK + Lt::O_PI];
                    const bool act = (par[Lt::LP_ACT + k] != 0.0f) && (piv > 0.0f);
                    float rs = (tensor == 0) ? reg_pi : reg_u;
                    if (kc.kcount_norm && tensor == 0) {                             // pis_l1 / count(qpis > 0), smoe.py:1022-1027
                        float cnt = 0.0f;
#pragma unroll
                        for (int kk = 0; kk < K; ++kk) cnt += ((patch_pis ? par[Lt::LP_QPI + kk] : par[kk * Lt::PK + Lt::O_PI]) > 0.0f) ? 1.0f : 0.0f;
                        rs = kc.pis_l1_raw / fmaxf(cnt, 1.0f);
                    }
                    gsum += act ? rs : 0.0f;
                }
                const float pv = par[j];
                if (patch_pis && tensor == 0) gsum = (pv >= kc.q_nmin[3] && pv <= kc.q_nmax[3]) ? gsum : 0.0f;   // straight-through range
                if (clip > 0.0f) gsum = fminf(fmaxf(gsum, -clip), clip);
                const float mv = s_mv[2 * j], vv = s_mv[2 * j + 1];
                const float alpha = lr * bias;
                const float m2 = mv + (gsum - mv) * (1.0f - beta1);
                const float v2 = vv + (gsum * gsum - vv) * (1.0f - beta2);
                const float p2 = pv - (m2 * alpha) * fast_rcp(__builtin_amdgcn_sqrtf(v2) + adam_eps);
                const bool upd = (lr != 0.0f) && !frozen;
                const float newp = upd ? p2 : pv;
                if (upd) { s_mv[2 * j] = m2; s_mv[2 * j + 1] = v2; }
                parn[j] = newp;
                if (tensor != 0 && tensor != 1) s_der[j] = derived_own(o, newp);
                if (tensor == 0) parn[Lt::LP_QPI + k] = patch_pis ? quant_pi(newp) : newp;
            } else if (j == Lt::S_LOSS) {
                float reg_loss = 0.0f;
                if (has_reg) {
                    float cnt = 0.0f;
#pragma unroll
                    for (int kk = 0; kk < K; ++kk) cnt += ((patch_pis ? par[Lt::LP_QPI + kk] : par[kk * Lt::PK + Lt::O_PI]) > 0.0f) ? 1.0f : 0.0f;
                    const float rp = kc.kcount_norm ? kc.pis_l1_raw / fmaxf(cnt, 1.0f) : reg_pi;
#pragma unroll
                    for (int kk = 0; kk < K; ++kk) {
                        const float piv = patch_pis ? par[Lt::LP_QPI + kk] : par[kk * Lt::PK + Lt::O_PI];
                        if ((par[Lt::LP_ACT + kk] != 0.0f) && (piv > 0.0f)) {
                            reg_loss += rp * piv;
#pragma unroll
                            for (int l = 0; l < D; ++l) reg_loss += reg_u * par[kk * Lt::PK + Lt::O_A + tri_index(l, l)];
                        }
                    }
                }
                bool bad = false;
                if (!frozen) {
                    const float lossv = T + reg_loss;
                    last_loss = lossv;
                    bad = (lossv != lossv) || (has_loss0 && (lossv + 1.0f > (loss0 + 100.0f) * 10.0f));
                }
                parn[Lt::LP_FROZEN] = (frozen || bad) ? 1.0f : 0.0f;        // takes effect from the next iteration
            } else if (j == Lt::S_SSE) {
                if (!frozen) last_sse = T;
            } else {
                const int k = j - Lt::S_CNT;
                parn[Lt::LP_ACT + k] = frozen ? par[Lt::LP_ACT + k] : ((T > 0.0f) ? 1.0f : 0.0f);
            }
        }
        __syncthreads();
        // ---- P5: derived constants that involve the neighbouring slots' new values ------------------------------------------
        for (int q = wave; q < TT::SPL; q += nw) {
            const int j = q * 16 + sub;
            if (j < Lt::NPAR) derive_cross(j, parn);
        }
        __syncthreads();
        cur ^= 1;
        b1p *= beta1;
        b2p *= beta2;
    }

    // ---- write back ----------------------------------------------------------------------------------------------------------
    if (valid_b) {
        const float* par = s_par0 + cur * TT::PARBUF;
        for (int q = wave; q < TT::SPL; q += nw) {
            const int j = q * 16 + sub;
            if (j < Lt::NPAR) {
                int tensor, kern; long off;
                decode_slot<D, C, K>(j, b, tensor, off, kern);
                pick(a.p, tensor)[off] = par[j];
                pick(a.m, tensor)[off] = s_mv[2 * j];
                pick(a.v, tensor)[off] = s_mv[2 * j + 1];
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
}

template <int D, int C, int K>
size_t team_lds_bytes(int N, bool has_lw, int nw) { return TeamTile<D, C, K>::bytes(N, has_lw, D - 1, nw); }

// nw wavefronts per workgroup (2, 4, 8); hoist >= 1 required (the caller checks)
template <int D, int C, int K, int TW>
hipError_t launch_fit_team(const FitArgs& a, int hoist, int nw, hipStream_t st) {
    using TT = TeamTile<D, C, K>;
    if (hoist < 1 || (nw != 2 && nw != 4 && nw != 8)) return hipErrorInvalidValue;
    auto kern = fit_team_kernel<D, C, K, 1, TW>;
    const size_t shm = TT::bytes(a.N, a.loss_w != nullptr, D - 1, nw);
    hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), shm);
    if (e != hipSuccess) return e;
    const int grid = (a.B + TT::NB - 1) / TT::NB;
    FitArgs aa = a;
    aa.lds_floats = (int)(shm / sizeof(float));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), shm, st, aa);
    return hipGetLastError();
}

template <int D, int C, int K, int TW>
int team_occupancy(int N, bool has_lw, int nw) {
    using TT = TeamTile<D, C, K>;
    auto kern = fit_team_kernel<D, C, K, 1, TW>;
    const size_t shm = TT::bytes(N, has_lw, D - 1, nw);
    int nb = 0;
    if (allow_lds(reinterpret_cast<const void*>(kern), shm) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * nw, shm) != hipSuccess) return -1;
    return nb * nw;
}

// table entries of the 16-lane variants (SMOE_VARIANT / SMOE_VARIANT_BASIC in smoe_block.hip.h); other tilings: null
template <int D, int C, int K, int G, int W>
constexpr auto team_fit_ptr() -> hipError_t (*)(const FitArgs&, int, int, hipStream_t) {
    if constexpr (G == 16) return &launch_fit_team<D, C, K, W>;
    else return nullptr;
}
template <int D, int C, int K, int G, int W>
constexpr auto team_lds_ptr() -> size_t (*)(int, bool, int) {
    if constexpr (G == 16) return &team_lds_bytes<D, C, K>;
    else return nullptr;
}
template <int D, int C, int K, int G, int W>
constexpr auto team_occ_ptr() -> int (*)(int, bool, int) {
    if constexpr (G == 16) return &team_occupancy<D, C, K, W>;
    else return nullptr;
}

}  // namespace smoe
#endif
