/*
 * smoe_oracle.c -- plain-C (fp32, scalar) restatement of the reference's per-block SMoE
 * hot path: forward, analytic gradients, TF1 Adam, kernel-list prune, divergence test.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (steered_mixture_of_experts_amd) never links
 * or calls it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors and its arithmetic lives in
 * TensorFlow 1.x (not installable here); this file is written from the reference's source
 * text (citations: file:line under /root/reference) and cross-checked against the numpy
 * restatement (oracle/smoe_oracle.py) and torch.autograd in tests/test_oracle.py.
 *
 * Layouts are those of include/smoe_hip.h: target [B,C,N], parameters in the get_params()
 * layout with a leading block axis, active = uint32 bitmask per block.
 * Build: gcc -O2 -fopenmp -shared -fPIC (oracle/Makefile).  -ffp-contract=off keeps the
 * TF "one rounding per op" semantics.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 3
#define MAXC 3
#define MAXK 16

typedef struct oracle_cfg {
    int32_t dim, channels, kernels, pixels;
    int32_t precision;
    float margin;
    int32_t use_determinant, use_yuv;
    int32_t train_pis, train_gammas, train_musx;
    float lr_expert, lr_pis, lr_steer;
    float beta1, beta2, adam_eps;
    float grad_clip;
    float pis_l1, u_l1;
    int32_t start_pis;
    int32_t only_y_gamma;       /* smoe.py:725-729 */
    int32_t quantize_pis;       /* smoe.py:474-478: pis go through fake_quant_with_min_max_args(lb, ub, bits) */
    int32_t pis_bits;
    float pis_lb, pis_ub;
} oracle_cfg;

typedef struct grads_t {
    float pis[MAXK], mu[MAXK][MAXD], A[MAXK][MAXD][MAXD], ga[MAXK][MAXD][MAXC], nu[MAXK][MAXC];
} grads_t;

/* One pass over one block (smoe.py:732-937,1012-1053).  Returns loss; fills sse, the new
 * influence mask, optionally recon [C][N] and the analytic gradients. */
static float block_pass(const oracle_cfg* c, const float* coords /*[D][N]*/, const float* tgt /*[C][N]*/,
                        const float* lw /*[N] or NULL*/, const float* pis_var, const float* mu, const float* Ad,
                        const float* Ac, const float* ga, const float* nu, uint32_t active,
                        float* sse_out, uint32_t* active_new, float* recon, grads_t* g) {
    const int D = c->dim, C = c->channels, K = c->kernels, N = c->pixels;
    const float two_p = ldexpf(1.0f, c->precision);
    const float tau = (float)(0.5 * 1.0 / (double)two_p);               /* smoe.py:825 */
    const float epsm = (float)((double)c->margin * 1.0 / (double)two_p); /* smoe.py:931 */
    const float levels = two_p - 1.0f;
    const float scale = 1.0f / levels, inv_scale = 1.0f / scale, nudged_max = levels * scale;
    const float n_dis = (float)sqrt(pow(2.0 * M_PI, (double)D));       /* smoe.py:812 */
    const int y_only = c->only_y_gamma && c->use_yuv && c->train_gammas; /* smoe.py:725 */
    float A[MAXK][MAXD][MAXD], coef[MAXK];
    int act[MAXK];
    /* quantize_pis: TF Nudge() + fake quant in fp32; the gradient passes inside the nudged range only */
    float qpis[MAXK];
    int pis_ste[MAXK];
    for (int k = 0; k < K; ++k) { qpis[k] = pis_var[k]; pis_ste[k] = 1; }
    if (c->quantize_pis) {
        const float qmax = (float)((1u << c->pis_bits) - 1u);
        const float sc = (c->pis_ub - c->pis_lb) / qmax;
        const float zp = 0.0f - c->pis_lb / sc;
        const float nzp = (zp < 0.0f) ? 0.0f : ((zp > qmax) ? qmax : roundf(zp));
        const float nmin = (0.0f - nzp) * sc, nmax = (qmax - nzp) * sc, inv = 1.0f / sc;
        for (int k = 0; k < K; ++k) {
            const float x = pis_var[k];
            const float cl = fminf(fmaxf(x, nmin), nmax);
            qpis[k] = floorf((cl - nmin) * inv + 0.5f) * sc + nmin;
            pis_ste[k] = (x >= nmin) && (x <= nmax);
        }
    }
    const float* pis = qpis;
    float cw[MAXC];
    for (int ch = 0; ch < C; ++ch)
        cw[ch] = c->use_yuv ? (float)(((ch == 0) ? 6.0 / 8.0 : 1.0 / 8.0) / (double)N) : (float)(1.0 / ((double)N * C));
    for (int k = 0; k < K; ++k) {
        act[k] = ((active >> k) & 1u) && (pis[k] > 0.0f);              /* smoe.py:480,738 */
        float det = 1.0f;
        for (int l = 0; l < D; ++l)
            for (int m = 0; m < D; ++m) {                               /* smoe.py:732-733 */
                A[k][l][m] = (l == m) ? Ad[(k * D + l) * D + m] : ((l > m) ? Ac[(k * D + l) * D + m] : 0.0f);
                if (l == m) det *= A[k][l][m];
            }
        const float nq = c->use_determinant ? det / n_dis : 1.0f;      /* smoe.py:809-815 */
        coef[k] = act[k] ? nq * pis[k] : 0.0f;                         /* smoe.py:819 */
    }
    if (g) memset(g, 0, sizeof *g);
    float su[MAXK];
    for (int k = 0; k < K; ++k) su[k] = 0.0f;
    float loss_acc = 0.0f, sse = 0.0f;
    uint32_t infl = 0u;
    for (int n = 0; n < N; ++n) {
        float x[MAXD], r[MAXK][MAXD], z[MAXK][MAXD], gk[MAXK], w[MAXK], wt[MAXK], e[MAXK][MAXC];
        for (int l = 0; l < D; ++l) x[l] = coords[l * N + n];
        float S = 0.0f;
        for (int k = 0; k < K; ++k) {                                  /* smoe.py:777-782,796,807 */
            float maha = 0.0f;
            for (int l = 0; l < D; ++l) r[k][l] = x[l] - mu[k * D + l];
            for (int m = 0; m < D; ++m) {
                float zz = 0.0f;
                for (int l = m; l < D; ++l) zz += r[k][l] * A[k][l][m];
                z[k][m] = zz;
                maha += zz * zz;
            }
            gk[k] = coef[k] * expf(-0.5f * maha);
            S += gk[k];
        }
        const float Sm = fmaxf(10e-12f, S);                            /* smoe.py:821 */
        const int passS = S > 10e-12f;
        float y[MAXC];
        for (int ch = 0; ch < C; ++ch) y[ch] = 0.0f;
        for (int k = 0; k < K; ++k) {
            w[k] = gk[k] / Sm;                                         /* smoe.py:823 */
            const int M = w[k] > tau;                                  /* smoe.py:826 */
            wt[k] = M ? w[k] : 0.0f;
            if (M) infl |= 1u << k;                                    /* smoe.py:829 */
            for (int ch = 0; ch < C; ++ch) {                           /* smoe.py:840-848 */
                float ee = nu[k * C + ch];
                if (c->train_gammas && !(y_only && ch > 0))
                    for (int l = 0; l < D; ++l) ee += ga[(k * D + l) * C + ch] * x[l];
                e[k][ch] = ee;
                y[ch] += wt[k] * ee;
            }
        }
        const float lwn = lw ? lw[n] : 1.0f;
        float G[MAXC];
        for (int ch = 0; ch < C; ++ch) {                               /* smoe.py:857,899,905-937 */
            const float yc = fminf(fmaxf(y[ch], 0.0f), 1.0f);
            const float cl = fminf(yc, nudged_max);
            const float q = floorf(cl * inv_scale + 0.5f) * scale;
            if (recon) recon[ch * N + n] = q;
            const float diff = q - tgt[ch * N + n];
            const float ad = fabsf(diff) - epsm;
            sse += diff * diff;
            loss_acc += cw[ch] * lwn * (ad * ad);
            const int inside = (y[ch] >= 0.0f) && (y[ch] <= 1.0f) && (y[ch] <= nudged_max);
            const float sg = (diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f);
            G[ch] = inside ? (2.0f * cw[ch] * lwn) * ad * sg : 0.0f;
        }
        if (!g) continue;
        float h[MAXK], dot = 0.0f;
        for (int k = 0; k < K; ++k) {
            float hh = 0.0f;
            for (int ch = 0; ch < C; ++ch) hh += e[k][ch] * G[ch];
            h[k] = (wt[k] > 0.0f) ? hh : 0.0f;
            dot += h[k] * w[k];
        }
        for (int k = 0; k < K; ++k) {
            const float u = passS ? w[k] * (h[k] - dot) : w[k] * h[k];
            su[k] += u;
            for (int m = 0; m < D; ++m) {
                const float uz = u * z[k][m];
                for (int l = m; l < D; ++l) g->A[k][l][m] -= r[k][l] * uz;        /* dm/dA[l][m] = 2 r_l z_m */
                for (int l = m; l < D; ++l) g->mu[k][l] += A[k][l][m] * uz;       /* dm/dmu = -2 A z */
            }
            for (int ch = 0; ch < C; ++ch) {
                const float wg = wt[k] * G[ch];
                g->nu[k][ch] += wg;
                if (c->train_gammas && !(y_only && ch > 0))
                    for (int l = 0; l < D; ++l) g->ga[k][l][ch] += wg * x[l];
            }
        }
    }
    float loss = loss_acc;
    const float k0 = (float)(c->start_pis > 0 ? c->start_pis : K);
    for (int k = 0; k < K; ++k) {
        if (!act[k]) continue;
        if (c->pis_l1 != 0.0f) loss += c->pis_l1 / k0 * pis[k];        /* smoe.py:1027 */
        if (c->u_l1 != 0.0f)
            for (int l = 0; l < D; ++l) loss += c->u_l1 * A[k][l][l];  /* smoe.py:1044 */
        if (g) {
            g->pis[k] = pis_ste[k] ? su[k] / pis[k] + c->pis_l1 / k0 : 0.0f;
            for (int l = 0; l < D; ++l) {
                if (c->use_determinant) g->A[k][l][l] += su[k] / A[k][l][l];
                g->A[k][l][l] += c->u_l1;
            }
        }
    }
    *sse_out = sse;
    *active_new = infl;
    return loss;
}

static void adam_one(float* var, float* m, float* v, float g, float lr, const oracle_cfg* c, float b1p, float b2p) {
    if (c->grad_clip > 0.0f) g = fminf(fmaxf(g, -c->grad_clip), c->grad_clip);     /* smoe.py:1152-1153 */
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);                     /* TF1 ApplyAdam */
    *m = *m + (g - *m) * (1.0f - c->beta1);
    *v = *v + (g * g - *v) * (1.0f - c->beta2);
    *var = *var - (*m * alpha) / (sqrtf(*v) + c->adam_eps);
}

/* Evaluation pass over B blocks (run_batched(train=False), smoe.py:1606-1793). */
int smoe_oracle_forward(const oracle_cfg* c, int B, const float* coords, const float* target, const float* loss_w,
                        float* const p[6], float* recon, float* loss, float* sse, uint32_t* active,
                        int update_active, int threads) {
    const int D = c->dim, C = c->channels, K = c->kernels, N = c->pixels;
    if (D > MAXD || C > MAXC || K > MAXK) return -1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int b = 0; b < B; ++b) {
        float s;
        uint32_t an;
        const float l = block_pass(c, coords, target + (size_t)b * C * N, loss_w ? loss_w + (size_t)b * N : NULL,
                                   p[0] + (size_t)b * K, p[1] + (size_t)b * K * D, p[2] + (size_t)b * K * D * D,
                                   p[3] + (size_t)b * K * D * D, p[4] + (size_t)b * K * D * C, p[5] + (size_t)b * K * C,
                                   active[b], &s, &an, recon ? recon + (size_t)b * C * N : NULL, NULL);
        if (loss) loss[b] = l;
        if (sse) sse[b] = s;
        if (update_active) active[b] = an;
    }
    return 0;
}

/* n_iters training iterations over B blocks (loop body of Smoe.train, smoe.py:1521-1570):
 * pass -> prune -> TF1 Adam -> divergence test, per block.  p/m/v: the six tensors
 * (pis, musX, A_diagonal, A_corr, gamma_e, nu_e).  beta_pow[2] in/out. */
int smoe_oracle_fit(const oracle_cfg* c, int B, const float* coords, const float* target, const float* loss_w,
                    float* const p[6], float* const m[6], float* const v[6], int n_iters, float* beta_pow,
                    float* loss_last, float* sse_last, uint32_t* active, uint32_t* diverged, const float* loss0,
                    int threads) {
    const int D = c->dim, C = c->channels, K = c->kernels, N = c->pixels;
    if (D > MAXD || C > MAXC || K > MAXK) return -1;
    const float b1p0 = beta_pow[0], b2p0 = beta_pow[1];
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int b = 0; b < B; ++b) {
        float* pis = p[0] + (size_t)b * K;
        float* mu = p[1] + (size_t)b * K * D;
        float* Ad = p[2] + (size_t)b * K * D * D;
        float* Ac = p[3] + (size_t)b * K * D * D;
        float* ga = p[4] + (size_t)b * K * D * C;
        float* nu = p[5] + (size_t)b * K * C;
        const size_t o0 = (size_t)b * K, o1 = (size_t)b * K * D, o2 = (size_t)b * K * D * D, o4 = (size_t)b * K * D * C,
                     o5 = (size_t)b * K * C;
        float b1p = b1p0, b2p = b2p0;
        int frozen = diverged ? (diverged[b] != 0u) : 0;
        grads_t g;
        for (int it = 0; it < n_iters; ++it) {
            if (!frozen) {
                float s;
                uint32_t an;
                const float l = block_pass(c, coords, target + (size_t)b * C * N, loss_w ? loss_w + (size_t)b * N : NULL,
                                           pis, mu, Ad, Ac, ga, nu, active[b], &s, &an, NULL, &g);
                active[b] = an;                                                      /* smoe.py:1763-1766 */
                if (loss_last) loss_last[b] = l;
                if (sse_last) sse_last[b] = s;
                for (int k = 0; k < K; ++k) {                                        /* smoe.py:1102-1193 */
                    if (c->train_pis && c->lr_pis != 0.0f)
                        adam_one(&pis[k], &m[0][o0 + k], &v[0][o0 + k], g.pis[k], c->lr_pis, c, b1p, b2p);
                    for (int l2 = 0; l2 < D; ++l2) {
                        if (c->train_musx && c->lr_expert != 0.0f)
                            adam_one(&mu[k * D + l2], &m[1][o1 + k * D + l2], &v[1][o1 + k * D + l2], g.mu[k][l2], c->lr_expert, c, b1p, b2p);
                        if (c->lr_steer != 0.0f) {
                            const int dd = (k * D + l2) * D + l2;
                            adam_one(&Ad[dd], &m[2][o2 + dd], &v[2][o2 + dd], g.A[k][l2][l2], c->lr_steer, c, b1p, b2p);
                            for (int m2 = 0; m2 < l2; ++m2) {
                                const int cc = (k * D + l2) * D + m2;
                                adam_one(&Ac[cc], &m[3][o2 + cc], &v[3][o2 + cc], g.A[k][l2][m2], c->lr_steer, c, b1p, b2p);
                            }
                        }
                        if (c->train_gammas && c->lr_expert != 0.0f)
                            for (int ch = 0; ch < ((c->only_y_gamma && c->use_yuv) ? 1 : C); ++ch) {
                                const int gg = (k * D + l2) * C + ch;
                                adam_one(&ga[gg], &m[4][o4 + gg], &v[4][o4 + gg], g.ga[k][l2][ch], c->lr_expert, c, b1p, b2p);
                            }
                    }
                    if (c->lr_expert != 0.0f)
                        for (int ch = 0; ch < C; ++ch)
                            adam_one(&nu[k * C + ch], &m[5][o5 + k * C + ch], &v[5][o5 + k * C + ch], g.nu[k][ch], c->lr_expert, c, b1p, b2p);
                }
                /* smoe.py:1565-1570, per block; takes effect from the next iteration */
                if (l != l || (loss0 && (l + 1.0f > (loss0[b] + 100.0f) * 10.0f))) frozen = 1;
            }
            b1p *= c->beta1;
            b2p *= c->beta2;
        }
        if (diverged) diverged[b] = frozen ? 1u : 0u;
    }
    for (int it = 0; it < n_iters; ++it) {
        beta_pow[0] *= c->beta1;
        beta_pow[1] *= c->beta2;
    }
    return 0;
}
