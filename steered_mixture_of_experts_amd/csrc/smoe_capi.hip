// smoe_capi.hip -- host side of libsmoe_hip.so: handle management, argument checking
// and kernel dispatch behind the C ABI declared in include/smoe_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smoe_device.h"
#include "smoe_hip.h"

struct smoe_context {
    smoe_config cfg;
    int N;
    float* d_coords;     // [D][N]
    float* d_probes;     // [D][3]
    int force_g;
    smoe::KernelConsts kc;
    std::vector<float> h_coords;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return SMOE_ERR_HIP;
}

#define HIP_TRY(expr, what)                              \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return fail_hip(_e, what); \
    } while (0)

// numpy.linspace(0, 1, n) as gen_domain uses it (smoe.py:2412): arange(n) * step, the
// last sample forced to the end point; fed to the graph as float32 (smoe.py:545).
void linspace01(int n, std::vector<float>& out) {
    out.resize(n);
    if (n == 1) { out[0] = 0.0f; return; }
    const double step = 1.0 / (double)(n - 1);
    for (int i = 0; i < n; ++i) out[i] = (float)((double)i * step);
    out[n - 1] = 1.0f;
}

bool params_ok(const smoe_params* p) {
    return p && p->pis && p->musX && p->A_diagonal && p->A_corr && p->gamma_e && p->nu_e;
}

const smoe::Variant* find_variant(const smoe_context* h, int num_blocks, bool has_lw) {
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    // lanes per block: few blocks -> spread each block over a whole wavefront (more
    // waves in flight); many blocks -> 16 lanes per block (4 blocks per wavefront, the
    // cross-lane reduction is amortised over 4x more pixels per lane).
    int want = h->force_g ? h->force_g : ((num_blocks >= 8192) ? 16 : 64);
    const smoe::Variant* fallback = nullptr;
    for (int i = 0; i < n; ++i) {
        if (v[i].D != h->cfg.dim || v[i].C != h->cfg.channels || v[i].K != h->cfg.kernels) continue;
        if (v[i].lds_bytes(h->N, has_lw) > 160u * 1024u) continue;
        if (v[i].G == want) return &v[i];
        if (!fallback) fallback = &v[i];
    }
    return h->force_g ? nullptr : fallback;
}

}  // namespace

extern "C" {

const char* smoe_last_error(void) { return g_err.c_str(); }
int smoe_abi_version(void) { return SMOE_ABI_VERSION; }

int smoe_is_supported(int32_t dim, int32_t channels, int32_t kernels) {
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    for (int i = 0; i < n; ++i)
        if (v[i].D == dim && v[i].C == channels && v[i].K == kernels) return 1;
    return 0;
}

int smoe_create(smoe_handle* out, const smoe_config* cfg) {
    if (!out || !cfg) return fail(SMOE_ERR_INVALID, "smoe_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != SMOE_ABI_VERSION) return fail(SMOE_ERR_INVALID, "smoe_create: abi_version mismatch");
    if (cfg->dim < 2 || cfg->dim > SMOE_MAX_DIM) return fail(SMOE_ERR_INVALID, "smoe_create: dim must be 2 or 3");
    if (cfg->channels < 1 || cfg->channels > SMOE_MAX_CHANNELS) return fail(SMOE_ERR_INVALID, "smoe_create: channels must be 1..3");
    if (cfg->kernels < 1 || cfg->kernels > 16) return fail(SMOE_ERR_INVALID, "smoe_create: kernels must be 1..16");
    if (cfg->precision < 1 || cfg->precision > 16) return fail(SMOE_ERR_INVALID, "smoe_create: precision must be 1..16");
    long N = 1;
    for (int l = 0; l < cfg->dim; ++l) {
        if (cfg->block_shape[l] < 1) return fail(SMOE_ERR_INVALID, "smoe_create: block_shape entries must be >= 1");
        N *= cfg->block_shape[l];
    }
    if (N > 8192) return fail(SMOE_ERR_INVALID, "smoe_create: more than 8192 pixels per block");
    if (!smoe_is_supported(cfg->dim, cfg->channels, cfg->kernels)) {
        char buf[160];
        snprintf(buf, sizeof buf, "smoe_create: no kernel instantiated for (dim=%d, channels=%d, kernels=%d)",
                 cfg->dim, cfg->channels, cfg->kernels);
        return fail(SMOE_ERR_UNSUPPORTED, buf);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SMOE_ERR_NO_DEVICE, "smoe_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SMOE_ERR_INVALID, "smoe_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device), "hipSetDevice");

    smoe_context* h = new (std::nothrow) smoe_context();
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_create: out of host memory");
    h->cfg = *cfg;
    h->N = (int)N;
    h->force_g = 0;
    h->d_coords = nullptr;
    h->d_probes = nullptr;
    const int D = cfg->dim;

    // per-pixel coordinates [D][N], 'ij' meshgrid flattened row-major (smoe.py:2418-2421,1650)
    std::vector<std::vector<float>> axes(D);
    for (int l = 0; l < D; ++l) linspace01(cfg->block_shape[l], axes[l]);
    h->h_coords.assign((size_t)D * N, 0.0f);
    for (long n = 0; n < N; ++n) {
        long rem = n;
        for (int l = D - 1; l >= 0; --l) {
            const int idx = (int)(rem % cfg->block_shape[l]);
            rem /= cfg->block_shape[l];
            h->h_coords[(size_t)l * N + n] = axes[l][idx];
        }
    }
    // probes {min, max, (min+max)/2} per axis (smoe.py:2322-2333)
    std::vector<float> probes((size_t)D * 3);
    for (int l = 0; l < D; ++l) {
        const double mn = axes[l].front(), mx = axes[l].back();
        probes[l * 3 + 0] = (float)mn;
        probes[l * 3 + 1] = (float)mx;
        probes[l * 3 + 2] = (float)((mn + mx) / 2.0);
    }
    hipError_t e = hipMalloc(&h->d_coords, sizeof(float) * D * N);
    if (e == hipSuccess) e = hipMalloc(&h->d_probes, sizeof(float) * D * 3);
    if (e == hipSuccess) e = hipMemcpy(h->d_coords, h->h_coords.data(), sizeof(float) * D * N, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_probes, probes.data(), sizeof(float) * D * 3, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->d_coords) (void)hipFree(h->d_coords);
        if (h->d_probes) (void)hipFree(h->d_probes);
        delete h;
        return fail_hip(e, "smoe_create: workspace");
    }

    smoe::KernelConsts& kc = h->kc;
    const double two_p = std::ldexp(1.0, cfg->precision);
    kc.tau = (float)(0.5 * 1.0 / two_p);                 // smoe.py:825
    kc.epsm = (float)((double)cfg->margin * 1.0 / two_p); // smoe.py:931
    const float levels = (float)(two_p - 1.0);
    kc.scale = 1.0f / levels;                            // TF Nudge(): (max-min)/(quant_max-quant_min)
    kc.inv_scale = 1.0f / kc.scale;
    kc.nudged_max = fminf(1.0f, levels * kc.scale);      // combined clip_by_value(0,1) + nudged range
    const int C = cfg->channels;
    for (int c = 0; c < SMOE_MAX_CHANNELS; ++c) kc.cw[c] = 0.0f;
    for (int c = 0; c < C; ++c) {
        if (cfg->use_yuv) kc.cw[c] = (float)(((c == 0) ? 6.0 / 8.0 : 1.0 / 8.0) / (double)N);   // smoe.py:934
        else kc.cw[c] = (float)(1.0 / ((double)N * C));                                          // smoe.py:937
    }
    kc.n_dis = (float)std::sqrt(std::pow(2.0 * M_PI, (double)D));                                // smoe.py:812
    kc.use_det = cfg->use_determinant ? 1 : 0;
    kc.train_gammas = cfg->train_gammas ? 1 : 0;
    *out = h;
    return SMOE_OK;
}

int smoe_destroy(smoe_handle h) {
    if (!h) return SMOE_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->d_coords) (void)hipFree(h->d_coords);
    if (h->d_probes) (void)hipFree(h->d_probes);
    delete h;
    return SMOE_OK;
}

int smoe_get_coords(smoe_handle h, float* host_out) {
    if (!h || !host_out) return fail(SMOE_ERR_INVALID, "smoe_get_coords: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    HIP_TRY(hipMemcpy(host_out, h->d_coords, sizeof(float) * h->cfg.dim * h->N, hipMemcpyDeviceToHost), "smoe_get_coords");
    return SMOE_OK;
}

int smoe_set_tiling(smoe_handle h, int32_t lanes_per_block) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_set_tiling: null handle");
    if (lanes_per_block != 0 && lanes_per_block != 16 && lanes_per_block != 64)
        return fail(SMOE_ERR_INVALID, "smoe_set_tiling: lanes_per_block must be 0, 16 or 64");
    h->force_g = lanes_per_block;
    return SMOE_OK;
}

const char* smoe_fit_variant(smoe_handle h, int32_t num_blocks) {
    if (!h) return "";
    const smoe::Variant* v = find_variant(h, num_blocks, false);
    return v ? v->name : "";
}

int smoe_fit_occupancy(smoe_handle h, int32_t num_blocks) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_fit_occupancy: null handle");
    const smoe::Variant* v = find_variant(h, num_blocks, false);
    if (!v) return fail(SMOE_ERR_UNSUPPORTED, "smoe_fit_occupancy: no variant");
    if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(SMOE_ERR_HIP, "hipSetDevice");
    return v->fit_waves_per_cu(h->N, false);
}

int smoe_forward(smoe_handle h, int32_t num_blocks, const float* target, const float* loss_w,
                 const smoe_params* p, float* recon, uint8_t* argmax, float* gate_w,
                 float* loss, float* sse, uint32_t* active, int32_t update_active, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_forward: null handle");
    if (num_blocks < 0) return fail(SMOE_ERR_INVALID, "smoe_forward: negative num_blocks");
    if (num_blocks == 0) return SMOE_OK;
    if (!target || !params_ok(p) || !active) return fail(SMOE_ERR_INVALID, "smoe_forward: target, params and active are required");
    const smoe::Variant* v = find_variant(h, num_blocks, loss_w != nullptr);
    if (!v) return fail(SMOE_ERR_UNSUPPORTED, "smoe_forward: no kernel variant fits this block size in LDS");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::FwdArgs a;
    a.target = target; a.loss_w = loss_w; a.p = *p;
    a.recon = recon; a.argmax = argmax; a.gate_w = gate_w; a.loss = loss; a.sse = sse; a.active = active;
    a.coords = h->d_coords; a.B = num_blocks; a.N = h->N; a.update_active = update_active;
    a.reg_pi = h->cfg.pis_l1 / (float)(h->cfg.start_pis > 0 ? h->cfg.start_pis : h->cfg.kernels);
    a.reg_u = h->cfg.u_l1;
    a.kc = h->kc;
    HIP_TRY(v->fwd(a, (hipStream_t)stream), "smoe_forward launch");
    return SMOE_OK;
}

int smoe_fit(smoe_handle h, int32_t num_blocks, const float* target, const float* loss_w,
             smoe_params* p, smoe_adam_state* s, int32_t n_iters,
             float* loss_last, float* sse_last, uint32_t* active, uint32_t* diverged,
             const float* loss0, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_fit: null handle");
    if (num_blocks < 0 || n_iters < 0) return fail(SMOE_ERR_INVALID, "smoe_fit: negative num_blocks / n_iters");
    if (num_blocks == 0 || n_iters == 0) return SMOE_OK;
    if (!target || !params_ok(p) || !s || !params_ok(&s->m) || !params_ok(&s->v) || !active)
        return fail(SMOE_ERR_INVALID, "smoe_fit: target, params, adam state and active are required");
    const smoe::Variant* v = find_variant(h, num_blocks, loss_w != nullptr);
    if (!v) return fail(SMOE_ERR_UNSUPPORTED, "smoe_fit: no kernel variant fits this block size in LDS");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    const smoe_config& c = h->cfg;
    smoe::FitArgs a;
    a.target = target; a.loss_w = loss_w; a.p = *p; a.m = s->m; a.v = s->v;
    a.loss_out = loss_last; a.sse_out = sse_last; a.active = active; a.diverged = diverged; a.loss0 = loss0;
    a.coords = h->d_coords; a.B = num_blocks; a.N = h->N; a.n_iters = n_iters;
    a.b1p = s->beta1_power; a.b2p = s->beta2_power; a.beta1 = c.beta1; a.beta2 = c.beta2; a.eps = c.adam_eps;
    a.lr_expert = c.lr_expert; a.lr_pis = c.lr_pis; a.lr_steer = c.lr_steer;
    a.train_pis = c.train_pis; a.train_musx = c.train_musx;
    a.clip = c.grad_clip;
    a.reg_pi = c.pis_l1 / (float)(c.start_pis > 0 ? c.start_pis : c.kernels);
    a.reg_u = c.u_l1;
    a.kc = h->kc;
    // lanes walk the block with stride G: when G is a multiple of the last axis, a lane's last
    // coordinate never changes and the kernel hoists it (fit_kernel<..., XL = true>)
    const bool xl = (v->G % c.block_shape[c.dim - 1]) == 0;
    HIP_TRY(v->fit(a, xl, (hipStream_t)stream), "smoe_fit launch");
    // TF multiplies the beta powers after every apply (fp32 running product)
    for (int i = 0; i < n_iters; ++i) {
        s->beta1_power *= c.beta1;
        s->beta2_power *= c.beta2;
    }
    s->step += n_iters;
    return SMOE_OK;
}

int smoe_update_kernel_list(smoe_handle h, int32_t num_blocks, const smoe_params* p,
                            uint32_t* active, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_update_kernel_list: null handle");
    if (num_blocks < 0) return fail(SMOE_ERR_INVALID, "smoe_update_kernel_list: negative num_blocks");
    if (num_blocks == 0) return SMOE_OK;
    if (!params_ok(p) || !active) return fail(SMOE_ERR_INVALID, "smoe_update_kernel_list: params and active are required");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::ReadmitArgs a;
    a.p = *p; a.active = active; a.probes = h->d_probes; a.B = num_blocks; a.K = h->cfg.kernels;
    HIP_TRY(smoe::launch_readmit(a, h->cfg.dim, (hipStream_t)stream), "smoe_update_kernel_list launch");
    return SMOE_OK;
}

int smoe_checkpoint_best(smoe_handle h, int32_t num_blocks, const float* loss, float* best_loss,
                         const smoe_params* p, smoe_params* best, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_checkpoint_best: null handle");
    if (num_blocks < 0) return fail(SMOE_ERR_INVALID, "smoe_checkpoint_best: negative num_blocks");
    if (num_blocks == 0) return SMOE_OK;
    if (!loss || !best_loss || !params_ok(p) || !params_ok(best))
        return fail(SMOE_ERR_INVALID, "smoe_checkpoint_best: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::BestArgs a;
    a.loss = loss; a.best_loss = best_loss; a.p = *p; a.best = *best;
    a.B = num_blocks; a.K = h->cfg.kernels; a.D = h->cfg.dim; a.C = h->cfg.channels;
    HIP_TRY(smoe::launch_best(a, (hipStream_t)stream), "smoe_checkpoint_best launch");
    return SMOE_OK;
}

int smoe_reduce_scalars(smoe_handle h, int32_t num_blocks, const float* loss, const float* sse,
                        const uint32_t* active, double* out_dev, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_reduce_scalars: null handle");
    if (num_blocks < 0 || !out_dev) return fail(SMOE_ERR_INVALID, "smoe_reduce_scalars: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::ReduceArgs a;
    a.loss = loss; a.sse = sse; a.active = active; a.out = out_dev; a.B = num_blocks; a.N = h->N;
    HIP_TRY(smoe::launch_reduce(a, (hipStream_t)stream), "smoe_reduce_scalars launch");
    return SMOE_OK;
}

}  // extern "C"
