// smoe_capi.hip -- host side of libsmoe_hip.so: handle management, argument checking
// and kernel dispatch behind the C ABI declared in include/smoe_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "smoe_device.h"
#include "smoe_hip.h"

struct smoe_context {
    smoe_config cfg;
    int N;
    float* d_coords;     // [D][N]
    float* d_probes;     // [D][3]
    float* d_ssim_T;     // ssim_opt: banded tap tables Tr [bh][11], Tc [bw][11]
    double* d_partials;  // workspace of smoe_reduce_scalars
    uint32_t* d_dbg;     // SMOE_DEBUG build: word the kernels' device-side checks report into (null otherwise)
    int force_g;
    const float* mus_grid;   // use_diff_center: kernel-grid centres [B,K,D] of the blocks the calls pass (smoe_set_center_grid), or null
    mutable int big_g;   // lanes per block for blocks of more than 512 pixels (big_block_lanes; -1: not asked yet)
    int pair_occ;        // wavefronts per CU the 64-lane fit kernel reaches (-1: not asked yet)
    mutable int duo_occ[2];   // wavefronts per CU of the duo kernel without / with loss weights (-2: not asked yet, -1: unknown)
    int force_pair;      // 0: by batch size, 1: one block per 2-wavefront workgroup (smoe_set_tiling 128), -1: never
    long long total_blocks;   // smoe_set_total_blocks: block count of the whole job the calls are shards of (0: each call's own)
    int force_team;      // 0: by batch size, 2 / 4 / 8: team tiling with that many wavefronts per workgroup, -1: never
    int force_duo;       // 0: by batch size, 1: duo tiling (smoe_set_tiling 264), -1: never
    int lw_is_sample;    // smoe_set_sampling: the loss_w of smoe_fit is a pixel sub-sample
    int simds;           // SIMDs of the device (4 per CU; 1 024 on MI355X): the batch-size thresholds of the tiling rules scale with it
    std::string variant_name;   // what smoe_fit_variant last returned (the team names are composed)
    // resident wavefronts per CU of the fit kernels asked about so far: (variant, kind 0 plain / 1 pair / 2 duo, loss weights) -> count
    std::vector<std::pair<std::tuple<const void*, int, int>, int>> occ_cache;
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

// SMOE_HOST_TEST (make hostcheck; tests/host/hostcheck_driver.cpp): the host layer built for a box without a GPU and run
// under AddressSanitizer + UBSan -- handles are created without a device and without device allocations, and the HIP calls of
// the entry points (hipSetDevice, copies, kernel launches) are compiled out, so that every entry point runs its argument
// checks, its kernel-variant selection and its size arithmetic up to the launch.  Never defined in the product build.
#ifndef SMOE_HOST_TEST
#define SMOE_HOST_TEST 0
#endif
#if SMOE_HOST_TEST
#define HIP_TRY(expr, what) do { } while (0)
#else
#define HIP_TRY(expr, what)                              \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return fail_hip(_e, what); \
    } while (0)
#endif

// Workspace memory of a handle.  SMOE_HOST_TEST: host heap instead of device memory, so that the sanitizers see every
// size and every copy of the set-up code.
#if SMOE_HOST_TEST
template <typename T> hipError_t dev_malloc(T** p, size_t n) { *p = (T*)std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t dev_upload(void* d, const void* src, size_t n) { std::memcpy(d, src, n); return hipSuccess; }
hipError_t dev_download(void* d, const void* src, size_t n) { std::memcpy(d, src, n); return hipSuccess; }
hipError_t dev_zero(void* d, size_t n) { std::memset(d, 0, n); return hipSuccess; }
void dev_free(void* p) { std::free(p); }
#else
template <typename T> hipError_t dev_malloc(T** p, size_t n) { return hipMalloc(p, n); }
hipError_t dev_upload(void* d, const void* src, size_t n) { return hipMemcpy(d, src, n, hipMemcpyHostToDevice); }
hipError_t dev_download(void* d, const void* src, size_t n) { return hipMemcpy(d, src, n, hipMemcpyDeviceToHost); }
hipError_t dev_zero(void* d, size_t n) { return hipMemset(d, 0, n); }
void dev_free(void* p) { (void)hipFree(p); }
#endif

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

// ssim_opt: per axis the b x b matrix of "SYMMETRIC pad by 5, correlate with the 11-tap Gaussian, VALID"
// (smoe.py:993-996; image_ops_impl.py:132-149: softmax of -0.5 (a-5)^2 / 1.5^2, separable):
// T[i][j] = sum_a g[a] [mirror(i + a - 5) == j], stored banded: out[i][a] = T[i][i + a - 5], 0 outside the axis
void ssim_axis_table(int b, float* out) {
    double g[11], sum = 0.0;
    for (int a = 0; a < 11; ++a) { g[a] = std::exp(-0.5 * (a - 5) * (a - 5) / (1.5 * 1.5)); sum += g[a]; }
    std::vector<double> T((size_t)b * b, 0.0);
    for (int i = 0; i < b; ++i)
        for (int a = 0; a < 11; ++a) {
            int r = i + a - 5;
            if (r < 0) r = -1 - r;                 // SYMMETRIC: the edge sample is repeated
            if (r >= b) r = 2 * b - 1 - r;
            T[(size_t)i * b + r] += g[a] / sum;
        }
    for (int i = 0; i < b; ++i)
        for (int a = 0; a < 11; ++a) {
            const int j = i + a - 5;
            out[(size_t)i * 11 + a] = (j >= 0 && j < b) ? (float)T[(size_t)i * b + j] : 0.0f;
        }
}

// fake-quantised variables (smoe.py:474-538): nudged constants of the fixed ranges, TF Nudge() in fp32
void fill_quant_consts(smoe::KernelConsts& kc, int mode, int quantize_pis, int train_musx, const int32_t* bits,
                       const float* lb, const float* ub) {
    kc.qmode = (mode >= 2) ? mode : 0;
    kc.qpis = (mode >= 2 || quantize_pis) ? 1 : 0;
    kc.q_musx = train_musx ? 1 : 0;
    for (int g = 0; g < 5; ++g) {
        kc.q_nmin[g] = kc.q_nmax[g] = kc.q_scale[g] = kc.q_inv[g] = 0.0f;
        kc.q_levels[g] = 1.0f;
        if (!(kc.qmode || (g == 3 && kc.qpis))) continue;
        const float levels = (float)(std::ldexp(1.0, bits[g]) - 1.0);
        const float mn = lb[g], mx = ub[g];
        const float sc = (mx - mn) / levels;
        const float zp = 0.0f - mn / sc;
        const float nzp = (zp < 0.0f) ? 0.0f : ((zp > levels) ? levels : std::round(zp));
        kc.q_levels[g] = levels;
        kc.q_scale[g] = sc;
        kc.q_inv[g] = 1.0f / sc;
        kc.q_nmin[g] = (0.0f - nzp) * sc;
        kc.q_nmax[g] = (levels - nzp) * sc;
    }
}

int check_quant_config(int mode, int quantize_pis, const int32_t* bits, const float* lb, const float* ub, const char** msg) {
    if (mode < 0 || mode > 3) { *msg = "quantization_mode must be 0..3"; return SMOE_ERR_INVALID; }
    if (mode >= 2 || quantize_pis) {
        for (int g = 0; g < 5; ++g) {
            if (mode < 2 && g != 3) continue;
            if (bits[g] < 2 || bits[g] > 24) { *msg = "bit_depths must be 2..24 (fp32 lattice)"; return SMOE_ERR_INVALID; }
            if ((mode != 3 || g == 3) && !(lb[g] < ub[g])) { *msg = "lower_bounds must be below upper_bounds"; return SMOE_ERR_INVALID; }
        }
    }
    return SMOE_OK;
}

// Lanes walk the block with stride G: when G is a multiple of the last axis (of the last two axes), a lane's last (two)
// coordinate(s) never change and the kernels hoist every term in them out of the pixel loop (fit_kernel<..., HL>).
int hoist_level(const smoe_context* h, const smoe::Variant* v) {
    const smoe_config& c = h->cfg;
    const int last = c.block_shape[c.dim - 1];
    if (v->G % last != 0) return 0;
    if (c.dim == 3 && v->G % (last * c.block_shape[c.dim - 2]) == 0) return 2;
    return 1;
}

// What a launch needs from a variant (the basic instantiations of smoe_variants.def lack the SSIM / mode-2,3 kernels).
bool variant_serves(const smoe::Variant& v, const smoe_context* h) {
    if (h->cfg.ssim_opt) return v.fit_ssim != nullptr;
    if (h->kc.qmode) return v.fit_quant != nullptr;
    return true;
}

// Lanes per block by batch size (measured on 16x16 blocks, profiles/r02/bench_shapes.txt): many blocks -> 16 lanes per
// block (4 blocks per wavefront: the per-iteration work outside the pixel loop is amortised over 16 pixels per lane);
// few blocks -> more lanes per block, so that every SIMD of the 256 CUs has wavefronts to interleave.
// Large blocks (> 512 pixels): big_block_lanes -- 16 lanes would leave 64+ pixels per lane.
// Blocks of more than 512 pixels: a whole wavefront per block -- unless that kernel's registers allow ONE wavefront per
// SIMD only and the 32-lane kernel hoists as much: lone wavefronts either way, and two blocks per wavefront share the
// per-iteration work outside the pixel loop (32x32 / K = 8 / RGB, 316 VGPRs: 76.9 -> 88.2 Gpx-it/s at 2 040 blocks;
// 16x16x4 blocks stay on 64 lanes, where both trailing axes are hoisted: 202 vs 120).
int big_block_lanes(const smoe_context* h) {
    if (h->big_g > 0) return h->big_g;
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    const smoe::Variant *v64 = nullptr, *v32 = nullptr;
    for (int i = 0; i < n; ++i) {
        if (v[i].D != h->cfg.dim || v[i].C != h->cfg.channels || v[i].K != h->cfg.kernels || !variant_serves(v[i], h)) continue;
        if (v[i].lds_bytes(h->N, false, h->cfg.quantization_mode >= 2) > 160u * 1024u) continue;
        if (v[i].G == 64) v64 = &v[i];
        if (v[i].G == 32) v32 = &v[i];
    }
    int g = 64;
    if (v64 && v32 && !h->cfg.ssim_opt && hoist_level(h, v32) >= hoist_level(h, v64) && hipSetDevice(h->cfg.device) == hipSuccess) {
        const int occ = v64->fit_waves_per_cu(h->N, false, hoist_level(h, v64), false);
        if (occ > 0 && occ < 8) g = 32;
    }
    h->big_g = g;
    return g;
}

// The block count the kernel choice is made from: the whole job's when the caller declared it (smoe_set_total_blocks:
// the tiling fixes the summation order of a block's gradient terms, so every shard of a job must take the same one).
int choice_blocks(const smoe_context* h, int num_blocks) {
    if (h->total_blocks <= 0) return num_blocks;
    return (h->total_blocks > 0x7fffffffLL) ? 0x7fffffff : (int)h->total_blocks;
}

// The thresholds were measured on MI355X (256 CUs = 1 024 SIMDs) and are kept as blocks PER SIMD, so that a part with another
// CU count (or a partitioned one) moves them with it: what decides is how many wavefronts each SIMD gets to interleave.
int wanted_lanes(const smoe_context* h, int num_blocks) {
    if (h->force_g) return h->force_g;
    const long s = h->simds;
    if (h->N > 512) return (2L * num_blocks >= 3 * s) ? big_block_lanes(h) : 64;     // < 1.5 per SIMD: 32 lanes would leave SIMDs without a wavefront
    if (num_blocks >= 8 * s) return 16;     // 8 192 blocks: 276 (16 lanes) vs 213 (32) Gpx-it/s
    if (num_blocks >= 3 * s) return 32;     // 4 096 blocks: 208 (32) vs 177 (16) vs 157 (64); 2 048 blocks: 2 wavefronts per SIMD on 64 lanes win
    return 64;
}

// One block on BOTH wavefronts of a workgroup (fit_kernel PAIR): when the batch leaves at most one wavefront per SIMD
// (1 024 SIMDs) a second one is nearly free to the SIMD (profiles/r02/ubench_valu.txt: 5.6 -> 3.2 cycles per instruction),
// provided the kernel's registers let two wavefronts share a SIMD.  Measured (scripts/pair_check.py): ONE 512x512 image
// 86 -> 93 Gpx-it/s, 1 020 blocks of 16x16x4 RGB 92 -> 123; 1 536 blocks 124 -> 108 (no longer pays).
constexpr int PAIR_MAX_BLOCKS = 1024;
#ifndef SMOE_TEAM_MAX_BLOCKS
#define SMOE_TEAM_MAX_BLOCKS 0          // automatic team tiling up to this many blocks (0: only when forced, smoe_set_tiling)
#endif
bool wants_pair(smoe_context* h, const smoe::Variant* v, int num_blocks) {
    if (v->G != 64 || v->W != 2 || h->N < 128) return false;
    if (h->cfg.ssim_opt || h->kc.qmode || h->kc.inverse_cov) return false;
    if (h->force_pair) return h->force_pair > 0;
    if (choice_blocks(h, num_blocks) > (long)PAIR_MAX_BLOCKS * h->simds / 1024) return false;     // at most one block per SIMD
    if (h->pair_occ < 0) h->pair_occ = v->fit_waves_per_cu(h->N, false, hoist_level(h, v), true);   // of the PAIR kernel itself
    return h->pair_occ >= 8;                                   // two wavefronts per SIMD can be resident together
}

// SMOE_DEBUG build (make EXTRA=-DSMOE_DEBUG=1): the kernels report failed device-side checks (LDS carve-up vs the dynamic LDS of
// the launch) into h->d_dbg; the call waits for its launch and fails if a bit is set.
#ifndef SMOE_DEBUG
#define SMOE_DEBUG 0
#endif
int check_debug_word(smoe_context* h, hipStream_t st, const char* who) {
#if SMOE_DEBUG && !SMOE_HOST_TEST
    uint32_t w = 0;
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(&w, h->d_dbg, sizeof w, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(SMOE_ERR_HIP, std::string(who) + ": SMOE_DEBUG read-back failed");
    if (w != 0) {
        (void)hipMemset(h->d_dbg, 0, sizeof w);
        return fail(SMOE_ERR_HIP, std::string(who) + ": SMOE_DEBUG device check failed (the kernel's LDS carve-up exceeds the dynamic LDS of "
                                  "the launch), code " + std::to_string(w));
    }
#else
    (void)h; (void)st; (void)who;
#endif
    return SMOE_OK;
}

// Team tiling (smoe_team.hip.h): four blocks per workgroup on the 16-lane layout, the workgroup's wavefronts split the pixel
// rows.  Returns the wavefronts per workgroup (2, 4, 8) or 0 = run the regular kernels.  The default margin-loss graph only
// (with quantize_pis, the l1 terms, loss weights), block shapes whose last axis divides 16.
const smoe::Variant* find_g16(const smoe_context* h) {
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    for (int i = 0; i < n; ++i)
        if (v[i].D == h->cfg.dim && v[i].C == h->cfg.channels && v[i].K == h->cfg.kernels && v[i].G == 16 && v[i].fit_team) return &v[i];
    return nullptr;
}

int team_waves(const smoe_context* h, int num_blocks, bool has_lw, const smoe::Variant** v16_out) {
    if (h->force_team < 0 || (h->force_g && h->force_team == 0)) return 0;
    if (h->cfg.ssim_opt || h->kc.qmode || h->kc.inverse_cov || h->kc.radial) return 0;
    const smoe::Variant* v16 = find_g16(h);
    if (!v16 || hoist_level(h, v16) < 1) return 0;
    int nw = h->force_team;
    if (nw == 0) {
        // measured crossings (profiles/r03/bench_shapes.txt): below ~16 000 blocks the 16-lane kernels leave SIMDs with fewer
        // than four wavefronts; the team keeps the wavefront count up with the cheap per-block overhead of that tiling
        const int cb = choice_blocks(h, num_blocks);
        if (h->N > 512) return 0;
        if (cb > SMOE_TEAM_MAX_BLOCKS) return 0;
        nw = (cb <= 3072) ? 8 : ((cb <= 6144) ? 4 : 2);
    }
    if (v16->team_lds_bytes(h->N, has_lw, nw) > 160u * 1024u) return 0;
    if (v16_out) *v16_out = v16;
    return nw;
}

// Duo tiling (smoe_duo.hip.h): one block on two symmetric wavefronts, a single joint reduction, owner state in registers.
// Returns the 64-lane variant that carries it, or null = run the regular kernels.  Same graphs as the two-wavefront form of
// fit_kernel (wants_pair), triples with at most 128 slots.
// Automatic up to one block per SIMD and for blocks of at most 512 pixels (profiles/r03/bench_team.txt: 1 024 blocks of 16x16
// 92 -> 100+ Gpx-it/s against the two-wavefront form of fit_kernel, RGB 63 -> 70; 2 048 blocks: 104 vs 162 for one wavefront per
// block -- four wavefronts per SIMD no longer pay; 1 024-pixel video blocks: 78 vs 120, their two trailing axes are hoisted in
// fit_kernel and the pixel loop is most of their iteration).
#ifndef SMOE_DUO_MAX_BLOCKS
#define SMOE_DUO_MAX_BLOCKS 1024        // per 1 024 SIMDs
#endif
#ifndef SMOE_DUO_MAX_PIXELS
#define SMOE_DUO_MAX_PIXELS 512
#endif
const smoe::Variant* duo_variant(const smoe_context* h, int num_blocks, bool has_lw) {
    if (h->force_duo < 0 || (h->force_g && h->force_duo == 0) || h->force_team > 0) return nullptr;
    if (h->cfg.ssim_opt || h->kc.qmode || h->kc.inverse_cov || h->kc.radial) return nullptr;
    const long cb = choice_blocks(h, num_blocks);
    const long one_per_simd = (long)SMOE_DUO_MAX_BLOCKS * h->simds / 1024;
    if (h->force_duo == 0 && (h->force_pair > 0 || h->N > SMOE_DUO_MAX_PIXELS || cb > one_per_simd * 3 / 2)) return nullptr;
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    for (int i = 0; i < n; ++i) {
        if (v[i].D != h->cfg.dim || v[i].C != h->cfg.channels || v[i].K != h->cfg.kernels || v[i].G != 64 || !v[i].fit_duo) continue;
        const size_t b = v[i].duo_lds_bytes(h->N, has_lw, hoist_level(h, &v[i]));
        if (b == (size_t)-1 || b > 160u * 1024u) return nullptr;
        if (h->force_duo == 0 && cb > one_per_simd) {
            // Up to one and a half blocks per SIMD -- three of the kernel's wavefronts on every SIMD, all resident at once -- it
            // still beats one wavefront per block (1 536 blocks of 16x16: 151 vs 132 Gpx-it/s, 1 280: 131 vs 112; 1 792, which
            // needs a second round: 113 vs 148): only if the kernel's registers and LDS allow three wavefronts per SIMD.
#if SMOE_HOST_TEST
            return nullptr;
#else
            int& occ = h->duo_occ[has_lw ? 1 : 0];
            if (occ == -2) occ = v[i].duo_waves_per_cu(h->N, has_lw, hoist_level(h, &v[i]));
            if (occ < 12) return nullptr;
#endif
        }
        return &v[i];
    }
    return nullptr;
}

const smoe::Variant* find_variant(const smoe_context* h, int num_blocks, bool has_lw) {
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    const int want = wanted_lanes(h, choice_blocks(h, num_blocks));
    const smoe::Variant* fallback = nullptr;
    int fallback_dist = 1 << 30;
    const bool hq = h->cfg.quantization_mode >= 2;     // the mode-2/3 fit kernels keep a quantised parameter image in LDS
    for (int i = 0; i < n; ++i) {
        if (v[i].D != h->cfg.dim || v[i].C != h->cfg.channels || v[i].K != h->cfg.kernels) continue;
        if (!variant_serves(v[i], h)) continue;
        if (h->cfg.ssim_opt) {
            // 16x16 blocks: the register/DPP SSIM stage on the 16-lanes-per-block tiling; any other shape: the
            // LDS stage with one block per wavefront
            const bool b16 = h->cfg.dim == 2 && h->cfg.block_shape[0] == 16 && h->cfg.block_shape[1] == 16;
            const int g = h->force_g ? h->force_g : (b16 ? 16 : 64);
            if (v[i].G != g) continue;
            if (v[i].lds_bytes_ssim(h->N, has_lw, h->cfg.block_shape[0], h->cfg.block_shape[1], h->cfg.block_shape[2], hq) > 160u * 1024u) continue;
            return &v[i];
        }
        if (v[i].lds_bytes(h->N, has_lw, hq) > 160u * 1024u) continue;
        if (v[i].G == want) return &v[i];
        const int dist = (v[i].G > want) ? (v[i].G - want) : 4 * (want - v[i].G);     // prefer the next LARGER tiling
        if (dist < fallback_dist) { fallback = &v[i]; fallback_dist = dist; }
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

int smoe_padded_kernels(int32_t dim, int32_t channels, int32_t kernels) {
    int n = 0, best = -1;
    const smoe::Variant* v = smoe::variants(&n);
    for (int i = 0; i < n; ++i)
        if (v[i].D == dim && v[i].C == channels && v[i].K >= kernels && (best < 0 || v[i].K < best)) best = v[i].K;
    return best;
}

int smoe_padded_kernels_full(int32_t dim, int32_t channels, int32_t kernels) {
    int n = 0, best = -1;
    const smoe::Variant* v = smoe::variants(&n);
    for (int i = 0; i < n; ++i)
        if (v[i].D == dim && v[i].C == channels && v[i].K >= kernels && v[i].fit_ssim != nullptr && v[i].fit_quant != nullptr &&
            (best < 0 || v[i].K < best)) best = v[i].K;
    return best;
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
    {
        const char* qmsg = nullptr;
        const int qrc = check_quant_config(cfg->quantization_mode, cfg->quantize_pis, cfg->bit_depths, cfg->lower_bounds,
                                           cfg->upper_bounds, &qmsg);
        if (qrc != SMOE_OK) return fail(qrc, std::string("smoe_create: ") + qmsg);
    }
    if (cfg->ssim_opt) {
        // the reference pads every axis SYMMETRIC by 5 (smoe.py:993-1003), which TF only accepts for axes of
        // at least 5 samples; 3-d blocks take the 11x11x11 window (smoe.py:999-1003, custom_ssim ndim=3)
        if (cfg->dim != 2 && cfg->dim != 3) return fail(SMOE_ERR_UNSUPPORTED, "smoe_create: ssim_opt is built for 2-d and 3-d blocks");
        for (int ax = 0; ax < cfg->dim; ++ax)
            if (cfg->block_shape[ax] < 5)
                return fail(SMOE_ERR_INVALID, "smoe_create: ssim_opt needs at least 5 pixels per block axis (SYMMETRIC padding by 5)");
    }
    if (!smoe_is_supported(cfg->dim, cfg->channels, cfg->kernels)) {
        char buf[160];
        snprintf(buf, sizeof buf, "smoe_create: no kernel instantiated for (dim=%d, channels=%d, kernels=%d)",
                 cfg->dim, cfg->channels, cfg->kernels);
        return fail(SMOE_ERR_UNSUPPORTED, buf);
    }
#if !SMOE_HOST_TEST
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SMOE_ERR_NO_DEVICE, "smoe_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SMOE_ERR_INVALID, "smoe_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device), "hipSetDevice");
#endif

    smoe_context* h = new (std::nothrow) smoe_context();
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_create: out of host memory");
    h->cfg = *cfg;
    h->N = (int)N;
    h->force_g = 0;
    h->force_pair = 0;
    h->force_team = 0;
    h->force_duo = 0;
    h->lw_is_sample = 0;
    h->simds = 1024;
#if !SMOE_HOST_TEST
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) h->simds = 4 * cus;
    }
#endif
    h->total_blocks = 0;
    h->pair_occ = -1;
    h->duo_occ[0] = h->duo_occ[1] = -2;
    h->big_g = -1;
    h->mus_grid = nullptr;
    h->d_coords = nullptr;
    h->d_probes = nullptr;
    h->d_ssim_T = nullptr;
    h->d_partials = nullptr;
    h->d_dbg = nullptr;
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
    hipError_t e = dev_malloc(&h->d_coords, sizeof(float) * D * N);
    if (e == hipSuccess) e = dev_malloc(&h->d_probes, sizeof(float) * D * 3);
    if (e == hipSuccess) e = dev_upload(h->d_coords, h->h_coords.data(), sizeof(float) * D * N);
    if (e == hipSuccess) e = dev_upload(h->d_probes, probes.data(), sizeof(float) * D * 3);
    if (e == hipSuccess) e = dev_malloc(&h->d_partials, sizeof(double) * smoe::reduce_partials_count());
#if SMOE_DEBUG
    if (e == hipSuccess) e = dev_malloc(&h->d_dbg, sizeof(uint32_t));
    if (e == hipSuccess) e = dev_zero(h->d_dbg, sizeof(uint32_t));
#endif
    if (e == hipSuccess && cfg->ssim_opt) {
        const int bh = cfg->block_shape[0], bw = cfg->block_shape[1], bt = (cfg->dim == 3) ? cfg->block_shape[2] : 0;
        std::vector<float> tabs((size_t)11 * (bh + bw + bt));
        ssim_axis_table(bh, tabs.data());
        ssim_axis_table(bw, tabs.data() + (size_t)11 * bh);
        if (bt) ssim_axis_table(bt, tabs.data() + (size_t)11 * (bh + bw));
        e = dev_malloc(&h->d_ssim_T, sizeof(float) * tabs.size());
        if (e == hipSuccess) e = dev_upload(h->d_ssim_T, tabs.data(), sizeof(float) * tabs.size());
    }
    if (e != hipSuccess) {
        if (h->d_coords) dev_free(h->d_coords);
        if (h->d_probes) dev_free(h->d_probes);
        if (h->d_ssim_T) dev_free(h->d_ssim_T);
        if (h->d_partials) dev_free(h->d_partials);
        if (h->d_dbg) dev_free(h->d_dbg);
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
    kc.inv_n_dis = 1.0f / kc.n_dis;
    kc.use_det = cfg->use_determinant ? 1 : 0;
    kc.train_gammas = cfg->train_gammas ? 1 : 0;
    kc.only_y_gamma = (cfg->only_y_gamma && cfg->use_yuv && cfg->train_gammas) ? 1 : 0;   // smoe.py:725
    for (int c = 0; c < SMOE_MAX_CHANNELS; ++c) kc.sw[c] = 0.0f;
    for (int c = 0; c < C; ++c) {                       // smoe.py:1006-1009, mean over the bh*bw window positions
        const double w = cfg->use_yuv ? ((c == 0) ? 6.0 / 8.0 : 1.0 / 8.0) : 1.0 / (double)C;
        kc.sw[c] = (float)(w / (double)N);
    }
    fill_quant_consts(kc, cfg->quantization_mode, cfg->quantize_pis, cfg->train_musx, cfg->bit_depths,
                      cfg->lower_bounds, cfg->upper_bounds);
    kc.inverse_cov = cfg->train_inverse_cov ? 1 : 0;
    kc.radial = cfg->radial_as ? 1 : 0;
    kc.kcount_norm = cfg->kernel_count_as_norm_l1 ? 1 : 0;
    kc.pis_l1_raw = cfg->pis_l1;
    {
        int nv = 0;
        const smoe::Variant* vv = smoe::variants(&nv);
        bool served = false;
        for (int i = 0; i < nv; ++i)
            served = served || (vv[i].D == cfg->dim && vv[i].C == cfg->channels && vv[i].K == cfg->kernels && variant_serves(vv[i], h));
        if (!served) {
            smoe_destroy(h);
            return fail(SMOE_ERR_UNSUPPORTED, "smoe_create: ssim_opt / quantization_mode 2, 3 are built for the triples marked FULL in "
                                              "csrc/smoe_variants.def only (add the triple there and rebuild)");
        }
    }
    if (cfg->ssim_opt && !find_variant(h, 1, false)) {
        smoe_destroy(h);
        return fail(SMOE_ERR_UNSUPPORTED, "smoe_create: ssim_opt planes of this block size do not fit in LDS");
    }
    *out = h;
    return SMOE_OK;
}

int smoe_destroy(smoe_handle h) {
    if (!h) return SMOE_OK;
#if !SMOE_HOST_TEST
    (void)hipSetDevice(h->cfg.device);
#endif
    if (h->d_coords) dev_free(h->d_coords);
    if (h->d_probes) dev_free(h->d_probes);
    if (h->d_ssim_T) dev_free(h->d_ssim_T);
    if (h->d_partials) dev_free(h->d_partials);
    if (h->d_dbg) dev_free(h->d_dbg);
    delete h;
    return SMOE_OK;
}

int smoe_get_coords(smoe_handle h, float* host_out) {
    if (!h || !host_out) return fail(SMOE_ERR_INVALID, "smoe_get_coords: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    HIP_TRY(dev_download(host_out, h->d_coords, sizeof(float) * h->cfg.dim * h->N), "smoe_get_coords");
    return SMOE_OK;
}

int smoe_set_center_grid(smoe_handle h, const float* grid) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_set_center_grid: null handle");
    h->mus_grid = grid;
    return SMOE_OK;
}

int smoe_set_tiling(smoe_handle h, int32_t lanes_per_block) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_set_tiling: null handle");
    const bool team = lanes_per_block == 216 || lanes_per_block == 416 || lanes_per_block == 816;
    const bool duo = lanes_per_block == 264;
    if (!team && !duo && lanes_per_block != 0 && lanes_per_block != 16 && lanes_per_block != 32 && lanes_per_block != 64 && lanes_per_block != 128)
        return fail(SMOE_ERR_INVALID, "smoe_set_tiling: lanes_per_block must be 0, 16, 32, 64, 128, 216, 264, 416 or 816");
    // 128 = the 64-lane kernels with one block on both wavefronts of a workgroup (margin loss, quantization_mode 0 / 1,
    // train_inverse_cov off; other graphs run the plain 64-lane kernel)
    // 216 / 416 / 816 = team tiling with 2 / 4 / 8 wavefronts per workgroup in smoe_fit (the graphs it covers; others and
    // the evaluation choose as with 0)
    // 264 = duo tiling of smoe_fit (one block on two symmetric wavefronts, csrc/smoe_duo.hip.h)
    h->force_g = (lanes_per_block == 128) ? 64 : ((team || duo) ? 0 : lanes_per_block);
    h->force_pair = (lanes_per_block == 128) ? 1 : ((lanes_per_block == 0 || team || duo) ? 0 : -1);
    h->force_team = team ? lanes_per_block / 100 : ((lanes_per_block == 0) ? 0 : -1);
    h->force_duo = duo ? 1 : ((lanes_per_block == 0) ? 0 : -1);
    return SMOE_OK;
}

int smoe_set_sampling(smoe_handle h, int32_t on) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_set_sampling: null handle");
    h->lw_is_sample = on ? 1 : 0;
    return SMOE_OK;
}

int smoe_set_total_blocks(smoe_handle h, int64_t total_blocks) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_set_total_blocks: null handle");
    if (total_blocks < 0) return fail(SMOE_ERR_INVALID, "smoe_set_total_blocks: negative block count");
    h->total_blocks = (long long)total_blocks;
    return SMOE_OK;
}

const char* smoe_fit_variant(smoe_handle h, int32_t num_blocks) {
    if (!h) return "";
#if !SMOE_HOST_TEST
    (void)hipSetDevice(h->cfg.device);       // (the duo rule asks the device for the kernel's occupancy)
#endif
    if (const smoe::Variant* vd = duo_variant(h, num_blocks, false)) {
        h->variant_name = std::string(vd->name);
        const size_t g = h->variant_name.find("_g64");
        if (g != std::string::npos) h->variant_name.resize(g);
        h->variant_name += "_duo64w2";
        return h->variant_name.c_str();
    }
    const smoe::Variant* v16 = nullptr;
    const int nw = team_waves(h, num_blocks, false, &v16);
    if (nw > 0) {
        h->variant_name = std::string(v16->name);
        const size_t g = h->variant_name.find("_g16");
        if (g != std::string::npos) h->variant_name.resize(g);
        h->variant_name += "_team16w" + std::to_string(nw);
        return h->variant_name.c_str();
    }
    const smoe::Variant* v = find_variant(h, num_blocks, false);
    return v ? v->name : "";
}

int smoe_fit_occupancy(smoe_handle h, int32_t num_blocks) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_fit_occupancy: null handle");
    const smoe::Variant* v = find_variant(h, num_blocks, false);
    if (!v) return fail(SMOE_ERR_UNSUPPORTED, "smoe_fit_occupancy: no variant");
    if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(SMOE_ERR_HIP, "hipSetDevice");
    if (const smoe::Variant* vd = duo_variant(h, num_blocks, false)) return vd->duo_waves_per_cu(h->N, false, hoist_level(h, vd));
    const smoe::Variant* v16 = nullptr;
    const int nw = team_waves(h, num_blocks, false, &v16);
    if (nw > 0) return v16->team_waves_per_cu(h->N, false, nw);
    return v->fit_waves_per_cu(h->N, false, hoist_level(h, v), false);
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
    a.hoist = hoist_level(h, v);
    a.reg_pi = h->cfg.pis_l1 / (float)(h->cfg.start_pis > 0 ? h->cfg.start_pis : h->cfg.kernels);
    a.reg_u = h->cfg.u_l1;
    a.kc = h->kc;
    a.ssim_T = h->d_ssim_T; a.bh = h->cfg.block_shape[0]; a.bw = h->cfg.block_shape[1]; a.bt = h->cfg.block_shape[2];
    a.mus_grid = h->mus_grid;
    a.lds_floats = 0; a.dbg = h->d_dbg;
    if (h->cfg.ssim_opt) HIP_TRY(v->fwd_ssim(a, (hipStream_t)stream), "smoe_forward (ssim) launch");
    else if (h->kc.qmode) HIP_TRY(v->fwd_quant(a, (hipStream_t)stream), "smoe_forward (quantised) launch");
    else if (h->kc.inverse_cov) HIP_TRY(v->fwd_ic(a, (hipStream_t)stream), "smoe_forward (inverse covariance) launch");
    else HIP_TRY(v->fwd(a, (hipStream_t)stream), "smoe_forward launch");
    return check_debug_word(h, (hipStream_t)stream, "smoe_forward");
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
    const int hoist = hoist_level(h, v);
    (void)hoist;
    a.ssim_T = h->d_ssim_T; a.bh = c.block_shape[0]; a.bw = c.block_shape[1]; a.bt = c.block_shape[2];
    a.pair = wants_pair(h, v, num_blocks) ? 1 : 0;
    a.mus_grid = h->mus_grid;
    a.lds_floats = 0; a.dbg = h->d_dbg; a.desc_off = 0;
    a.lw_is_sample = (h->lw_is_sample && loss_w != nullptr && !c.ssim_opt) ? 1 : 0;
    const smoe::Variant* v16 = nullptr;
    const int team = team_waves(h, num_blocks, loss_w != nullptr, &v16);
    const smoe::Variant* vduo = duo_variant(h, num_blocks, loss_w != nullptr);
    a.prio_rotate = 0;
#if !SMOE_HOST_TEST
    {
        // one round = every wavefront of the launch resident at once: the wavefronts' priorities rotate (rotate_priority).
        // Occupancy of the plain kernel of the tiling (the SSIM / quantised / inverse-covariance instantiations need at least
        // as many registers: their launches are classed as one round a little too often, which costs at most 2 %)
        long waves = 0, per_cu = 0;
        const smoe::Variant* vo = vduo ? vduo : v;
        const int kind = vduo ? 2 : (a.pair ? 1 : 0);
        if (team <= 0) {
            waves = (kind != 0) ? 2L * num_blocks : ((long)num_blocks * v->G + 63) / 64;
            const auto key = std::make_tuple((const void*)vo, kind, (int)(loss_w != nullptr));
            bool found = false;
            for (const auto& e : h->occ_cache)
                if (e.first == key) { per_cu = e.second; found = true; break; }
            if (!found) {          // (an occupancy query per launch would cost the short launches of a small batch microseconds)
                per_cu = vduo ? vduo->duo_waves_per_cu(h->N, loss_w != nullptr, hoist_level(h, vduo))
                              : v->fit_waves_per_cu(h->N, loss_w != nullptr, hoist, a.pair != 0);
                h->occ_cache.emplace_back(key, (int)per_cu);
            }
        }
        if (per_cu > 0 && waves <= per_cu * (long)(h->simds / 4)) a.prio_rotate = 1;
    }
#endif
    if (vduo) HIP_TRY(vduo->fit_duo(a, hoist_level(h, vduo), (hipStream_t)stream), "smoe_fit (duo) launch");
    else if (team > 0) HIP_TRY(v16->fit_team(a, hoist_level(h, v16), team, (hipStream_t)stream), "smoe_fit (team) launch");
    else if (c.ssim_opt) HIP_TRY(v->fit_ssim(a, hoist, (hipStream_t)stream), "smoe_fit (ssim) launch");
    else if (h->kc.qmode) HIP_TRY(v->fit_quant(a, hoist, (hipStream_t)stream), "smoe_fit (quantised) launch");
    else if (h->kc.inverse_cov) HIP_TRY(v->fit_ic(a, hoist, (hipStream_t)stream), "smoe_fit (inverse covariance) launch");
    else HIP_TRY(v->fit(a, hoist, (hipStream_t)stream), "smoe_fit launch");
    // TF multiplies the beta powers after every apply (fp32 running product)
    for (int i = 0; i < n_iters; ++i) {
        s->beta1_power *= c.beta1;
        s->beta2_power *= c.beta2;
    }
    s->step += n_iters;
    return check_debug_word(h, (hipStream_t)stream, "smoe_fit");
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
    a.inverse_cov = h->kc.inverse_cov;
    a.mus_grid = h->mus_grid;
    if (h->kc.qmode || h->kc.qpis) {                 // the probe test runs on the fake-quantised variables
        const smoe::Variant* v = find_variant(h, num_blocks, false);
        if (!v) return fail(SMOE_ERR_UNSUPPORTED, "smoe_update_kernel_list: no kernel variant");
        HIP_TRY(v->readmit_quant(a, h->kc, (hipStream_t)stream), "smoe_update_kernel_list (quantised) launch");
        return SMOE_OK;
    }
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
    a.loss = loss; a.sse = sse; a.active = active; a.out = out_dev; a.partials = h->d_partials; a.B = num_blocks; a.N = h->N;
    HIP_TRY(smoe::launch_reduce(a, (hipStream_t)stream), "smoe_reduce_scalars launch");
    return SMOE_OK;
}

}  // extern "C"

// =================================================================================================
// shared-kernel image mode
// =================================================================================================
struct smoe_shared_context {
    smoe_shared_config cfg;
    int NB, Nb, KW, PK;
    int grid[SMOE_MAX_DIM];
    int axis_off[SMOE_MAX_DIM];
    float* d_axes;        // concatenated per-axis coordinate tables
    float* d_probes;      // [NB][D][3]
    double* d_racc;       // [K*PK + K]
    // fixed-order gradient accumulation (SharedGatherArgs): per-batch rows of raw sums, the kernels each batch trained, the
    // pass number a row belongs to; null when the rows would not fit (then fp64 atomics, not bit-deterministic)
    float* d_part;        // [NB][K][PK]
    uint32_t* d_trained;  // [NB][KW]
    uint32_t* d_batch_epoch;  // [NB]
    uint32_t epoch;       // current pass (bumped by smoe_shared_apply)
    uint32_t* d_bar;      // grid barrier of the one-launch fit (shared_fit_kernel): SharedFitArgs::bar
    size_t bar_words;
    bool fit_pending;     // a one-launch fit has been issued and its abort word not looked at yet
    int num_cus;          // compute units of the device (co-residency test of the one-launch fit)
    float* d_ssim_T;      // ssim_opt: banded tap tables of the batch shape
    float* d_qrng;        // SharedRangesArgs records (mode-3 ranges, count of qpis > 0)
    bool need_ranges;     // quantization_mode 3 or kernel_count_as_norm_l1
    const float* loss_w;  // caller-owned [NB][Nb] loss weights (smoe_shared_set_loss_weights) or null
    const float* mus_grid;   // caller-owned kernel-grid centres [K][D] (smoe_shared_set_center_grid) or null
    smoe::KernelConsts kc;
};

namespace {

void fill_shared_args(const smoe_shared_context* h, smoe::SharedArgs& a) {
    const smoe_shared_config& c = h->cfg;
    a.axis_coords = h->d_axes;
    for (int l = 0; l < SMOE_MAX_DIM; ++l) {
        a.axis_off[l] = h->axis_off[l];
        a.batch_shape[l] = (l < c.dim) ? c.batch_shape[l] : 1;
        a.grid[l] = (l < c.dim) ? h->grid[l] : 1;
        a.image_shape[l] = (l < c.dim) ? c.image_shape[l] : 1;
    }
    a.overlap = c.overlap;
    a.loss_w = h->loss_w;
    a.ssim = c.ssim_opt ? 1 : 0;
    a.ssim_T = h->d_ssim_T;
    a.ssim_off = (int)(smoe::shared_lds_bytes(c.dim, c.channels, c.kernels, h->KW) / sizeof(float));
    a.Nb = h->Nb; a.K = c.kernels; a.KW = h->KW;
    a.kc = h->kc;
    a.reg_pi = c.pis_l1 / (float)(c.start_pis > 0 ? c.start_pis : c.kernels);
    a.reg_u = c.u_l1;
    a.racc = h->d_racc;
    a.nact = h->d_racc + (size_t)c.kernels * h->PK;
    a.qrng = h->d_qrng;
    a.mus_grid = h->mus_grid;
    a.part = h->d_part; a.trained = h->d_trained; a.batch_epoch = h->d_batch_epoch; a.epoch = h->epoch;
}

void fill_gather_args(const smoe_shared_context* h, smoe::SharedGatherArgs& g) {
    g.part = h->d_part; g.trained = h->d_trained; g.batch_epoch = h->d_batch_epoch; g.epoch = h->epoch;
    g.NB_total = h->NB; g.K = h->cfg.kernels; g.KW = h->KW; g.PK = h->PK;
    g.racc = h->d_racc; g.nact = h->d_racc + (size_t)h->cfg.kernels * h->PK;
}

// the image-wide records follow the parameters of THIS call (the C ABI is stateless in the parameters)
hipError_t refresh_ranges(const smoe_shared_context* h, const smoe_params* p, hipStream_t st) {
    if (!h->need_ranges) return hipSuccess;
    smoe::SharedRangesArgs r;
    r.p = *p; r.qrng = h->d_qrng; r.K = h->cfg.kernels; r.kc = h->kc; r.mus_grid = h->mus_grid;
    return smoe::launch_shared_ranges(r, h->cfg.dim, h->cfg.channels, st);
}

int check_range(const smoe_shared_context* h, int first, int count, const char* who) {
    if (first < 0 || count < 0 || first + count > h->NB) return fail(SMOE_ERR_INVALID, std::string(who) + ": batch range out of bounds");
    return SMOE_OK;
}

}  // namespace

extern "C" {

int smoe_shared_create(smoe_shared_handle* out, const smoe_shared_config* cfg) {
    if (!out || !cfg) return fail(SMOE_ERR_INVALID, "smoe_shared_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != SMOE_ABI_VERSION) return fail(SMOE_ERR_INVALID, "smoe_shared_create: abi_version mismatch");
    if (cfg->dim < 2 || cfg->dim > SMOE_MAX_DIM) return fail(SMOE_ERR_INVALID, "smoe_shared_create: dim must be 2 or 3");
    if (cfg->channels != 1 && cfg->channels != 3) return fail(SMOE_ERR_UNSUPPORTED, "smoe_shared_create: channels must be 1 or 3");
    if (cfg->kernels < 1 || cfg->kernels > 8192) return fail(SMOE_ERR_INVALID, "smoe_shared_create: kernels must be 1..8192");
    if (cfg->precision < 1 || cfg->precision > 16) return fail(SMOE_ERR_INVALID, "smoe_shared_create: precision must be 1..16");
    long Nb = 1, NB = 1;
    for (int l = 0; l < cfg->dim; ++l) {
        if (cfg->batch_shape[l] < 1 || cfg->image_shape[l] < 1 || cfg->image_shape[l] % cfg->batch_shape[l] != 0)
            return fail(SMOE_ERR_INVALID, "smoe_shared_create: Required BatchSize is not compatible to input dimensions");  // smoe.py:241
        Nb *= cfg->batch_shape[l];
        NB *= cfg->image_shape[l] / cfg->batch_shape[l];
    }
    if (cfg->overlap < 0 || cfg->overlap > 64) return fail(SMOE_ERR_INVALID, "smoe_shared_create: overlap must be 0..64");
    {
        const char* qmsg = nullptr;
        const int qrc = check_quant_config(cfg->quantization_mode, cfg->quantize_pis, cfg->bit_depths, cfg->lower_bounds,
                                           cfg->upper_bounds, &qmsg);
        if (qrc != SMOE_OK) return fail(qrc, std::string("smoe_shared_create: ") + qmsg);
    }
    if (cfg->ssim_opt) {
        for (int ax = 0; ax < cfg->dim; ++ax)
            if (cfg->batch_shape[ax] < 5)
                return fail(SMOE_ERR_INVALID, "smoe_shared_create: ssim_opt needs at least 5 pixels per batch axis (SYMMETRIC padding by 5)");
    }
    if (!smoe::shared_supported(cfg->dim, cfg->channels, (int)Nb))
        return fail(SMOE_ERR_UNSUPPORTED, "smoe_shared_create: batch too large (<= 2048 pixels for 1 channel, <= 1024 for 3)");
    const int KW = (cfg->kernels + 31) / 32;
    if (smoe::shared_lds_bytes(cfg->dim, cfg->channels, cfg->kernels, KW) > 160u * 1024u)
        return fail(SMOE_ERR_UNSUPPORTED, "smoe_shared_create: kernel list does not fit in LDS");
#if !SMOE_HOST_TEST
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SMOE_ERR_NO_DEVICE, "smoe_shared_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SMOE_ERR_INVALID, "smoe_shared_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device), "hipSetDevice");
#endif
    smoe_shared_context* h = new (std::nothrow) smoe_shared_context();
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_create: out of host memory");
    h->cfg = *cfg;
    h->NB = (int)NB; h->Nb = (int)Nb; h->KW = KW;
    const int D = cfg->dim, C = cfg->channels;
    h->PK = 1 + D + D * (D + 1) / 2 + C + D * C;
    // global axis tables linspace(0,1,size) (smoe.py:2412) and per-batch probes (smoe.py:2322-2333)
    std::vector<float> axes;
    std::vector<std::vector<double>> axd(D);
    for (int l = 0; l < SMOE_MAX_DIM; ++l) { h->grid[l] = 1; h->axis_off[l] = 0; }
    for (int l = 0; l < D; ++l) {
        h->grid[l] = cfg->image_shape[l] / cfg->batch_shape[l];
        h->axis_off[l] = (int)axes.size();
        const int n = cfg->image_shape[l];
        axd[l].resize(n);
        const double step = n > 1 ? 1.0 / (double)(n - 1) : 0.0;
        for (int i = 0; i < n; ++i) axd[l][i] = (n > 1 && i == n - 1) ? 1.0 : (double)i * step;
        for (int i = 0; i < n; ++i) axes.push_back((float)axd[l][i]);
    }
    std::vector<float> probes((size_t)NB * D * 3);
    for (long b = 0; b < NB; ++b) {
        // min / max / mid of the batch's window per axis (smoe.py:2322-2331).  With a halo the window is
        // cut from the zero-padded joint domain (smoe.py:21,28): a window that leaves the image on ANY
        // axis contains all-zero pixels, which pulls the minimum of every coordinate to 0
        long rem = b;
        int org[SMOE_MAX_DIM] = {0, 0, 0};
        bool leaves_image = false;
        for (int l = D - 1; l >= 0; --l) {
            org[l] = (int)(rem % h->grid[l]) * cfg->batch_shape[l];
            rem /= h->grid[l];
            leaves_image = leaves_image || (org[l] - cfg->overlap < 0) ||
                           (org[l] + cfg->batch_shape[l] + cfg->overlap > cfg->image_shape[l]);
        }
        for (int l = 0; l < D; ++l) {
            const int lo = std::max(0, org[l] - cfg->overlap);
            const int hi = std::min(cfg->image_shape[l], org[l] + cfg->batch_shape[l] + cfg->overlap) - 1;
            const double mn = leaves_image ? 0.0 : axd[l][lo], mx = axd[l][hi];
            probes[((size_t)b * D + l) * 3 + 0] = (float)mn;
            probes[((size_t)b * D + l) * 3 + 1] = (float)mx;
            probes[((size_t)b * D + l) * 3 + 2] = (float)((mn + mx) / 2.0);
        }
    }
    const size_t nacc = (size_t)cfg->kernels * h->PK + cfg->kernels;
    h->d_axes = nullptr; h->d_probes = nullptr; h->d_racc = nullptr; h->d_ssim_T = nullptr; h->d_qrng = nullptr; h->loss_w = nullptr; h->mus_grid = nullptr;
    h->d_part = nullptr; h->d_trained = nullptr; h->d_batch_epoch = nullptr; h->epoch = 1u;
    h->d_bar = nullptr; h->num_cus = 0; h->fit_pending = false;
#if !SMOE_HOST_TEST
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) h->num_cus = cus;
    }
#endif
    h->need_ranges = cfg->quantization_mode == 3 || cfg->kernel_count_as_norm_l1 != 0;
    if (cfg->ssim_opt) {
        const size_t need = smoe::shared_lds_bytes(cfg->dim, cfg->channels, cfg->kernels, h->KW) +
                            smoe::shared_ssim_lds_bytes(cfg->channels, (int)Nb, cfg->batch_shape[0], cfg->batch_shape[1], (cfg->dim == 3) ? cfg->batch_shape[2] : 0);
        if (need > 160u * 1024u) {
            delete h;
            return fail(SMOE_ERR_UNSUPPORTED, "smoe_shared_create: ssim_opt planes of this batch size do not fit in LDS");
        }
    }
    hipError_t e = dev_malloc(&h->d_axes, sizeof(float) * axes.size());
    if (e == hipSuccess) e = dev_malloc(&h->d_probes, sizeof(float) * probes.size());
    if (e == hipSuccess) e = dev_malloc(&h->d_racc, sizeof(double) * nacc);
    if (e == hipSuccess) e = dev_upload(h->d_axes, axes.data(), sizeof(float) * axes.size());
    if (e == hipSuccess) e = dev_upload(h->d_probes, probes.data(), sizeof(float) * probes.size());
    if (e == hipSuccess) e = dev_zero(h->d_racc, sizeof(double) * nacc);
    {
        // one row of raw sums per (batch, kernel): 512x512 / 32x32 batches / 144 kernels = 1.3 MB; 2048x2048 / 2304 kernels =
        // 340 MB.  Beyond 4 GB the rows are not allocated and the pass falls back to fp64 atomics.
        const size_t rows = (size_t)NB * (size_t)cfg->kernels * (size_t)h->PK;
        if (rows * sizeof(float) <= ((size_t)4 << 30)) {
            if (e == hipSuccess) e = dev_malloc(&h->d_part, sizeof(float) * rows);
            if (e == hipSuccess) e = dev_malloc(&h->d_trained, sizeof(uint32_t) * (size_t)NB * KW);
            if (e == hipSuccess) e = dev_malloc(&h->d_batch_epoch, sizeof(uint32_t) * (size_t)NB);
            if (e == hipSuccess) e = dev_zero(h->d_batch_epoch, sizeof(uint32_t) * (size_t)NB);
        }
    }
    h->bar_words = (size_t)16 + (((size_t)NB + 3) & ~(size_t)3);
    if (e == hipSuccess) e = dev_malloc(&h->d_bar, sizeof(uint32_t) * h->bar_words);
    if (e == hipSuccess) e = dev_zero(h->d_bar, sizeof(uint32_t) * h->bar_words);
    if (e == hipSuccess) e = dev_malloc(&h->d_qrng, sizeof(float) * smoe::SHARED_QRNG_FLOATS);
    if (e == hipSuccess) e = dev_zero(h->d_qrng, sizeof(float) * smoe::SHARED_QRNG_FLOATS);
    if (e == hipSuccess && cfg->ssim_opt) {
        const int bh = cfg->batch_shape[0], bw = cfg->batch_shape[1], bt = (cfg->dim == 3) ? cfg->batch_shape[2] : 0;
        std::vector<float> tabs((size_t)11 * (bh + bw + bt));
        ssim_axis_table(bh, tabs.data());
        ssim_axis_table(bw, tabs.data() + (size_t)11 * bh);
        if (bt) ssim_axis_table(bt, tabs.data() + (size_t)11 * (bh + bw));
        e = dev_malloc(&h->d_ssim_T, sizeof(float) * tabs.size());
        if (e == hipSuccess) e = dev_upload(h->d_ssim_T, tabs.data(), sizeof(float) * tabs.size());
    }
    if (e != hipSuccess) {
        if (h->d_axes) dev_free(h->d_axes);
        if (h->d_probes) dev_free(h->d_probes);
        if (h->d_racc) dev_free(h->d_racc);
        if (h->d_ssim_T) dev_free(h->d_ssim_T);
        if (h->d_qrng) dev_free(h->d_qrng);
        if (h->d_part) dev_free(h->d_part);
        if (h->d_trained) dev_free(h->d_trained);
        if (h->d_batch_epoch) dev_free(h->d_batch_epoch);
        if (h->d_bar) dev_free(h->d_bar);
        delete h;
        return fail_hip(e, "smoe_shared_create: workspace");
    }
    smoe::KernelConsts& kc = h->kc;
    const double two_p = std::ldexp(1.0, cfg->precision);
    kc.tau = (float)(0.5 * 1.0 / two_p);
    kc.epsm = (float)((double)cfg->margin * 1.0 / two_p);
    const float levels = (float)(two_p - 1.0);
    kc.scale = 1.0f / levels;
    kc.inv_scale = 1.0f / kc.scale;
    kc.nudged_max = fminf(1.0f, levels * kc.scale);
    for (int c = 0; c < SMOE_MAX_CHANNELS; ++c) kc.cw[c] = 0.0f;
    for (int c = 0; c < C; ++c) {                                   // per-BATCH means (smoe.py:934-937)
        if (cfg->use_yuv) kc.cw[c] = (float)(((c == 0) ? 6.0 / 8.0 : 1.0 / 8.0) / (double)Nb);
        else kc.cw[c] = (float)(1.0 / ((double)Nb * C));
    }
    kc.n_dis = (float)std::sqrt(std::pow(2.0 * M_PI, (double)D));
    kc.inv_n_dis = 1.0f / kc.n_dis;
    kc.use_det = cfg->use_determinant ? 1 : 0;
    kc.train_gammas = cfg->train_gammas ? 1 : 0;
    kc.only_y_gamma = (cfg->only_y_gamma && cfg->use_yuv && cfg->train_gammas) ? 1 : 0;
    fill_quant_consts(kc, cfg->quantization_mode, cfg->quantize_pis, cfg->train_musx, cfg->bit_depths,
                      cfg->lower_bounds, cfg->upper_bounds);
    for (int ch = 0; ch < SMOE_MAX_CHANNELS; ++ch) kc.sw[ch] = 0.0f;
    for (int ch = 0; ch < C; ++ch) {                   // smoe.py:1006-1009, mean over the Nb window positions of a batch
        const double w = cfg->use_yuv ? ((ch == 0) ? 6.0 / 8.0 : 1.0 / 8.0) : 1.0 / (double)C;
        kc.sw[ch] = (float)(w / (double)Nb);
    }
    kc.inverse_cov = cfg->train_inverse_cov ? 1 : 0;
    kc.radial = cfg->radial_as ? 1 : 0;
    kc.kcount_norm = cfg->kernel_count_as_norm_l1 ? 1 : 0;
    kc.pis_l1_raw = cfg->pis_l1;
    *out = h;
    return SMOE_OK;
}

int smoe_shared_destroy(smoe_shared_handle h) {
    if (!h) return SMOE_OK;
#if !SMOE_HOST_TEST
    (void)hipSetDevice(h->cfg.device);
#endif
    if (h->d_axes) dev_free(h->d_axes);
    if (h->d_probes) dev_free(h->d_probes);
    if (h->d_racc) dev_free(h->d_racc);
    if (h->d_ssim_T) dev_free(h->d_ssim_T);
    if (h->d_qrng) dev_free(h->d_qrng);
    if (h->d_part) dev_free(h->d_part);
    if (h->d_trained) dev_free(h->d_trained);
    if (h->d_batch_epoch) dev_free(h->d_batch_epoch);
    if (h->d_bar) dev_free(h->d_bar);
    delete h;
    return SMOE_OK;
}

int smoe_shared_set_center_grid(smoe_shared_handle h, const float* grid) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_set_center_grid: null handle");
    h->mus_grid = grid;
    return SMOE_OK;
}

int smoe_shared_set_loss_weights(smoe_shared_handle h, const float* loss_w) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_set_loss_weights: null handle");
    h->loss_w = loss_w;
    return SMOE_OK;
}

int smoe_shared_num_batches(smoe_shared_handle h) { return h ? h->NB : fail(SMOE_ERR_INVALID, "null handle"); }
int smoe_shared_list_words(smoe_shared_handle h) { return h ? h->KW : fail(SMOE_ERR_INVALID, "null handle"); }

int smoe_shared_grad_buffer(smoe_shared_handle h, double** dev_ptr, int64_t* count) {
    if (!h || !dev_ptr || !count) return fail(SMOE_ERR_INVALID, "smoe_shared_grad_buffer: null argument");
    *dev_ptr = h->d_racc;
    *count = (int64_t)h->cfg.kernels * h->PK + h->cfg.kernels;
    return SMOE_OK;
}

// opt-in (SMOE_SHARED_ONE_LAUNCH=1): measured SLOWER than the two launches per iteration on MI355X (32.4 vs 27.7 us per
// iteration, 512x512 / 144 kernels; profiles/r03/shared_one_launch.txt) -- what crosses workgroups inside a launch has to bypass
// the per-XCD L2 caches, ~2 us per dependent access, nine of them per iteration; two kernel boundaries are cheaper
static bool shared_one_launch_enabled() {
    const char* v = std::getenv("SMOE_SHARED_ONE_LAUNCH");
    return v && v[0] == '1';
}

// a one-launch fit whose grid barrier timed out raised the abort word and left early: reported by the next call of the handle
static int shared_check_abort(smoe_shared_handle h, const char* who) {
    if (!h->fit_pending) return SMOE_OK;
    h->fit_pending = false;
#if !SMOE_HOST_TEST
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (hipSetDevice(h->cfg.device) != hipSuccess) return SMOE_OK;
    hipError_t e = hipMemcpy(w, h->d_bar, sizeof(w), hipMemcpyDeviceToHost);     // synchronises with the fit
    if (e != hipSuccess) return fail_hip(e, who);
    if (w[2] != 0u) return fail(SMOE_ERR_HIP, std::string(who) + ": the previous smoe_shared_fit launch aborted (grid barrier timed out); its results are invalid");
#endif
    return SMOE_OK;
}

int smoe_shared_forward(smoe_shared_handle h, int32_t first_batch, int32_t num_batches, const float* target,
                        const smoe_params* p, float* recon, int32_t* argmax, float* loss, float* sse,
                        uint32_t* lists, int32_t update_lists, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_forward: null handle");
    int rc = check_range(h, first_batch, num_batches, "smoe_shared_forward");
    if (rc) return rc;
    if (num_batches == 0) return SMOE_OK;
    if (!target || !params_ok(p) || !lists) return fail(SMOE_ERR_INVALID, "smoe_shared_forward: target, params and lists are required");
    rc = shared_check_abort(h, "smoe_shared_forward");
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::SharedArgs a;
    fill_shared_args(h, a);
    a.target = target; a.p = *p; a.lists = lists; a.b0 = first_batch; a.NB = num_batches;
    a.loss = loss; a.sse = sse; a.recon = recon; a.argmax = argmax; a.update_lists = update_lists;
    HIP_TRY(refresh_ranges(h, p, (hipStream_t)stream), "smoe_shared_forward ranges");
    HIP_TRY(smoe::launch_shared_pass(a, h->cfg.dim, h->cfg.channels, false, (hipStream_t)stream), "smoe_shared_forward launch");
    return SMOE_OK;
}

// gather: sum the batches' rows into the gradient buffer right away (what smoe_shared_grad_buffer hands out for the
// all-reduce); smoe_shared_fit leaves it to the step kernel (one launch less per iteration)
static int shared_accumulate_impl(smoe_shared_handle h, int32_t first_batch, int32_t num_batches, const float* target,
                                  const smoe_params* p, float* loss, float* sse, uint32_t* lists, void* stream, bool gather) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_accumulate: null handle");
    int rc = check_range(h, first_batch, num_batches, "smoe_shared_accumulate");
    if (rc) return rc;
    if (num_batches == 0) return SMOE_OK;
    if (!target || !params_ok(p) || !lists) return fail(SMOE_ERR_INVALID, "smoe_shared_accumulate: target, params and lists are required");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::SharedArgs a;
    fill_shared_args(h, a);
    a.target = target; a.p = *p; a.lists = lists; a.b0 = first_batch; a.NB = num_batches;
    a.loss = loss; a.sse = sse; a.recon = nullptr; a.argmax = nullptr; a.update_lists = 1;
    HIP_TRY(refresh_ranges(h, p, (hipStream_t)stream), "smoe_shared_accumulate ranges");
    HIP_TRY(smoe::launch_shared_pass(a, h->cfg.dim, h->cfg.channels, true, (hipStream_t)stream), "smoe_shared_accumulate launch");
    if (gather && h->d_part != nullptr) {
        smoe::SharedGatherArgs g;
        fill_gather_args(h, g);
        HIP_TRY(smoe::launch_shared_gather(g, (hipStream_t)stream), "smoe_shared_accumulate gather");
    }
    return SMOE_OK;
}

static void fill_adam_args(const smoe_shared_context* h, const smoe_params* p, const smoe_adam_state* s, smoe::SharedAdamArgs& a) {
    const smoe_shared_config& c = h->cfg;
    a.p = *p; a.m = s->m; a.v = s->v;
    a.racc = h->d_racc; a.nact = h->d_racc + (size_t)c.kernels * h->PK; a.K = c.kernels;
    a.b1p = s->beta1_power; a.b2p = s->beta2_power; a.beta1 = c.beta1; a.beta2 = c.beta2; a.eps = c.adam_eps;
    a.clip = c.grad_clip;
    a.lr_expert = c.lr_expert; a.lr_pis = c.lr_pis; a.lr_steer = c.lr_steer;
    a.train_pis = c.train_pis; a.train_musx = c.train_musx; a.train_gammas = c.train_gammas; a.use_det = c.use_determinant; a.only_y_gamma = h->kc.only_y_gamma;
    a.kc = h->kc;
    a.reg_pi = c.pis_l1 / (float)(c.start_pis > 0 ? c.start_pis : c.kernels);
    a.reg_u = c.u_l1;
    a.qrng = h->d_qrng;
    a.mus_grid = h->mus_grid;
    fill_gather_args(h, a.gather);
}

static int shared_apply_impl(smoe_shared_handle h, smoe_params* p, smoe_adam_state* s, void* stream, bool gather) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_apply: null handle");
    if (!params_ok(p) || !s || !params_ok(&s->m) || !params_ok(&s->v)) return fail(SMOE_ERR_INVALID, "smoe_shared_apply: params and adam state are required");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    const smoe_shared_config& c = h->cfg;
    smoe::SharedAdamArgs a;
    fill_adam_args(h, p, s, a);
    const bool fused = gather && h->d_part != nullptr && h->kc.qmode != 3;      // the mode-3 step is ONE workgroup over all kernels
    if (gather && h->d_part != nullptr && !fused)
        HIP_TRY(smoe::launch_shared_gather(a.gather, (hipStream_t)stream), "smoe_shared_apply gather");
    if (!fused) a.gather.part = nullptr;
    HIP_TRY(refresh_ranges(h, p, (hipStream_t)stream), "smoe_shared_apply ranges");
    HIP_TRY(smoe::launch_shared_adam(a, c.dim, c.channels, (hipStream_t)stream), "smoe_shared_apply launch");
    s->beta1_power *= c.beta1;
    s->beta2_power *= c.beta2;
    s->step += 1;
    h->epoch += 1u;                                   // the rows of this pass are spent
    if (h->epoch == 0u) h->epoch = 1u;
    return SMOE_OK;
}

int smoe_shared_accumulate(smoe_shared_handle h, int32_t first_batch, int32_t num_batches, const float* target,
                           const smoe_params* p, float* loss, float* sse, uint32_t* lists, void* stream) {
    return shared_accumulate_impl(h, first_batch, num_batches, target, p, loss, sse, lists, stream, true);
}

int smoe_shared_apply(smoe_shared_handle h, smoe_params* p, smoe_adam_state* s, void* stream) {
    return shared_apply_impl(h, p, s, stream, false);
}

int smoe_shared_discard(smoe_shared_handle h, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_discard: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    const size_t nacc = (size_t)h->cfg.kernels * h->PK + h->cfg.kernels;
    (void)nacc;
    HIP_TRY(hipMemsetAsync(h->d_racc, 0, sizeof(double) * nacc, (hipStream_t)stream), "smoe_shared_discard");
    h->epoch += 1u;
    if (h->epoch == 0u) h->epoch = 1u;
    return SMOE_OK;
}

int smoe_shared_fit(smoe_shared_handle h, const float* target, smoe_params* p, smoe_adam_state* s, int32_t n_iters,
                    float* loss_last, float* sse_last, uint32_t* lists, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_fit: null handle");
    if (n_iters < 0) return fail(SMOE_ERR_INVALID, "smoe_shared_fit: negative n_iters");
    if (n_iters == 0) return SMOE_OK;
    {
        int rc = shared_check_abort(h, "smoe_shared_fit");
        if (rc) return rc;
    }
    // SMOE_SHARED_ONE_LAUNCH=1: ONE launch for all n_iters iterations when every batch can keep a workgroup on the device
    // (shared_fit_kernel: the same pass / gather / step code between grid barriers, bit-identical to the loop below)
    if (h->d_part != nullptr && !h->need_ranges && !h->cfg.ssim_opt && h->kc.qmode != 3 && h->num_cus > 0 && shared_one_launch_enabled()) {
        if (!target || !params_ok(p) || !lists || !s || !params_ok(&s->m) || !params_ok(&s->v))
            return fail(SMOE_ERR_INVALID, "smoe_shared_fit: target, params, adam state and lists are required");
        HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
        smoe::SharedFitArgs f;
        fill_shared_args(h, f.pass);
        f.pass.target = target; f.pass.p = *p; f.pass.lists = lists; f.pass.b0 = 0; f.pass.NB = h->NB;
        f.pass.loss = loss_last; f.pass.sse = sse_last; f.pass.recon = nullptr; f.pass.argmax = nullptr; f.pass.update_lists = 1;
        fill_adam_args(h, p, s, f.adam);
        f.n_iters = n_iters;
        f.bar = h->d_bar;
#if !SMOE_HOST_TEST
        hipError_t e = hipMemsetAsync(h->d_bar, 0, sizeof(uint32_t) * h->bar_words, (hipStream_t)stream);
        if (e == hipSuccess) e = smoe::launch_shared_fit(f, h->cfg.dim, h->cfg.channels, h->num_cus, (hipStream_t)stream);
        if (e == hipSuccess) {
            for (int i = 0; i < n_iters; ++i) {
                s->beta1_power *= h->cfg.beta1;
                s->beta2_power *= h->cfg.beta2;
                h->epoch += 1u;
                if (h->epoch == 0u) h->epoch = 1u;
            }
            s->step += n_iters;
            h->fit_pending = true;
#if defined(SMOE_PHASE_CLOCKS) && SMOE_PHASE_CLOCKS
            {   // diagnostic build: ticks per iteration of the four phases, workgroup 0
                uint32_t w[16];
                if (hipMemcpy(w, h->d_bar, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess)
                    std::fprintf(stderr, "shared_fit_kernel ticks/iteration: pass %.0f  barrier %.0f  gather+step %.0f  barrier %.0f\n",
                                 w[8] / (double)n_iters, w[9] / (double)n_iters, w[10] / (double)n_iters, w[11] / (double)n_iters);
            }
#endif
            return SMOE_OK;
        }
        (void)hipGetLastError();          // too many batches for one resident grid (or no cooperative launch): the loop below
#endif
    }
    for (int i = 0; i < n_iters; ++i) {
        int rc = shared_accumulate_impl(h, 0, h->NB, target, p, loss_last, sse_last, lists, stream, false);
        if (rc) return rc;
        rc = shared_apply_impl(h, p, s, stream, true);
        if (rc) return rc;
    }
    return SMOE_OK;
}

int smoe_shared_update_kernel_list(smoe_shared_handle h, int32_t first_batch, int32_t num_batches,
                                   const smoe_params* p, uint32_t* lists, void* stream) {
    if (!h) return fail(SMOE_ERR_INVALID, "smoe_shared_update_kernel_list: null handle");
    int rc = check_range(h, first_batch, num_batches, "smoe_shared_update_kernel_list");
    if (rc) return rc;
    if (num_batches == 0) return SMOE_OK;
    if (!params_ok(p) || !lists) return fail(SMOE_ERR_INVALID, "smoe_shared_update_kernel_list: params and lists are required");
    HIP_TRY(hipSetDevice(h->cfg.device), "hipSetDevice");
    smoe::SharedReadmitArgs a;
    a.p = *p; a.lists = lists; a.probes = h->d_probes + (size_t)first_batch * h->cfg.dim * 3;
    a.NB = num_batches; a.K = h->cfg.kernels; a.KW = h->KW; a.kc = h->kc; a.qrng = h->d_qrng; a.mus_grid = h->mus_grid;
    HIP_TRY(refresh_ranges(h, p, (hipStream_t)stream), "smoe_shared_update_kernel_list ranges");
    HIP_TRY(smoe::launch_shared_readmit(a, h->cfg.dim, (hipStream_t)stream), "smoe_shared_update_kernel_list launch");
    return SMOE_OK;
}

}  // extern "C"
