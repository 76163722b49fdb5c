// hostcheck_driver.cpp -- drives the HOST layer of libsmoe_hip (csrc/smoe_capi.hip: argument validation, kernel-variant
// selection, tiling rules, LDS-size arithmetic, error strings) under AddressSanitizer + UndefinedBehaviorSanitizer on a box
// WITHOUT a GPU.  Built by `make -C steered_mixture_of_experts_amd/csrc hostcheck` with -DSMOE_HOST_TEST=1: in that build
// smoe_create / smoe_shared_create skip the device requirement and the device allocations, and every HIP call of the entry
// points (hipSetDevice, the kernel launches) is compiled out, so each entry point runs up to the point where it would
// launch.  GPU sanitizers are not available on the GPU pool; this is the sanitizer coverage of the pointer / size arithmetic
// of the host layer (SURVEY section 5 "race detection / sanitizers").  Test infrastructure: tests/test_host_sanitizers.py.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "smoe_device.h"
#include "smoe_hip.h"

static int g_checks = 0, g_fail = 0;
#define EXPECT(cond)                                                                   \
    do {                                                                               \
        ++g_checks;                                                                    \
        if (!(cond)) { ++g_fail; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static smoe_config base_config(int dim, int ch, int k, int b0, int b1, int b2) {
    smoe_config c;
    std::memset(&c, 0, sizeof c);
    c.abi_version = SMOE_ABI_VERSION;
    c.dim = dim; c.block_shape[0] = b0; c.block_shape[1] = b1; c.block_shape[2] = b2;
    c.channels = ch; c.kernels = k; c.precision = 8; c.margin = 0.5f;
    c.use_determinant = 1; c.train_pis = c.train_gammas = c.train_musx = 1;
    c.lr_expert = 1e-3f; c.lr_pis = 1e-5f; c.lr_steer = 1.0f; c.beta1 = 0.9f; c.beta2 = 0.999f; c.adam_eps = 1e-8f;
    c.start_pis = k;
    const int bits[5] = {20, 18, 6, 10, 10};
    const float lb[5] = {-2500.f, -.3f, -5.f, 0.f, -32.f}, ub[5] = {2500.f, 1.3f, 5.f, 2.f, 32.f};
    for (int i = 0; i < 5; ++i) { c.bit_depths[i] = bits[i]; c.lower_bounds[i] = lb[i]; c.upper_bounds[i] = ub[i]; }
    return c;
}

static void check_create_refusals() {
    smoe_handle h = nullptr;
    smoe_config c = base_config(2, 1, 4, 16, 16, 1);
    EXPECT(smoe_create(nullptr, &c) == SMOE_ERR_INVALID);
    EXPECT(smoe_create(&h, nullptr) == SMOE_ERR_INVALID);
    c.abi_version = 1;  EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID && h == nullptr);  c = base_config(2, 1, 4, 16, 16, 1);
    c.dim = 1;          EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);  c.dim = 4; EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 0, 4, 16, 16, 1);   EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 4, 4, 16, 16, 1);   EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 0, 16, 16, 1);   EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 17, 16, 16, 1);  EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 4, 16, 0, 1);    EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 4, 128, 128, 1); EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);       // > 8192 pixels
    c = base_config(3, 1, 4, 2048, 2048, 2048); EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);   // product overflows int32
    c = base_config(2, 1, 4, 16, 16, 1); c.precision = 0;  EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c.precision = 17; EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 4, 16, 16, 1); c.quantization_mode = 4; EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c.quantization_mode = 2; c.bit_depths[0] = 1;   EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c.bit_depths[0] = 25;                           EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c.bit_depths[0] = 20; c.lower_bounds[1] = 2.0f; EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);
    c = base_config(2, 1, 4, 16, 4, 1); c.ssim_opt = 1; EXPECT(smoe_create(&h, &c) == SMOE_ERR_INVALID);      // axis < 5
    c = base_config(2, 2, 5, 16, 16, 1);   EXPECT(smoe_create(&h, &c) == SMOE_ERR_UNSUPPORTED);
    c = base_config(2, 1, 3, 16, 16, 1); c.ssim_opt = 1; EXPECT(smoe_create(&h, &c) == SMOE_ERR_UNSUPPORTED);  // basic triple
    c = base_config(2, 1, 3, 16, 16, 1); c.quantization_mode = 3; EXPECT(smoe_create(&h, &c) == SMOE_ERR_UNSUPPORTED);
    EXPECT(std::strlen(smoe_last_error()) > 10);
    EXPECT(smoe_is_supported(2, 1, 4) == 1 && smoe_is_supported(2, 2, 4) == 0 && smoe_is_supported(7, 1, 4) == 0);
    EXPECT(smoe_padded_kernels(2, 1, 5) == 6 && smoe_padded_kernels(2, 1, 17) == -1 && smoe_padded_kernels(2, 1, 4) == 4);
    EXPECT(smoe_padded_kernels_full(2, 1, 3) == 4 && smoe_padded_kernels_full(2, 1, 5) == 8 && smoe_padded_kernels_full(3, 1, 2) == -1);
    EXPECT(smoe_abi_version() == SMOE_ABI_VERSION);
    EXPECT(smoe_destroy(nullptr) == SMOE_OK);
}

// every entry point of one handle, up to the launch
static void drive_handle(const smoe_config& c, bool expect_variant) {
    smoe_handle h = nullptr;
    const int rc = smoe_create(&h, &c);
    if (rc != SMOE_OK) { EXPECT((rc == SMOE_ERR_UNSUPPORTED || (c.ssim_opt && rc == SMOE_ERR_INVALID)) && h == nullptr); return; }
    EXPECT(h != nullptr);
    float dummy[64];
    uint32_t udummy[16];
    uint8_t bdummy[16];
    double ddummy[3];
    smoe_params p = {dummy, dummy, dummy, dummy, dummy, dummy};
    smoe_params bad = p; bad.A_corr = nullptr;
    smoe_adam_state st; st.m = p; st.v = p; st.beta1_power = 0.9f; st.beta2_power = 0.999f; st.step = 0;
    static const int tilings[] = {0, 16, 32, 64, 128, 216, 264, 416, 816, 0};
    static const int blocks[] = {1, 3, 4, 5, 1023, 1024, 1025, 1536, 3071, 3072, 8191, 8192, 65536, 1 << 30};
    for (int t : tilings) {
        EXPECT(smoe_set_tiling(h, t) == SMOE_OK);
        for (long long total : {0LL, 1LL, 4050LL, 32400LL, 1LL << 40}) {
            EXPECT(smoe_set_total_blocks(h, total) == SMOE_OK);
            for (int B : blocks) {
                const char* name = smoe_fit_variant(h, B);
                EXPECT(name != nullptr);
                if (expect_variant && t == 0) EXPECT(std::strlen(name) > 4);
                if (total > 0) EXPECT(std::string(name) == std::string(smoe_fit_variant(h, 1)));     // partition invariance of the choice
                (void)smoe_fit_occupancy(h, B);            // no device: an error code, never a crash
                // launches: everything up to the launch itself
                const int f = smoe_fit(h, B, dummy, (B & 1) ? dummy : nullptr, &p, &st, 3, dummy, dummy, udummy, udummy, dummy, nullptr);
                const int e = smoe_forward(h, B, dummy, nullptr, &p, dummy, bdummy, dummy, dummy, dummy, udummy, 1, nullptr);
                EXPECT((f == SMOE_OK || f == SMOE_ERR_UNSUPPORTED) && (e == SMOE_OK || e == SMOE_ERR_UNSUPPORTED));
                if (expect_variant && t == 0) EXPECT(f == SMOE_OK && e == SMOE_OK);
            }
        }
    }
    if (expect_variant) EXPECT(st.step > 0 && st.beta1_power < 0.9f);
    EXPECT(smoe_set_total_blocks(h, -1) == SMOE_ERR_INVALID);
    EXPECT(smoe_set_total_blocks(h, 0) == SMOE_OK);
    for (int t : {1, 8, 17, 48, 100, 256, 316, 817, -16}) EXPECT(smoe_set_tiling(h, t) == SMOE_ERR_INVALID);
    EXPECT(smoe_set_tiling(h, 0) == SMOE_OK);
    // argument checks
    EXPECT(smoe_fit(nullptr, 4, dummy, nullptr, &p, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, -1, dummy, nullptr, &p, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 4, dummy, nullptr, &p, &st, -1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 0, nullptr, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SMOE_OK);
    EXPECT(smoe_fit(h, 4, dummy, nullptr, &p, &st, 0, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_OK);
    EXPECT(smoe_fit(h, 4, nullptr, nullptr, &p, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 4, dummy, nullptr, &bad, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 4, dummy, nullptr, &p, nullptr, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 4, dummy, nullptr, &p, &st, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_forward(h, -3, dummy, nullptr, &p, nullptr, nullptr, nullptr, dummy, dummy, udummy, 1, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_forward(h, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr) == SMOE_OK);
    EXPECT(smoe_forward(h, 2, dummy, nullptr, &bad, nullptr, nullptr, nullptr, dummy, dummy, udummy, 1, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_forward(h, 2, dummy, nullptr, &p, nullptr, nullptr, nullptr, dummy, dummy, nullptr, 1, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_update_kernel_list(h, -1, &p, udummy, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_update_kernel_list(h, 0, nullptr, nullptr, nullptr) == SMOE_OK);
    EXPECT(smoe_update_kernel_list(h, 3, &bad, udummy, nullptr) == SMOE_ERR_INVALID);
    { const int u = smoe_update_kernel_list(h, 3, &p, udummy, nullptr); EXPECT(u == SMOE_OK || (!expect_variant && u == SMOE_ERR_UNSUPPORTED)); }
    EXPECT(smoe_checkpoint_best(h, 3, nullptr, dummy, &p, &p, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_checkpoint_best(h, 3, dummy, dummy, &p, &bad, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_checkpoint_best(h, 3, dummy, dummy, &p, &p, nullptr) == SMOE_OK);
    EXPECT(smoe_reduce_scalars(h, 3, dummy, dummy, udummy, nullptr, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_reduce_scalars(h, 3, dummy, dummy, udummy, ddummy, nullptr) == SMOE_OK);
    EXPECT(smoe_set_center_grid(h, dummy) == SMOE_OK && smoe_set_center_grid(h, nullptr) == SMOE_OK);
    EXPECT(smoe_set_sampling(h, 1) == SMOE_OK && smoe_set_sampling(nullptr, 1) == SMOE_ERR_INVALID);
    EXPECT(smoe_fit(h, 4, dummy, dummy, &p, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr) == (expect_variant ? SMOE_OK : smoe_fit(h, 4, dummy, dummy, &p, &st, 1, nullptr, nullptr, udummy, nullptr, nullptr, nullptr)));
    EXPECT(smoe_set_sampling(h, 0) == SMOE_OK);
    EXPECT(smoe_get_coords(h, nullptr) == SMOE_ERR_INVALID);
    EXPECT(smoe_destroy(h) == SMOE_OK);
}

static void check_block_handles() {
    struct Shape { int d, b0, b1, b2; };
    const Shape shapes[] = {{2, 16, 16, 1}, {2, 32, 32, 1}, {2, 7, 5, 1}, {2, 1, 9, 1}, {2, 64, 128, 1}, {2, 16, 12, 1},
                            {3, 16, 16, 4}, {3, 8, 8, 8}, {3, 12, 10, 3}, {3, 5, 5, 5}, {3, 32, 32, 8}};
    for (const Shape& s : shapes)
        for (int ch = 1; ch <= 3; ++ch)
            for (int k = 1; k <= 16; ++k) {
                if (!smoe_is_supported(s.d, ch, k)) continue;
                smoe_config c = base_config(s.d, ch, k, s.b0, s.b1, s.b2);
                c.use_yuv = (ch == 3);
                // blocks of up to 1 024 pixels always have a kernel whose planes fit the 160 KB of LDS; larger ones may not
                const bool fits = (long)s.b0 * s.b1 * s.b2 <= 1024;
                drive_handle(c, fits);
                c.quantize_pis = 1; c.pis_l1 = 0.1f; c.u_l1 = 0.01f; c.grad_clip = 1e-3f;
                drive_handle(c, fits);
                c.train_inverse_cov = 1;                  drive_handle(c, fits);
                c.train_inverse_cov = 0; c.radial_as = 1; drive_handle(c, fits);
                c.radial_as = 0; c.quantization_mode = 2; drive_handle(c, false);     // FULL triples only
                c.quantization_mode = 3;                  drive_handle(c, false);
                c.quantization_mode = 0; c.ssim_opt = 1;  drive_handle(c, false);     // may not fit in LDS / need >= 5 pixels
            }
}

static void check_variant_tables() {
    int n = 0;
    const smoe::Variant* v = smoe::variants(&n);
    EXPECT(n > 50 && n <= 256);
    for (int i = 0; i < n; ++i) {
        EXPECT(v[i].name != nullptr && std::strlen(v[i].name) > 6);
        EXPECT(v[i].G == 16 || v[i].G == 32 || v[i].G == 64);
        EXPECT((v[i].fit_team != nullptr) == (v[i].G == 16));
        EXPECT((v[i].fit_duo != nullptr) == (v[i].G == 64));
        if (v[i].duo_lds_bytes)
            for (int N : {1, 35, 256, 1024, 8192})
                for (int hl = 0; hl < 2; ++hl) {
                    const size_t db = v[i].duo_lds_bytes(N, true, hl);
                    EXPECT(db == (size_t)-1 || (db > sizeof(float) * (size_t)(v[i].C * N) && db < (size_t)1 << 31 && (db & 15) == 0));
                }
        for (int N : {1, 35, 256, 1000, 1024, 4096, 8192})
            for (int lw = 0; lw < 2; ++lw)
                for (int hq = 0; hq < 2; ++hq) {
                    const size_t b = v[i].lds_bytes(N, lw != 0, hq != 0);
                    EXPECT(b > 0 && b < (size_t)1 << 31 && (b & 3) == 0);
                    EXPECT(b >= sizeof(float) * (size_t)(v[i].C * N * (64 / v[i].G) * v[i].W));     // at least the staged targets
                    if (lw) EXPECT(b >= v[i].lds_bytes(N, false, hq != 0));
                    if (v[i].lds_bytes_ssim) {
                        const size_t s2 = v[i].lds_bytes_ssim(256, lw != 0, 16, 16, 0, hq != 0);
                        const size_t s3 = v[i].lds_bytes_ssim(1024, lw != 0, 16, 16, 4, hq != 0);
                        EXPECT(s2 == (size_t)-1 || s2 > v[i].lds_bytes(256, lw != 0, hq != 0));
                        EXPECT(s3 == (size_t)-1 || s3 > 0);
                    }
                    if (v[i].team_lds_bytes)
                        for (int nw : {2, 4, 8}) {
                            const size_t t = v[i].team_lds_bytes(N, lw != 0, nw);
                            EXPECT(t > sizeof(float) * (size_t)(4 * v[i].C * N) && t < (size_t)1 << 31 && (t & 3) == 0);
                            if (nw > 2) EXPECT(t > v[i].team_lds_bytes(N, lw != 0, nw / 2));
                        }
                }
    }
}

static smoe_shared_config shared_config(int dim, int ch, int k, const int* img, const int* bat) {
    smoe_shared_config c;
    std::memset(&c, 0, sizeof c);
    c.abi_version = SMOE_ABI_VERSION;
    c.dim = dim;
    for (int l = 0; l < 3; ++l) { c.image_shape[l] = (l < dim) ? img[l] : 1; c.batch_shape[l] = (l < dim) ? bat[l] : 1; }
    c.channels = ch; c.kernels = k; c.precision = 8; c.margin = 0.5f; c.use_determinant = 1;
    c.train_pis = c.train_gammas = c.train_musx = 1;
    c.lr_expert = 1e-3f; c.lr_pis = 1e-5f; c.lr_steer = 1.0f; c.beta1 = 0.9f; c.beta2 = 0.999f; c.adam_eps = 1e-8f;
    c.start_pis = k;
    const int bits[5] = {20, 18, 6, 10, 10};
    const float lb[5] = {-2500.f, -.3f, -5.f, 0.f, -32.f}, ub[5] = {2500.f, 1.3f, 5.f, 2.f, 32.f};
    for (int i = 0; i < 5; ++i) { c.bit_depths[i] = bits[i]; c.lower_bounds[i] = lb[i]; c.upper_bounds[i] = ub[i]; }
    return c;
}

static void check_shared() {
    smoe_shared_handle h = nullptr;
    const int img2[3] = {512, 512, 1}, bat2[3] = {32, 32, 1};
    smoe_shared_config c = shared_config(2, 1, 144, img2, bat2);
    EXPECT(smoe_shared_create(nullptr, &c) == SMOE_ERR_INVALID);
    c.abi_version = 0; EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID); c.abi_version = SMOE_ABI_VERSION;
    c.dim = 5;         EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID); c.dim = 2;
    c.channels = 2;    EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_UNSUPPORTED); c.channels = 1;
    c.kernels = 0;     EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID); c.kernels = 9000; EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID);
    c.kernels = 144;
    c.batch_shape[0] = 30; EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID); c.batch_shape[0] = 32;       // does not divide the image
    c.overlap = -1;    EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID); c.overlap = 65; EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_INVALID);
    c.overlap = 0;
    c.batch_shape[0] = c.batch_shape[1] = 64; EXPECT(smoe_shared_create(&h, &c) == SMOE_ERR_UNSUPPORTED);        // 4096 pixels per batch
    c.batch_shape[0] = c.batch_shape[1] = 32;
    c.kernels = 8192; EXPECT(smoe_shared_create(&h, &c) == SMOE_OK || smoe_shared_create(&h, &c) == SMOE_ERR_UNSUPPORTED);
    if (h) { EXPECT(smoe_shared_destroy(h) == SMOE_OK); h = nullptr; }
    for (int dim = 2; dim <= 3; ++dim)
        for (int ch : {1, 3})
            for (int overlap : {0, 2, 3})
                for (int variant = 0; variant < 4; ++variant) {
                    const int img3[3] = {64, 96, 8}, bat3[3] = {16, 16, 4};
                    smoe_shared_config s = (dim == 2) ? shared_config(2, ch, 24, img2, bat2) : shared_config(3, ch, 24, img3, bat3);
                    s.overlap = overlap; s.use_yuv = (ch == 3);
                    if (variant == 1) { s.quantization_mode = 3; s.kernel_count_as_norm_l1 = 1; }
                    if (variant == 2) s.ssim_opt = 1;
                    if (variant == 3) { s.train_inverse_cov = 1; s.quantize_pis = 1; }
                    smoe_shared_handle hs = nullptr;
                    const int rc = smoe_shared_create(&hs, &s);
                    if (rc != SMOE_OK) { EXPECT(rc == SMOE_ERR_UNSUPPORTED || rc == SMOE_ERR_INVALID); EXPECT(hs == nullptr); continue; }
                    const int NB = smoe_shared_num_batches(hs), KW = smoe_shared_list_words(hs);
                    EXPECT(NB == ((dim == 2) ? 256 : 4 * 6 * 2) && KW == 1);
                    float dummy[16]; uint32_t lists[4]; int32_t am[4];
                    smoe_params p = {dummy, dummy, dummy, dummy, dummy, dummy};
                    smoe_adam_state st; st.m = p; st.v = p; st.beta1_power = 0.9f; st.beta2_power = 0.999f; st.step = 0;
                    EXPECT(smoe_shared_forward(hs, 0, NB, dummy, &p, dummy, am, dummy, dummy, lists, 1, nullptr) == SMOE_OK);
                    EXPECT(smoe_shared_forward(hs, NB - 1, 2, dummy, &p, nullptr, nullptr, dummy, dummy, lists, 1, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_forward(hs, -1, 1, dummy, &p, nullptr, nullptr, dummy, dummy, lists, 1, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_forward(hs, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr) == SMOE_OK);
                    EXPECT(smoe_shared_forward(hs, 0, 1, dummy, &p, nullptr, nullptr, dummy, dummy, nullptr, 1, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_accumulate(hs, 3, NB - 3, dummy, &p, dummy, dummy, lists, nullptr) == SMOE_OK);
                    EXPECT(smoe_shared_accumulate(hs, 3, NB, dummy, &p, dummy, dummy, lists, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_discard(hs, nullptr) == SMOE_OK && smoe_shared_discard(nullptr, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_accumulate(hs, 0, 2, dummy, &p, dummy, dummy, lists, nullptr) == SMOE_OK);
                    EXPECT(smoe_shared_apply(hs, &p, &st, nullptr) == SMOE_OK && st.step == 1);
                    EXPECT(smoe_shared_apply(hs, &p, nullptr, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_fit(hs, dummy, &p, &st, 3, dummy, dummy, lists, nullptr) == SMOE_OK && st.step == 4);
                    EXPECT(smoe_shared_fit(hs, dummy, &p, &st, -1, dummy, dummy, lists, nullptr) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_update_kernel_list(hs, 0, NB, &p, lists, nullptr) == SMOE_OK);
                    EXPECT(smoe_shared_update_kernel_list(hs, 1, NB, &p, lists, nullptr) == SMOE_ERR_INVALID);
                    double* gb = nullptr; int64_t cnt = 0;
                    EXPECT(smoe_shared_grad_buffer(hs, &gb, &cnt) == SMOE_OK && cnt == 24 * (1 + dim + dim * (dim + 1) / 2 + ch + dim * ch) + 24);
                    EXPECT(smoe_shared_grad_buffer(hs, nullptr, &cnt) == SMOE_ERR_INVALID);
                    EXPECT(smoe_shared_set_loss_weights(hs, dummy) == SMOE_OK && smoe_shared_set_center_grid(hs, dummy) == SMOE_OK);
                    EXPECT(smoe_shared_destroy(hs) == SMOE_OK);
                }
    EXPECT(smoe_shared_destroy(nullptr) == SMOE_OK);
    EXPECT(smoe_shared_num_batches(nullptr) == SMOE_ERR_INVALID);
}

int main() {
    check_create_refusals();
    check_variant_tables();
    check_block_handles();
    check_shared();
    std::printf("hostcheck: %d checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
