// smoe_device.h -- argument blocks shared by the kernels (smoe_block.hip.h, smoe_kernels.hip, smoe_shared.hip) and the
// C-ABI host layer (smoe_capi.hip).  Internal; the public surface is include/smoe_hip.h.
#ifndef SMOE_DEVICE_H
#define SMOE_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "smoe_hip.h"

namespace smoe {

// Block-independent constants of the forward / loss maths.
struct KernelConsts {
    float tau;          // 0.5 / 2^p                       smoe.py:825
    float epsm;         // margin / 2^p                    smoe.py:931
    float scale;        // 1 / (2^p - 1)                   fake_quant nudged scale, smoe.py:899
    float inv_scale;    // 1 / scale
    float nudged_max;   // min(1, (2^p - 1) * scale): upper clamp of clip_by_value + fake quant
    float cw[SMOE_MAX_CHANNELS];  // per-channel loss weight / N   smoe.py:933-937
    float n_dis;        // sqrt((2 pi)^d)                  smoe.py:812
    float inv_n_dis;    // 1 / n_dis
    int use_det;        // smoe.py:809
    int train_gammas;   // smoe.py:841
    int only_y_gamma;   // gamma_mask: slopes only for channel 0 (smoe.py:725-729)
    float sw[SMOE_MAX_CHANNELS];  // ssim_opt: SSIM channel weight / window count (smoe.py:1006-1009)
    // fake-quantised parameters inside the graph (smoe.py:474-538).  Group order: 0 A, 1 musX, 2 nu_e, 3 pis, 4 gamma_e
    int qmode;          // 0 none, 2 fixed ranges (lower/upper_bounds), 3 min/max over the model's kernels
    int qpis;           // pis through the fixed range (mode >= 2 or quantize_pis)
    int q_musx;         // mode 3 quantises musX only when it is trained (smoe.py:515)
    float q_nmin[5], q_nmax[5], q_scale[5], q_inv[5];   // nudged fixed ranges (TF Nudge(), fp32)
    float q_levels[5];  // 2^bits - 1
    int inverse_cov;    // train_inverse_cov (smoe.py:734-735,791-793): A symmetric, maha = r^T A r
    int radial;         // radial_as (smoe.py:714-719): the steering diagonal is one value per kernel
    int kcount_norm;    // kernel_count_as_norm_l1: pis l1 term normalised by count(qpis > 0) (smoe.py:1022-1027)
    float pis_l1_raw;   // the un-normalised pis_l1 for that
};

struct FitArgs {
    const float* target;      // [B,C,N]
    const float* loss_w;      // [B,N] or null
    smoe_params p, m, v;
    float* loss_out;          // [B] or null
    float* sse_out;           // [B] or null
    uint32_t* active;         // [B]
    uint32_t* diverged;       // [B] or null
    const float* loss0;       // [B] or null
    const float* coords;      // [D][N]
    int B, N, n_iters;
    float b1p, b2p, beta1, beta2, eps;
    float lr_expert, lr_pis, lr_steer;
    int train_pis, train_musx;
    float clip;
    float reg_pi;             // pis_l1 / start_pis
    float reg_u;              // u_l1
    const float* ssim_T;      // ssim_opt: banded tap tables Tr [bh][11], Tc [bw][11], 3-d blocks: Tt [bt][11] (null otherwise)
    int bh, bw, bt;
    int desc_off;             // float offset of the owner-side gradient descriptors in the dynamic LDS (set by the launcher; 0 = off)
    const float* mus_grid;    // use_diff_center with quantization_mode 2 / 3: kernel-grid centres [B,K,D] (null otherwise)
    int pair;                 // few blocks: one block per 2-wavefront workgroup (64-lane tiling, margin loss; fit_kernel PAIR)
    int lds_floats;           // dynamic LDS of the launch in floats (set by the launcher; read by the SMOE_DEBUG carve-up checks)
    uint32_t* dbg;            // SMOE_DEBUG build: device word that collects failed device-side checks (null otherwise)
    int prio_rotate;          // the launch is ONE round (every wavefront resident at once): rotate the wavefronts' priorities (smoe_block.hip.h)
    int lw_is_sample;         // loss_w is a pixel sub-sample (smoe_set_sampling): weight-0 pixels are not fed, they do not vote in the kernel-list prune
    KernelConsts kc;
};

struct FwdArgs {
    const float* target;
    const float* loss_w;
    smoe_params p;
    float* recon;             // [B,C,N] or null
    uint8_t* argmax;          // [B,N] or null
    float* gate_w;            // [B,K,N] or null
    float* loss;              // [B] or null
    float* sse;               // [B] or null
    uint32_t* active;         // [B]
    const float* coords;
    int B, N, update_active;
    int hoist;                // trailing axes whose coordinate is a lane constant (same rule as smoe_fit)
    const float* mus_grid;    // as in FitArgs
    int regt;                 // evaluation kernel: targets in registers, no staged planes in LDS (set by the launcher)
    float reg_pi, reg_u;
    const float* ssim_T;
    int bh, bw, bt;
    int lds_floats;           // as in FitArgs
    uint32_t* dbg;
    KernelConsts kc;
};

struct ReadmitArgs {
    smoe_params p;
    uint32_t* active;
    const float* probes;      // [D][3] = {min, max, mid} per axis
    int B, K;
    int inverse_cov;          // train_inverse_cov: maha = r^T A r
    const float* mus_grid;    // as in FitArgs
};

struct BestArgs {
    const float* loss;
    float* best_loss;
    smoe_params p, best;
    int B, K, D, C;
};

struct ReduceArgs {
    const float* loss;
    const float* sse;
    const uint32_t* active;
    double* out;
    double* partials;         // [reduce_partials_count()] workspace of the handle
    int B, N;
};

struct Variant {
    int D, C, K, G, W;
    const char* name;
    hipError_t (*fit)(const FitArgs&, int hoist_level, hipStream_t);
    hipError_t (*fwd)(const FwdArgs&, hipStream_t);
    size_t (*lds_bytes)(int N, bool has_lw, bool quant_image);
    int (*fit_waves_per_cu)(int N, bool has_lw, int hoist_level, bool pair);   // of the kernel smoe_fit would launch
    hipError_t (*fit_ssim)(const FitArgs&, int hoist_level, hipStream_t);   // ssim_opt (D == 2, G == 64)
    hipError_t (*fwd_ssim)(const FwdArgs&, hipStream_t);
    size_t (*lds_bytes_ssim)(int N, bool has_lw, int bh, int bw, int bt, bool quant_image);
    hipError_t (*readmit_quant)(const ReadmitArgs&, const KernelConsts&, hipStream_t);   // fake-quantised graph
    hipError_t (*fit_quant)(const FitArgs&, int hoist_level, hipStream_t);               // quantization_mode 2 / 3
    hipError_t (*fwd_quant)(const FwdArgs&, hipStream_t);
    hipError_t (*fit_ic)(const FitArgs&, int hoist_level, hipStream_t);                  // train_inverse_cov
    hipError_t (*fwd_ic)(const FwdArgs&, hipStream_t);
    // team tiling (smoe_team.hip.h; entries of the 16-lane variants only): four blocks per workgroup of nw wavefronts
    hipError_t (*fit_team)(const FitArgs&, int hoist_level, int nw, hipStream_t);
    size_t (*team_lds_bytes)(int N, bool has_lw, int nw);
    int (*team_waves_per_cu)(int N, bool has_lw, int nw);
    // duo tiling (smoe_duo.hip.h; entries of the 64-lane variants only): one block on two symmetric wavefronts
    hipError_t (*fit_duo)(const FitArgs&, int hoist_level, hipStream_t);
    size_t (*duo_lds_bytes)(int N, bool has_lw, int hoist_level);      // (size_t)-1: the triple has too many slots
    int (*duo_waves_per_cu)(int N, bool has_lw, int hoist_level);
};

// ---- shared-kernel image mode (smoe_shared.hip) ----------------------------------------------
struct SharedArgs {
    const float* target;      // [nb][C][Nb] of the batches [b0, b0+nb)
    smoe_params p;            // global kernels [K,...]
    const float* axis_coords; // per-axis global coordinate tables, concatenated
    int axis_off[SMOE_MAX_DIM];
    int batch_shape[SMOE_MAX_DIM];
    int grid[SMOE_MAX_DIM];   // batches per axis
    int image_shape[SMOE_MAX_DIM];
    int overlap;              // halo width (pixels per side) used by the influence test only
    uint32_t* lists;          // [nb][KW] kernel-list bitmaps
    int b0, NB, Nb, K, KW;    // NB = batches in this launch
    float* loss;              // [nb] or null
    float* sse;               // [nb] or null
    float* recon;             // [nb][C][Nb] or null
    int32_t* argmax;          // [nb][Nb] or null
    double* racc;             // [K][PK] raw gradient sums (train)
    double* nact;             // [K] number of batches that listed the kernel (train, for the l1 terms)
    int update_lists;
    KernelConsts kc;
    float reg_pi, reg_u;
    const float* loss_w;      // [num_batches][Nb] per-pixel loss weights of the WHOLE image (global batch index) or null
    const float* ssim_T;      // ssim_opt: banded tap tables Tr [bh][11], Tc [bw][11]
    int ssim;                 // 1: loss_pixel = 1 - SSIM of the batch
    int ssim_off;             // float offset of the SSIM planes inside the workgroup's LDS
    const float* qrng;        // image-wide records of shared_ranges_kernel (mode 3 ranges, count of qpis > 0); see SharedRangesArgs
    const float* mus_grid;    // use_diff_center with quantization_mode 2 / 3: kernel-grid centres [K][D] (null otherwise)
    // fixed-order gradient accumulation (train passes): every batch leaves the raw sums of its listed kernels in its own rows
    // part[global batch][kernel][PK] and the set of those kernels in trained[global batch][KW], stamped with the pass number
    // (batch_epoch); SharedGatherArgs sums them per kernel in batch order.  part == null: fp64 atomics into racc / nact.
    float* part;
    uint32_t* trained;
    uint32_t* batch_epoch;
    uint32_t epoch;
};

// Per-kernel sum over the batches of the current pass, in a fixed order (lane = batch mod 64 ascending, then a butterfly
// over the lanes): racc[k][PK] and nact[k] are OVERWRITTEN -- bit-identical from run to run and for every split of the
// batches of a rank over smoe_shared_accumulate calls.
struct SharedGatherArgs {
    const float* part;
    const uint32_t* trained;
    const uint32_t* batch_epoch;
    uint32_t epoch;
    int NB_total, K, KW, PK;
    double* racc;
    double* nact;
};

struct SharedAdamArgs {
    smoe_params p, m, v;
    double* racc;
    double* nact;
    int K;
    float b1p, b2p, beta1, beta2, eps, clip;
    float lr_expert, lr_pis, lr_steer;
    int train_pis, train_musx, train_gammas, use_det, only_y_gamma;
    float reg_pi, reg_u;
    KernelConsts kc;          // fake quant of the variables (quantize_pis, quantization_mode 2 / 3)
    float* qrng;              // records of shared_ranges_kernel for the CURRENT parameters (mode 3, kernel_count_as_norm_l1)
    const float* mus_grid;    // as in SharedArgs
    SharedGatherArgs gather;  // gather.part != null: the step sums the batches' rows itself (single launch per step, smoe_shared_fit)
};

struct SharedReadmitArgs {
    smoe_params p;
    uint32_t* lists;          // [nb][KW]
    const float* probes;      // [nb][D][3]
    int NB, K, KW;
    KernelConsts kc;
    const float* qrng;
    const float* mus_grid;    // as in SharedArgs
};

// Image-wide quantities of the fake-quantised graph, recomputed from the parameters before every launch that reads them:
// record t = 0..4 (A_diagonal, A_corr, musX, nu_e, gamma_e) = 8 floats {nudged min, nudged max, step, 1/step, offset
// added back, zero-range flag, raw min, raw max} of quantization_mode 3 (min / max over the kernels with qpis > 0,
// smoe.py:497-530); qrng[40] = count(qpis > 0) (kernel_count_as_norm_l1, smoe.py:1012,1022-1027).
static constexpr int SHARED_QRNG_FLOATS = 48;
struct SharedRangesArgs {
    smoe_params p;
    float* qrng;
    int K;
    KernelConsts kc;
    const float* mus_grid;    // as in SharedArgs
};

// smoe_shared_fit as one launch (shared_fit_kernel): the pass and step arguments of the first iteration (the kernel advances the
// pass number and the beta powers itself), the iteration count, and the device words of the grid barrier, zeroed by the host
// before the launch: bar[2] abort, bar[8 .. 11] phase clocks of the diagnostic build, bar[16 + w] arrival flag of workgroup w
// (16 + NB rounded up to a multiple of four words)
struct SharedFitArgs {
    SharedArgs pass;
    SharedAdamArgs adam;
    int n_iters;
    uint32_t* bar;
};

size_t shared_lds_bytes(int D, int C, int K, int KW);
size_t shared_ssim_lds_bytes(int C, int Nb, int bh, int bw, int bt);
bool shared_supported(int D, int C, int Nb);
hipError_t launch_shared_pass(const SharedArgs& a, int D, int C, bool train, hipStream_t st);
hipError_t launch_shared_adam(const SharedAdamArgs& a, int D, int C, hipStream_t st);
hipError_t launch_shared_gather(const SharedGatherArgs& g, hipStream_t st);
// hipErrorCooperativeLaunchTooLarge: the batches do not all fit on the device at once (the caller keeps its two launches per
// iteration); num_cus = compute units of the device
hipError_t launch_shared_fit(const SharedFitArgs& f, int D, int C, int num_cus, hipStream_t st);
hipError_t launch_shared_readmit(const SharedReadmitArgs& a, int D, hipStream_t st);
hipError_t launch_shared_ranges(const SharedRangesArgs& a, int D, int C, hipStream_t st);

const Variant* variants(int* count);
hipError_t launch_readmit(const ReadmitArgs& a, int D, hipStream_t st);
hipError_t launch_best(const BestArgs& a, hipStream_t st);
hipError_t launch_reduce(const ReduceArgs& a, hipStream_t st);
int reduce_partials_count();

}  // namespace smoe
#endif
