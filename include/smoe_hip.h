/*
 * smoe_hip.h -- C ABI of libsmoe_hip.so: the MI355X (gfx950) implementation of the
 * per-block Steered-Mixture-of-Experts fit / reconstruction hot path.
 *
 * The reference (roljon/Steered-Mixture-of-Experts, /root/reference) has no FFI or
 * plugin interface: its only seam is Python -> tf.Session.run().  Each entry point
 * below replaces one group of session.run() calls of the reference; the reference
 * lines are cited per function.  A maintainer binds these with ctypes (see
 * INTEGRATION.md); the signatures carry plain pointers and sizes only.
 *
 * Semantics: every image block is an independent model ("one Smoe instance per
 * block"): own [0,1]^d pixel domain, own K kernels, own Adam state.  B blocks are
 * processed by one launch.
 *
 * Memory: every array argument is CALLER-OWNED DEVICE memory (hipMalloc'd or a
 * torch tensor's data_ptr()) on the handle's device unless it says "host".  The
 * library allocates only a small workspace inside the handle.  Calls are
 * asynchronous on the hipStream_t passed as `stream` (NULL = default stream).
 *
 * Errors: every call returns SMOE_OK (0) or a negative smoe_status; no exceptions,
 * no aborts.  smoe_last_error() returns a thread-local message for the last failure.
 *
 * Threading: a handle is bound to one device and is not thread-safe; use one handle
 * per device (one process per GPU).  No global state besides the error string.
 */
#ifndef SMOE_HIP_H
#define SMOE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMOE_ABI_VERSION 2
#define SMOE_MAX_DIM 3
#define SMOE_MAX_CHANNELS 3

typedef enum smoe_status {
    SMOE_OK = 0,
    SMOE_ERR_INVALID = -1,      /* bad argument / shape                                  */
    SMOE_ERR_UNSUPPORTED = -2,  /* (dim, channels, kernels) combination not instantiated */
    SMOE_ERR_HIP = -3,          /* a HIP runtime call failed                             */
    SMOE_ERR_NO_DEVICE = -4     /* no usable gfx950 device                               */
} smoe_status;

/* Hyper-parameters of the model + optimiser.  Defaults a caller should use are the
 * smoe_test.py CLI defaults (smoe_test.py:262-352) with kernel adding off. */
typedef struct smoe_config {
    int32_t abi_version;        /* = SMOE_ABI_VERSION                                                    */
    int32_t device;             /* HIP device ordinal                                                    */
    int32_t dim;                /* d: 2 (image) or 3 (video)              smoe.py:227                   */
    int32_t block_shape[SMOE_MAX_DIM]; /* pixels per block per axis (y, x[, t]); unused axes = 1  smoe.py:231-245 */
    int32_t channels;           /* C                                      smoe.py:542                   */
    int32_t kernels;            /* K kernels per block                    smoe.py:2156-2161             */
    int32_t precision;          /* bits of the pixel lattice (8)          utils.py:126-131              */
    float   margin;             /* epsilon = margin / 2^precision         smoe.py:931                   */
    int32_t use_determinant;    /* Gaussian normalisation by prod diag(A) smoe.py:809-815               */
    int32_t use_yuv;            /* 6/8,1/8,1/8 channel loss weights       smoe.py:933-935               */
    int32_t train_pis;          /* trainable flags                        smoe.py:389-396               */
    int32_t train_gammas;
    int32_t train_musx;
    float   lr_expert;          /* optimizer1 {nu_e, gamma_e, musX}       smoe_test.py:84, smoe.py:1102 */
    float   lr_pis;             /* optimizer2 {pis}                       smoe_test.py:85, smoe.py:1103 */
    float   lr_steer;           /* optimizer3 {A_diagonal, A_corr}        smoe_test.py:86, smoe.py:1104 */
    float   beta1, beta2, adam_eps;   /* tf.train.AdamOptimizer defaults .9/.999/1e-8                    */
    float   grad_clip;          /* clip(grad, +-v) if > 0                 smoe.py:1152-1153             */
    float   pis_l1;             /* pis_l1 * sum(pis) / start_pis          smoe.py:1027                  */
    float   u_l1;               /* u_l1 * sum(diag A)                     smoe.py:1044                  */
    int32_t start_pis;          /* normaliser K0 of the l1 term           smoe.py:264,1025              */
    int32_t only_y_gamma;       /* slopes only for channel 0 (gamma_mask) smoe.py:725-729               */
    int32_t ssim_opt;           /* loss_pixel = 1 - SSIM instead of the margin loss (2-d / 3-d blocks, every axis >= 5
                                   pixels): custom_ssim on SYMMETRIC-padded blocks, 11x11(x11) Gaussian, channel
                                   weights 6/8,1/8,1/8 (yuv) or the mean; loss_w is ignored as in the reference
                                   smoe.py:929,980-1011, ops/image_ops_impl.py:77-233                     */
    /* Fake-quantised variables inside the graph (smoe.py:474-538): forward, gradients and update_kernel_list see
     * tf.quantization.fake_quant_with_min_max_{args,vars} of the variables; Adam keeps updating the raw ones.
     * 5-tuples in the reference's order A, musX, nu_e, pis, gamma_e (smoe_test.py:302-309). */
    int32_t quantization_mode;  /* 0/1: nothing in the graph; 2: fixed ranges lower/upper_bounds; 3: min/max over the
                                   block's kernels with qpis > 0 (A_diagonal and nu_e as x - min)   smoe.py:482-530.
                                   Mode 3 assumes A_corr is zero on and above its diagonal (as the reference keeps it) */
    int32_t quantize_pis;       /* pis through [lower_bounds[3], upper_bounds[3]] (implied by mode >= 2). NOTE: the
                                   reference CLI passes True by default (smoe_test.py:304, smoe.py:474)           */
    int32_t bit_depths[5];
    float   lower_bounds[5];
    float   upper_bounds[5];
    int32_t train_inverse_cov;  /* A symmetric (diag + A_corr + A_corr^T), maha = r^T A r; the determinant factor keeps
                                   prod(diag A).  Reference CONSTRUCTOR default True, CLI default False
                                   (smoe.py:41,734-735,791-793; smoe_test.py:342)                               */
    int32_t radial_as;          /* one steering value per kernel (A = a I): the caller keeps A_diagonal [B,K,d,d] with equal
                                   diagonal entries (and equal Adam slots); their gradient is the trace of dL/dA, A_corr
                                   is not trained                                    smoe.py:349-365,429-434,714-719 */
    int32_t kernel_count_as_norm_l1; /* pis_l1 * sum(pis) / count(qpis > 0) instead of / start_pis     smoe.py:1022-1027 */
} smoe_config;

/* Parameter set in the reference's get_params() layout (smoe.py:1795-1800) with a
 * leading block axis B, fp32, C-contiguous:
 *   pis[B,K]  musX[B,K,d]  A_diagonal[B,K,d,d]  A_corr[B,K,d,d]  gamma_e[B,K,d,C]  nu_e[B,K,C]
 * Only the diagonal of A_diagonal and the strict lower triangle of A_corr are read or
 * written (smoe.py:732-733); the other entries are carried untouched. */
typedef struct smoe_params {
    float* pis;
    float* musX;
    float* A_diagonal;
    float* A_corr;
    float* gamma_e;
    float* nu_e;
} smoe_params;

/* TF1 Adam slots (m, v per variable, same layout as smoe_params) and the running
 * beta powers, which TF keeps per optimizer and multiplies after every apply. */
typedef struct smoe_adam_state {
    smoe_params m;
    smoe_params v;
    float   beta1_power;        /* host, in/out; initialise to beta1 */
    float   beta2_power;        /* host, in/out; initialise to beta2 */
    int64_t step;               /* host, in/out; number of applied steps */
} smoe_adam_state;

typedef struct smoe_context* smoe_handle;

/* Create / destroy a handle.  Replaces Smoe.__init__ -> init_model graph construction
 * (smoe.py:229,313,331-1064): builds the per-block pixel domain linspace(0,1,size) per
 * axis (gen_domain, smoe.py:2395-2426) on the device. */
int smoe_create(smoe_handle* out, const smoe_config* cfg);
int smoe_destroy(smoe_handle h);

/* 1 if the (dim, channels, kernels) combination has a compiled kernel (csrc/smoe_variants.def). */
int smoe_is_supported(int32_t dim, int32_t channels, int32_t kernels);

/* The smallest instantiated kernel count >= `kernels` for (dim, channels), or -1.  The reference accepts any kernel
 * grid (generate_kernel_grid, smoe.py:2146-2163); a caller whose count is not instantiated pads every block to this
 * count with kernels of prior 0: `bool_mask = kernel_list & pis > 0` (smoe.py:480,738) takes them out of the graph,
 * their gradients are 0 and ApplyAdam leaves a variable with zero slots and zero gradient where it is. */
int smoe_padded_kernels(int32_t dim, int32_t channels, int32_t kernels);
/* The same over the triples built with EVERY graph variant (ssim_opt, quantization_mode 2 / 3; the lines marked FULL in
 * csrc/smoe_variants.def): what a caller pads to when its graph needs one of those kernels. */
int smoe_padded_kernels_full(int32_t dim, int32_t channels, int32_t kernels);

/* Copy the device-resident per-pixel coordinates [d][N] (fp32) to a HOST buffer. */
int smoe_get_coords(smoe_handle h, float* host_out);

/* Evaluation pass.  Replaces run_batched(train=False, update_reconstruction=True)
 * (smoe.py:1606-1793; fetches at 1646-1648,1686-1697): for every block computes the
 * gate, the reconstruction and the loss with the block's current active-kernel mask,
 * then replaces the mask by the kernels that have influence (smoe.py:829-836,1763-1766).
 *   target  [B,C,N] fp32 in [0,1] (channel-planar per block)      loss_w [B,N] or NULL (smoe.py:550)
 *   recon   [B,C,N] or NULL  (quantised reconstruction, smoe.py:857,899)
 *   argmax  [B,N] uint8 or NULL (smoe.py:833 mapped to kernel ids as at smoe.py:1706-1716)
 *   gate_w  [B,K,N] or NULL  (masked gate, smoe.py:837, zero rows for pruned kernels)
 *   loss    [B]  loss_op (smoe.py:1051)      sse [B]  sum of squared error (mse_op = sse/(N*C)*(2^p)^2, smoe.py:1053)
 *   active  [B] uint32 bit k = kernel k in kernel_list, in/out; update_active=0 leaves it unchanged */
int smoe_forward(smoe_handle h, int32_t num_blocks, const float* target, const float* loss_w,
                 const smoe_params* p, float* recon, uint8_t* argmax, float* gate_w,
                 float* loss, float* sse, uint32_t* active, int32_t update_active, void* stream);

/* n_iters training iterations.  Replaces the loop body of Smoe.train (smoe.py:1521-1529):
 * run_batched(train=True) = zero accumulators (1613), forward + tf.gradients (1702,1148),
 * kernel-list prune (1763-1766), one ApplyAdam per group (1788,1173-1193), and the
 * per-iteration divergence test (1565-1570, per block).
 *   p, s      in/out           loss_last/sse_last [B]: values of the LAST train pass (may be NULL)
 *   active    [B] in/out       diverged [B] uint32 in/out (non-zero = block frozen), may be NULL
 *   loss0     [B] iteration-0 loss for the blow-up test, or NULL (NaN test only) */
int smoe_fit(smoe_handle h, int32_t num_blocks, const float* target, const float* loss_w,
             smoe_params* p, smoe_adam_state* s, int32_t n_iters,
             float* loss_last, float* sse_last, uint32_t* active, uint32_t* diverged,
             const float* loss0, void* stream);

/* Kernel re-admission.  Replaces update_kernel_list (smoe.py:2287-2365) for a per-block
 * [0,1]^d domain: active |= (pis > 0) & any_probe(maha < 800), probes = {min,max,mid}^d. */
int smoe_update_kernel_list(smoe_handle h, int32_t num_blocks, const smoe_params* p,
                            uint32_t* active, void* stream);

/* Best snapshot.  Replaces checkpoint_best_op (smoe.py:861-896, trigger 1574-1576), per
 * block: where loss[b] < best_loss[b] copy p -> best and loss -> best_loss. */
int smoe_checkpoint_best(smoe_handle h, int32_t num_blocks, const float* loss, float* best_loss,
                         const smoe_params* p, smoe_params* best, void* stream);

/* Block -> scalars.  Replaces the host accumulation of smoe.py:1758-1759,1761:
 * out_dev[0] = sum_b loss[b]*N, out_dev[1] = sum_b sse[b], out_dev[2] = sum_b popcount(active[b]).
 * out_dev: 3 doubles in DEVICE memory (ready for an RCCL all-reduce by the host layer). */
int smoe_reduce_scalars(smoe_handle h, int32_t num_blocks, const float* loss, const float* sse,
                        const uint32_t* active, double* out_dev, void* stream);

/* Name of the kernel variant smoe_fit would launch for num_blocks (diagnostics / profiles). */
const char* smoe_fit_variant(smoe_handle h, int32_t num_blocks);

/* Resident wavefronts per CU the runtime grants smoe_fit's kernel for num_blocks (diagnostics). */
int smoe_fit_occupancy(smoe_handle h, int32_t num_blocks);

/* use_diff_center (smoe.py:390-394,746-747: the trained variable is the OFFSET from the kernel grid, the graph reads
 * fake_quant(offset) + grid).  The engines keep grid + offset in musX; with quantization_mode 2 / 3 they need the grid to
 * quantise the offsets: caller-owned device array [num_blocks, K, d] laid out like musX and indexed like the musX of every
 * later smoe_forward / smoe_fit / smoe_update_kernel_list call; it must stay alive until cleared with NULL or the handle
 * is destroyed.  Without quantization_mode 2 / 3 it is not read. */
int smoe_set_center_grid(smoe_handle h, const float* grid);

/* Force the lanes-per-block tiling (16, 32, 64; 0 = automatic).  128 = the 64-lane kernels with ONE block on both
 * wavefronts of a workgroup in smoe_fit (other graphs and the evaluation run the plain 64-lane kernel).  216 / 416 / 816 =
 * the team tiling of smoe_fit with 2 / 4 / 8 wavefronts per workgroup: four blocks per workgroup on the 16-lane layout, the
 * wavefronts split the pixel rows (csrc/smoe_team.hip.h; the automatic choice for small batches of the plain margin-loss
 * graph; other graphs and the evaluation choose as with 0).  264 = the duo tiling of smoe_fit: one block on two symmetric
 * wavefronts with a single joint reduction and the slot owners' state in registers (csrc/smoe_duo.hip.h).  Tuning / test hook. */
int smoe_set_tiling(smoe_handle h, int32_t lanes_per_block);

/* Partition invariance.  The reference walks ALL blocks of an image in one host loop (smoe.py:1643-1702): a block's
 * result does not depend on how many other blocks the pass holds.  Here the lanes-per-block tiling -- and with it the
 * order in which a block's per-pixel gradient terms are summed -- is chosen from the number of blocks, so that a shard
 * of an image (one of R ranks, smoe_fit called with num_blocks = B / R) would round differently from the whole image.
 * total_blocks > 0: every later call of this handle chooses its kernels as if it held total_blocks blocks (the block
 * count of the WHOLE job the calls are shards of), whatever num_blocks it is given: per-block results are then
 * bit-identical for every split of the job into calls / ranks.  0 (the default) = choose from each call's num_blocks. */
int smoe_set_total_blocks(smoe_handle h, int64_t total_blocks);

/* Pixel sub-sampling (run_batched(train=True, sampling_percentage < 100), smoe.py:1664-1667): the reference feeds a random
 * subset of a block's pixels.  Here the caller passes the subset as loss weights (N / n for the drawn pixels, 0 for the
 * others: the same loss and gradients, `mean` over the n drawn pixels).  on != 0 tells the following smoe_fit calls that
 * their loss_w is such a sample: pixels with weight 0 are "not fed", i.e. they take no part in the influence test that
 * prunes the kernel list either (smoe.py:829,1763-1766).  With on == 0 (default) weight-0 pixels are loss-mask pixels,
 * which the reference does feed (smoe.py:1674-1677) and which therefore do vote. */
int smoe_set_sampling(smoe_handle h, int32_t on);

/* ---------------------------------------------------------------------------------------------
 * Shared-kernel image mode: the reference's whole-image fit.  ONE global set of K
 * kernels over the [0,1]^d image domain; the image is cut into batches (sliding_window,
 * smoe.py:18-35); every batch evaluates the kernels of its kernel list (smoe.py:738-753); the
 * gradients of all batches of a pass are accumulated (smoe.py:1148-1150) and one ApplyAdam step
 * follows (smoe.py:1788).  overlap_of_batches > 0 adds the halo: the window's extra pixels only take
 * part in the kernel-list influence test (their loss is cropped, smoe.py:909-923); window pixels outside
 * the image carry all-zero coordinates (np.pad of the joint domain, smoe.py:21,28).
 *   target [NB,C,Nb] (batch-planar, as the block layout)   params: get_params() layout, leading K
 *   lists  [NB, KW] uint32 bitmaps, KW = smoe_shared_list_words(h) = ceil(K/32)
 * For multi-GPU the host layer shards the BATCHES: every call takes the range
 * [first_batch, first_batch + num_batches) and launch-local buffers; between
 * smoe_shared_accumulate and smoe_shared_apply it all-reduces the buffer returned by
 * smoe_shared_grad_buffer (K*P+K doubles) over the ranks (RCCL sum). */
typedef struct smoe_shared_config {
    int32_t abi_version, device, dim;
    int32_t image_shape[SMOE_MAX_DIM];  /* pixels per axis (y, x[, t]), a multiple of batch_shape (smoe.py:239-241) */
    int32_t batch_shape[SMOE_MAX_DIM];
    int32_t channels, kernels, precision;
    float   margin;
    int32_t use_determinant, use_yuv, train_pis, train_gammas, train_musx;
    float   lr_expert, lr_pis, lr_steer, beta1, beta2, adam_eps, grad_clip, pis_l1, u_l1;
    int32_t start_pis;
    int32_t only_y_gamma;
    int32_t overlap;                    /* overlap_of_batches (smoe.py:244), pixels per side */
    int32_t quantization_mode;          /* as smoe_config; mode 3 ranges are IMAGE-wide min / max  smoe.py:474-530 */
    int32_t quantize_pis;
    int32_t bit_depths[5];
    float   lower_bounds[5];
    float   upper_bounds[5];
    int32_t ssim_opt;                   /* loss_pixel = 1 - SSIM of every batch (2-d, >= 5 pixels per axis)  smoe.py:980-1011 */
    int32_t train_inverse_cov;          /* as smoe_config                                       smoe.py:734-735,791-793 */
    int32_t radial_as;                  /* as smoe_config: A_diagonal [K,d,d] with equal diagonals        smoe.py:714-719 */
    int32_t kernel_count_as_norm_l1;    /* pis_l1 / count(qpis > 0) over the image instead of / start_pis  smoe.py:1022-1027 */
} smoe_shared_config;

typedef struct smoe_shared_context* smoe_shared_handle;

int smoe_shared_create(smoe_shared_handle* out, const smoe_shared_config* cfg);
int smoe_shared_destroy(smoe_shared_handle h);
/* Per-pixel loss weights of the whole image (the graph's loss_weights placeholder fed from Smoe.loss_mask,
 * smoe.py:550,932,1674-1677): caller-owned device array [num_batches][Nb] in batch order, indexed with the GLOBAL
 * batch index by every later forward / accumulate / fit call; it must stay alive until cleared with NULL or the
 * handle is destroyed.  Ignored with ssim_opt, as in the reference. */
int smoe_shared_set_loss_weights(smoe_shared_handle h, const float* loss_w);
/* use_diff_center in the shared-kernel mode: the kernel-grid centres [K, d] (see smoe_set_center_grid). */
int smoe_shared_set_center_grid(smoe_shared_handle h, const float* grid);
int smoe_shared_num_batches(smoe_shared_handle h);
int smoe_shared_list_words(smoe_shared_handle h);

/* run_batched(train=False, update_reconstruction=True) (smoe.py:1606-1793).  recon [nb,C,Nb],
 * argmax [nb,Nb] int32 (global kernel ids), loss/sse [nb]; any output may be NULL. */
int smoe_shared_forward(smoe_shared_handle h, int32_t first_batch, int32_t num_batches, const float* target,
                        const smoe_params* p, float* recon, int32_t* argmax, float* loss, float* sse,
                        uint32_t* lists, int32_t update_lists, void* stream);

/* run_batched(train=True) minus train_op: forward + tf.gradients of the batches, accumulated into
 * the handle's gradient buffer (accum_ops, smoe.py:1150); kernel lists pruned (smoe.py:1763-1766).
 * Summation order: every batch leaves the raw sums of its listed kernels in a row of its own; the buffer is the sum of
 * the rows of ALL batches accumulated since the last step, per kernel in batch order (fp64) -- bit-identical from run to run
 * and for any split of a rank's batches over calls.  (Only if the rows would exceed 4 GB -- num_batches x kernels x ~10..22
 * floats -- the pass falls back to fp64 atomics, whose order is not fixed.) */
int smoe_shared_accumulate(smoe_shared_handle h, int32_t first_batch, int32_t num_batches, const float* target,
                           const smoe_params* p, float* loss, float* sse, uint32_t* lists, void* stream);

/* train_op (smoe.py:1788): one ApplyAdam step per optimizer group on the accumulated gradients; clears
 * the accumulators (zero_op, smoe.py:1613). */
int smoe_shared_apply(smoe_shared_handle h, smoe_params* p, smoe_adam_state* s, void* stream);

/* zero_op alone (smoe.py:1613): drop what the smoe_shared_accumulate calls since the last step have accumulated. */
int smoe_shared_discard(smoe_shared_handle h, void* stream);

/* Device pointer + length (in doubles) of the gradient accumulation buffer. */
int smoe_shared_grad_buffer(smoe_shared_handle h, double** dev_ptr, int64_t* count);

/* n_iters x (accumulate over ALL batches; apply): the single-GPU loop body of Smoe.train.  Two launches per iteration.
 * Environment SMOE_SHARED_ONE_LAUNCH=1: all n_iters iterations in ONE cooperative launch (grid barriers between the pass and
 * the step; bit-identical results; used when every batch fits on the device at once, margin loss, quantization_mode != 3) --
 * measured slower on MI355X, so it is opt-in.  If such a launch had to give up (a grid barrier timed
 * out), the next smoe_shared_fit / smoe_shared_forward call of the handle returns SMOE_ERR_HIP. */
int smoe_shared_fit(smoe_shared_handle h, const float* target, smoe_params* p, smoe_adam_state* s, int32_t n_iters,
                    float* loss_last, float* sse_last, uint32_t* lists, void* stream);

/* update_kernel_list (smoe.py:2287-2365) for the batches of the range. */
int smoe_shared_update_kernel_list(smoe_shared_handle h, int32_t first_batch, int32_t num_batches,
                                   const smoe_params* p, uint32_t* lists, void* stream);

const char* smoe_last_error(void);
int smoe_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SMOE_HIP_H */
