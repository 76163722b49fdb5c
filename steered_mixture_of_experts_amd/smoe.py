"""``Smoe`` -- host-side mirror of the reference's model facade (reference smoe.py:37-2578)
for the per-block hot path, backed by the HIP kernels in libsmoe_hip.so.

Same entry points, argument names and parameter-dict layout as the reference
(``Smoe(image, kernels_per_dim, ...)``, ``set_optimizer``, ``train``, ``run_batched``,
``get_params`` / ``get_best_params``, ``get_reconstruction`` ...), with block-independent
semantics: ``batch_size`` is the block shape, ``kernels_per_dim`` the kernels per block per
axis, and every block is its own model (own [0,1]^d domain, own K kernels, own Adam state) --
``get_params()[name][b]`` is what the reference's ``Smoe(block_b, ...).get_params()[name]``
returns.  All blocks are fitted by ONE kernel launch per chunk of iterations instead of one
``session.run`` per block per iteration (smoe.py:1643-1702).

Also built: the shared-kernel image mode with batch overlap (``SharedSmoe``), the SSIM loss
(2-d blocks), the parameter quantiser and quantisation-aware fitting (``quantization_mode`` 1-3, ``quantize_pis``),
``train_inverse_cov``, ``radial_as``, ``use_diff_center``, pixel sub-sampling (``sampling_percentage``).  Any kernel grid
is accepted: a kernel count without its own kernel instantiation is padded to the next instantiated one with prior-zero
kernels, which the graph drops (``bool_mask = kernel_list & pis > 0``, smoe.py:480,738).  Not rebuilt (SURVEY section 2
"OUT OF SCOPE"): support vectors, motion models, kernel adding; passing those options raises NotImplementedError
instead of silently doing something else.

Partition invariance: the engine is told the block count of the WHOLE image (``set_total_blocks``), so a rank that
holds a shard of the blocks runs the kernels the whole image would run and every block's result is bit-identical for
any number of ranks -- the reference has one host loop over all blocks (smoe.py:1643-1702).
"""
from __future__ import annotations

import pickle
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import blocks as blk
from . import dist as sdist
from .engine import PARAM_NAMES, EngineConfig


class Adam:
    """Optimizer configuration object standing in for ``tf.train.AdamOptimizer`` (the
    reference only reads ``_lr`` before handing the object to TF, smoe.py:1120-1144)."""

    def __init__(self, learning_rate: float = 0.001, beta1: float = 0.9, beta2: float = 0.999,
                 epsilon: float = 1e-8):
        self._lr = float(learning_rate)
        self._beta1 = float(beta1)
        self._beta2 = float(beta2)
        self._epsilon = float(epsilon)


def _default_engine_factory(cfg: EngineConfig, device):
    from .engine import BlockEngine          # loads libsmoe_hip.so; fails loudly if absent
    return BlockEngine(cfg, device)


def _default_padded_kernels(dim: int, channels: int, kernels: int, need_full: bool = False) -> int:
    """Smallest instantiated kernel count >= kernels (include/smoe_hip.h: smoe_padded_kernels); need_full: among the
    triples built with the SSIM / quantization_mode 2, 3 kernels (smoe_padded_kernels_full)."""
    from . import _lib
    lib = _lib.load()
    kp = int(lib.smoe_padded_kernels_full(dim, channels, kernels) if need_full else lib.smoe_padded_kernels(dim, channels, kernels))
    if kp < 0:
        raise NotImplementedError(f"no per-block kernel is instantiated for {kernels} or more kernels per block with "
                                  f"(dim={dim}, channels={channels}{', every graph variant' if need_full else ''}); "
                                  "add the triple to csrc/smoe_variants.def")
    return kp


def _pad_kernels(p: Dict[str, np.ndarray], kp: int) -> Dict[str, np.ndarray]:
    """Pad every block's parameter set to kp kernels with kernels that are not part of the graph: prior 0 (smoe.py:480,738:
    bool_mask = kernel_list & pis > 0), centre 0.5, steering identity (any finite value: it is never used)."""
    k = p["pis"].shape[1]
    if kp == k:
        return p
    out = {}
    for name, v in p.items():
        pad = np.zeros((v.shape[0], kp - k) + v.shape[2:], dtype=np.float32)
        if name == "musX":
            pad[...] = 0.5
        elif name == "A_diagonal":
            pad[...] = np.eye(v.shape[-1], dtype=np.float32)
        out[name] = np.ascontiguousarray(np.concatenate([v, pad], axis=1))
    return out


def _fake_quant_fixed(x: torch.Tensor, lb: float, ub: float, bits: int) -> torch.Tensor:
    """tf.quantization.fake_quant_with_min_max_args in fp32 (TF Nudge(): the zero point is rounded so that 0 stays
    representable), for host-side bookkeeping only -- the fit itself quantises inside the kernels."""
    f = torch.float32
    levels = torch.tensor(float(2 ** bits - 1), dtype=f)
    mn, mx = torch.tensor(float(lb), dtype=f), torch.tensor(float(ub), dtype=f)
    scale = (mx - mn) / levels
    zp = -mn / scale
    nzp = torch.clamp(torch.sign(zp) * torch.floor(zp.abs() + 0.5), 0.0, float(levels))
    nmin, nmax = (0.0 - nzp) * scale, (levels - nzp) * scale
    cl = torch.minimum(torch.maximum(x.to(f), nmin.to(x.device)), nmax.to(x.device))
    return torch.floor((cl - nmin.to(x.device)) * (1.0 / scale).to(x.device) + 0.5) * scale.to(x.device) + nmin.to(x.device)


def _train_loop(m, num_iter, val_iter, ukl_iter, pis_l1, u_l1, callbacks, chunk_limit=None):
    """The iteration / validation / stop logic of ``Smoe.train`` (smoe.py:1485-1603), shared by both facades.  ``m`` supplies
    ``run_batched`` and four hooks: ``_quantize`` (quantize_params [+ rescaler], smoe.py:1498-1505,1539-1545),
    ``_fit_iterations(n)`` (n training passes), ``_readmit_kernels`` (update_kernel_list, smoe.py:1531-1536) and
    ``_keep_best(loss)`` (best snapshot, smoe.py:1574-1576).  The iterations up to the next validation / kernel-list boundary
    run as ONE engine call (``chunk_limit`` = 1 for a fresh pixel draw per pass)."""
    if m.quantization_mode >= 1:                                      # smoe.py:1498-1499
        m._quantize()
    if m.quantization_mode == 1:                                      # smoe.py:1500-1505
        m.best_qloss, m.best_qmse, _, _ = m.run_batched(
            pis_l1=pis_l1, u_l1=u_l1, train=False, update_reconstruction=True, with_quantized_params=True)
        m.qlosses.append((0, m.best_qloss))
        m.qmses.append((0, m.best_qmse))
    # iteration-0 evaluation (smoe.py:1507-1519)
    m.best_loss, m.best_mse, num_pi, num_sv = m.run_batched(pis_l1=pis_l1, u_l1=u_l1, train=False, update_reconstruction=True)
    m._after_first_evaluation()
    m.losses.append((m.iter, m.best_loss))
    m.mses.append((m.iter, m.best_mse))
    m.num_pis.append((m.iter, num_pi))
    m.num_svs.append((m.iter, num_sv))
    for callback in callbacks:
        callback(m)

    loss_val, mse_val = m.best_loss, m.best_mse
    i = 0
    try:
        while i < num_iter:
            # run up to the next validation / kernel-list boundary in ONE launch
            nxt = min(num_iter, (i // val_iter + 1) * val_iter, (i // ukl_iter + 1) * ukl_iter)
            if chunk_limit:
                nxt = min(nxt, i + chunk_limit)
            n = nxt - i
            m._fit_iterations(n)
            i = nxt
            m.iter += n
            m.valid = False
            validate = i % val_iter == 0
            if i % ukl_iter == 0:                                         # smoe.py:1531-1536
                m._readmit_kernels()
                if not validate:
                    loss_val, mse_val, num_pi, num_sv = m.run_batched(pis_l1=pis_l1, u_l1=u_l1, train=False)
            if validate:                                                  # smoe.py:1538-1594
                if m.quantization_mode >= 1:                              # smoe.py:1539-1540
                    m._quantize()
                if m.quantization_mode == 1:                              # smoe.py:1541-1545,1585-1587
                    qloss_val, qmse_val, _, _ = m.run_batched(
                        pis_l1=pis_l1, u_l1=u_l1, train=False, update_reconstruction=True, with_quantized_params=True)
                    m.qlosses.append((i, qloss_val))
                    m.qmses.append((i, qmse_val))
                loss_val, mse_val, num_pi, num_sv = m.run_batched(
                    pis_l1=pis_l1, u_l1=u_l1, train=False, update_reconstruction=True)
                # global divergence rule at validation cadence (per block it is applied on the device every iteration,
                # smoe.py:1565-1570)
                if np.isnan(loss_val) or (len(m.losses) > 0 and loss_val + 1 > (m.losses[0][1] + 100) * 10):
                    print("stop")
                    break
                m._keep_best(loss_val)
                m.losses.append((m.iter, loss_val))
                if not m.best_mse or mse_val < m.best_mse:
                    m.best_mse = mse_val
                m.mses.append((m.iter, mse_val))
                m.num_pis.append((m.iter, num_pi))
                m.num_svs.append((m.iter, num_sv))
                for callback in callbacks:
                    callback(m)
    except KeyboardInterrupt:
        pass
    m.losses_history.append(m.losses)
    m.mses_history.append(m.mses)
    if m.rank == 0:
        print("end loss/mse: ", loss_val, "/", mse_val, "@iter: ", i)
        print("best loss/mse: ", m.best_loss, "/", m.best_mse)


class Smoe:
    def __init__(self, image, kernels_per_dim=None, train_pis=True, init_params=None, start_batches=1,
                 batch_size=None, train_gammas=True, train_musx=True, use_diff_center=False, radial_as=False,
                 use_determinant=False, normalize_pis=True, quantization_mode=0, bit_depths=None,
                 quantize_pis=False, lower_bounds=None, upper_bounds=None, use_yuv=True, only_y_gamma=False,
                 ssim_opt=False, precision=8, add_kernel_slots=0, iter_offset=0, margin=0.5,
                 overlap_of_batches=0, kernel_count_as_norm_l1=False, train_svs=False, affines=None,
                 train_trafo=False, num_params_model=6, train_inverse_cov=True, init_flag=1,
                 only_rec_from_checkpoint=False, loss_mask=None, device=None, engine_factory=None):
        # -- options outside the hot path: refuse loudly ---------------------------------
        unsupported = {

            "train_svs": train_svs, "train_trafo": train_trafo,
        }
        for name, val in unsupported.items():
            if val:
                raise NotImplementedError(f"Smoe({name}=True) is outside the per-block hot path (SURVEY section 8)")
        if quantization_mode not in (0, 1, 2, 3):
            raise ValueError("quantization_mode must be 0, 1, 2 or 3")                # smoe_test.py:298-301
        if add_kernel_slots:
            raise NotImplementedError("progressive kernel adding changes K over time; not part of the hot path")
        if overlap_of_batches:
            raise NotImplementedError("overlapping batches belong to the shared-kernel mode (SURVEY 8(f-1))")
        if affines is not None:
            raise NotImplementedError("per-frame motion models are out of scope")
        assert kernels_per_dim is not None or init_params is not None, \
            "You need to specify the kernel grid size or give initial parameters."   # smoe.py:249-250

        image = np.asarray(image, dtype=np.float32)
        self.image = image
        self.dim_domain = image.ndim - 1                                # smoe.py:227
        self.num_pixel = int(np.prod(image.shape[:self.dim_domain]))    # smoe.py:228
        self.precision = precision
        self.use_yuv = bool(use_yuv) and image.shape[-1] == 3           # smoe_test.py:41-44
        self.only_y_gamma = bool(only_y_gamma) and self.use_yuv         # smoe_test.py:43-44
        self.ssim_opt = ssim_opt
        self.use_diff_center = use_diff_center
        self.radial_as = radial_as
        self.kernel_count_as_norm_l1 = bool(kernel_count_as_norm_l1)
        self.use_determinant = use_determinant
        self.quantization_mode = quantization_mode
        self.quantize_pis = bool(quantize_pis) or quantization_mode >= 2               # smoe_test.py:36-37, smoe.py:474
        # bit depths / bounds in the order A, musX, nu_e, pis, gamma_e; None -> the CLI defaults (smoe_test.py:302-309)
        self.bit_depths = [20, 18, 6, 10, 10] if bit_depths is None else list(bit_depths)
        self.lower_bounds = [-2500, -.3, -5, 0, -32] if lower_bounds is None else list(lower_bounds)
        self.upper_bounds = [2500, 1.3, 5, 2, 32] if upper_bounds is None else list(upper_bounds)
        self.train_pis, self.train_gammas, self.train_musx = train_pis, train_gammas, train_musx
        self.train_inverse_cov = train_inverse_cov
        self.train_trafo = train_trafo
        self.affines = affines
        self.margin = margin
        self.overlap = overlap_of_batches
        self.add_kernel_slots = 0
        self.loss_mask = loss_mask
        self.qparams = None
        self.rparams = None

        # -- block shape (smoe.py:231-247) -----------------------------------------------
        d = self.dim_domain
        if batch_size is None or batch_size[0] is None:
            # smoe.py:229,243: block shape from the desired number of batches (1 -> the whole image)
            bs = blk.get_batch_shape(start_batches, tuple(image.shape[:d]) + (d + image.shape[-1],))[:-1]
        elif len(batch_size) == d:
            bs = tuple(int(b) for b in batch_size)
        elif len(batch_size) == 1:
            bs = tuple(int(batch_size[0]) for _ in range(d))
        else:
            raise ValueError("Required BatchSize doesn't fit to input dimension")    # smoe.py:237
        self.batch_size_valued = bs
        self.batch_size = bs
        self.grid = blk.grid_shape(image.shape[:d], bs)
        self.num_blocks = int(np.prod(self.grid))
        self.start_batches = self.num_blocks                             # smoe.py:247
        self.padded = blk.padded_shape(image.shape[:d], bs) != tuple(image.shape[:d])
        if self.ssim_opt:
            # loss_pixel = 1 - SSIM (smoe.py:929,980-1011).  The reference's SSIM branch ignores the per-pixel loss
            # weights, so neither a loss mask nor the edge-replicated padding of ragged images can be honoured.
            # 3-d blocks: the 11x11x11 window (smoe.py:999-1003), the time axis padded by 5 like the others
            if min(bs) < 5:
                raise ValueError("ssim_opt pads every block SYMMETRIC by 5: blocks need at least 5 pixels per axis")
            if loss_mask is not None or self.padded:
                raise NotImplementedError("ssim_opt ignores loss weights (as the reference does): no loss_mask, and "
                                          "the image must be a multiple of the block shape")

        # -- shard the independent blocks over ranks ------------------------
        self.rank, self.world_size = sdist.world()
        self.lo, self.hi = sdist.shard_range(self.num_blocks, self.rank, self.world_size)
        all_blocks, valid = blk.image_to_blocks(image, bs)
        blocks_local = all_blocks[self.lo:self.hi]
        self.B = blocks_local.shape[0]
        N = int(np.prod(bs))
        self.N = N
        C = image.shape[-1]
        self.channels = C

        # -- initial parameters (smoe.py:252-262) ----------------------------------------
        if init_params:
            p0 = {k: np.asarray(init_params[k], dtype=np.float32) for k in PARAM_NAMES}
            if radial_as and p0["A_diagonal"].ndim == p0["pis"].ndim:        # the reference's (K,) variable (smoe.py:429-433)
                p0["A_diagonal"] = p0["A_diagonal"][..., None, None] * np.eye(d, dtype=np.float32)
            if p0["pis"].ndim == 1:                       # a single block's dict: broadcast
                p0 = {k: np.broadcast_to(v, (self.num_blocks,) + v.shape).copy() for k, v in p0.items()}
            if p0["pis"].shape[0] != self.num_blocks:
                raise ValueError("init_params must carry one parameter set per block")
            p0 = {k: np.ascontiguousarray(v[self.lo:self.hi]) for k, v in p0.items()}
            K = p0["pis"].shape[1]
            self.musX_init = p0["musX"]
        else:
            kpd = list(kernels_per_dim)
            if len(kpd) == 1:
                kpd = kpd * d
            p0 = blk.init_block_params(blocks_local, kpd, normalize_pis, train_inverse_cov)
            K = p0["pis"].shape[1]
            self.musX_init = blk.gen_domain_grid(kpd, d)
        if radial_as:
            # smoe.py:429-434,714-719: ONE steering value per kernel, A_init[:, 0, 0], tiled over the diagonal; A_corr is
            # not trainable.  The engine keeps A_diagonal (B,K,d,d) with equal diagonal entries and ties their gradient.
            a0 = p0["A_diagonal"][:, :, 0, 0]
            p0["A_diagonal"] = np.ascontiguousarray(a0[..., None, None] * np.eye(d, dtype=np.float32))
        self.kernels = K
        # kernel count of the engine: K itself when a kernel is instantiated for it, else the next instantiated count
        pk = getattr(engine_factory, "padded_kernels", None) if engine_factory is not None else _default_padded_kernels
        need_full = bool(ssim_opt) or int(quantization_mode) >= 2      # graphs only the FULL triples are built for
        if pk is None:
            self._kp = K
        elif engine_factory is None:
            self._kp = int(pk(d, C, K, need_full))
        else:
            self._kp = int(pk(d, C, K))
        p0 = _pad_kernels(p0, self._kp)
        # use_diff_center (smoe.py:390-394,746-747): the trained variable is the OFFSET from the kernel
        # grid (initialised to zero); the engine works on grid + offset, the getters subtract the grid.
        self._mus_grid = None
        if use_diff_center:
            self._mus_grid = np.ascontiguousarray(p0["musX"]).copy()
        self.start_pis = K                                               # smoe.py:264, per block
        self.kernel_count = K * self.num_blocks

        # -- device state ---------------------------------------------------------------
        self._engine_factory = engine_factory or _default_engine_factory
        self._device = device
        self._engine = None
        self._engine_key = None
        self.optimizer1 = self.optimizer2 = self.optimizer3 = None
        self.grad_clip_value_abs = None
        self._make_engine(pis_l1=0.0, u_l1=0.0)
        dev = self._engine.device
        self._target = torch.from_numpy(blk.to_planar(blocks_local)).to(dev)
        lw = None
        if self.padded:
            lw = valid[self.lo:self.hi]
        if loss_mask is not None:                                        # smoe.py:1674-1677
            lm, _ = blk.image_to_blocks(np.asarray(loss_mask, dtype=np.float32)[..., None], bs)
            lm = lm.reshape(self.num_blocks, N)[self.lo:self.hi]
            lw = lm if lw is None else lw * lm
        self._use_loss_mask_default = loss_mask is not None
        self._valid = None if not self.padded else torch.from_numpy(np.ascontiguousarray(valid[self.lo:self.hi])).to(dev)
        self._loss_w = None if lw is None else torch.from_numpy(np.ascontiguousarray(lw, dtype=np.float32)).to(dev)
        self._params = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in p0.items()}
        self._best = {k: v.clone() for k, v in self._params.items()}      # smoe.py:861-866
        self._state = self._engine.new_adam_state(self._params)
        self._active = torch.full((self.B,), (1 << K) - 1, dtype=torch.int32, device=dev)   # smoe.py:315 (padding kernels unlisted)
        self._diverged = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self._loss0 = None
        self._best_loss_blocks = None
        # per-pixel sampling probabilities of the sub-sampled passes: uniform until the first reconstruction pass replaces
        # them by the error-proportional ones (smoe.py:270-272,906-907,1768-1769)
        self._sampl_prob = torch.full((self.B, N), 1.0 / N, dtype=torch.float32, device=dev)

        # -- histories (smoe.py:183-199) -------------------------------------------------
        self.losses, self.mses, self.num_pis, self.num_svs = [], [], [], []
        self.qlosses, self.qmses = [], []
        self.losses_history, self.mses_history = [], []
        self.best_loss = None
        self.best_mse = []
        self.iter = iter_offset
        self.valid = False
        self._images = None                  # reconstruction_image / weight_matrix_argmax / weight_matrix (lazy, see _stitch)
        self.qvalid = False
        self._qimages = None                 # their with_quantized_params counterparts
        self.best_qloss = None
        self.best_qmse = []

    # ------------------------------------------------------------------------------------
    def _make_engine(self, pis_l1: float, u_l1: float):
        o1, o2, o3 = self.optimizer1, self.optimizer2, self.optimizer3
        cfg = EngineConfig(
            block_shape=self.batch_size_valued, channels=self.image.shape[-1], kernels=self._kp,
            precision=self.precision, margin=self.margin, use_determinant=bool(self.use_determinant),
            use_yuv=bool(self.use_yuv), train_pis=bool(self.train_pis), train_gammas=bool(self.train_gammas),
            train_musx=bool(self.train_musx),
            lr_expert=o1._lr if o1 else 0.0, lr_pis=o2._lr if o2 else 0.0, lr_steer=o3._lr if o3 else 0.0,
            beta1=o1._beta1 if o1 else 0.9, beta2=o1._beta2 if o1 else 0.999,
            adam_eps=o1._epsilon if o1 else 1e-8,
            grad_clip=float(self.grad_clip_value_abs or 0.0), pis_l1=float(pis_l1), u_l1=float(u_l1),
            start_pis=self.kernels, only_y_gamma=bool(self.only_y_gamma), ssim_opt=bool(self.ssim_opt),
            quantization_mode=int(self.quantization_mode), quantize_pis=bool(self.quantize_pis),
            bit_depths=tuple(self.bit_depths), lower_bounds=tuple(self.lower_bounds),
            upper_bounds=tuple(self.upper_bounds), train_inverse_cov=bool(self.train_inverse_cov),
            radial_as=bool(self.radial_as), kernel_count_as_norm_l1=self.kernel_count_as_norm_l1)
        key = tuple(sorted(cfg.__dict__.items(), key=lambda kv: kv[0]))
        key = repr(key)
        if key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = self._engine_factory(cfg, self._device)
            self._engine_key = key
            # the kernels (= the summation order inside a block) are chosen for the whole image, not for this rank's shard
            if hasattr(self._engine, "set_total_blocks"):
                self._engine.set_total_blocks(self.num_blocks)
            self._hand_over_center_grid()

    def _hand_over_center_grid(self):
        """use_diff_center with quantization_mode 2 / 3: the graph quantises the OFFSETS (smoe.py:746-747), so the engine
        needs the kernel grid its ``musX`` (= grid + offset) is relative to."""
        if getattr(self, "_mus_grid", None) is None or int(self.quantization_mode) < 2:
            return
        self._mus_grid_dev = torch.from_numpy(np.ascontiguousarray(self._mus_grid, dtype=np.float32)).to(self._engine.device)
        self._engine.set_center_grid(self._mus_grid_dev)

    @property
    def kernel_list_per_batch(self) -> List[np.ndarray]:
        """Per-block boolean kernel lists (smoe.py:315,1763-1766), local blocks only."""
        bits = self._active.cpu().numpy().view(np.uint32)
        K = self.kernels
        mask = ((bits[:, None] >> np.arange(K, dtype=np.uint32)[None, :]) & 1).astype(bool)
        return [m for m in mask]

    # -- optimizer (smoe.py:1079-1204) ----------------------------------------------------
    def set_optimizer(self, optimizer1, optimizer2=None, optimizer3=None, optimizer4=None, optimizer5=None,
                      grad_clip_value_abs=None):
        self.optimizer1 = optimizer1
        self.optimizer2 = optimizer1 if optimizer2 is None else optimizer2
        self.optimizer3 = optimizer1 if optimizer3 is None else optimizer3
        self.grad_clip_value_abs = grad_clip_value_abs
        # a fresh set of optimizers means fresh slots and beta powers, as in TF
        self._make_engine(0.0, 0.0)
        self._state = self._engine.new_adam_state(self._params)

    # -- passes ----------------------------------------------------------------------------
    def _global(self, loss, sse, valid_only=False):
        s = self._engine.reduce_scalars(loss, sse, None)
        pis = self._params["pis"]
        if self.quantize_pis:                                        # num_pi_op counts pis_mask = qpis > 0 (smoe.py:480,1012)
            pis = _fake_quant_fixed(pis, self.lower_bounds[3], self.upper_bounds[3], self.bit_depths[3])
        npi = (pis > 0).sum().to(torch.float64)
        s[2] = npi
        sdist.allreduce_sum_(s)
        s = s.cpu().numpy()
        total_px = float(self.num_blocks) * self.N
        loss_val = s[0] / total_px                                   # smoe.py:1758
        # smoe.py:1053,1759.  Ragged images (the reference raises, smoe.py:239-241): passes with a reconstruction count the
        # image's own pixels (run_batched); training passes report the SSE over the padded tiling.
        mse_px = float(self.num_pixel) if valid_only else total_px
        mse_val = s[1] / (mse_px * self.channels) * (2 ** self.precision) ** 2
        return float(loss_val), float(mse_val), int(s[2])

    def run_batched(self, pis_l1=0, u_l1=0, sv_l1_sub_l2=0, train=True, update_reconstruction=False,
                    with_quantized_params=False, sampling_percentage=100, with_inc=False, train_inc=False,
                    thr_sv=None, use_loss_mask=False):
        """One pass over every block (smoe.py:1606-1793).  Returns (loss, mse, num_pi, num_sv)."""
        if with_inc or train_inc:
            raise NotImplementedError("kernel adding is out of scope")
        self.valid = False
        self._make_engine(pis_l1, u_l1)
        eng = self._engine
        sub_w = None
        if train and not self.ssim_opt and sampling_percentage < 100:       # smoe.py:1664-1667
            sub_w = self._sample_pixels(sampling_percentage)
        if with_quantized_params:
            # smoe.py:1688-1689: the rescaled parameters are fed over the masked-parameter tensors;
            # the kernel lists are not touched (smoe.py:1763)
            assert self.rparams is not None, "quantize_params + rescaler first (smoe.py:1499-1501)"
            self.qvalid = False
            dev = eng.device
            rp = {k: np.ascontiguousarray(self.rparams[k][self.lo:self.hi], dtype=np.float32) for k in PARAM_NAMES}
            rp = {k: torch.from_numpy(v).to(dev) for k, v in _pad_kernels(rp, self._kp).items()}
            out = eng.forward(self._target, rp, self._active, loss_w=self._loss_w,
                              want_recon=update_reconstruction, want_argmax=update_reconstruction,
                              want_gate=update_reconstruction, update_active=False)
            if update_reconstruction:
                self._stitch(out, quantised=True)
            loss_val, mse_val, num_pi = self._global(out["loss"], out["sse"])
            return loss_val, mse_val, num_pi, 0
        if train:
            assert self.optimizer1 is not None, "no optimizer found, you have to specify one!"
            loss = torch.empty((self.B,), dtype=torch.float32, device=eng.device)
            sse = torch.empty((self.B,), dtype=torch.float32, device=eng.device)
            eng.fit(self._target, self._params, self._state, self._active, 1,
                    loss_w=self._loss_w if sub_w is None else sub_w,
                    diverged=self._diverged, loss0=self._loss0, loss_out=loss, sse_out=sse,
                    **({} if sub_w is None else {"loss_w_is_sample": True}))
        else:
            out = eng.forward(self._target, self._params, self._active, loss_w=self._loss_w,
                              want_recon=update_reconstruction, want_argmax=update_reconstruction,
                              want_gate=update_reconstruction)
            loss, sse = out["loss"], out["sse"]
            if update_reconstruction and self._valid is not None:
                # ragged image: the kernels' SSE runs over the edge-replicated padding too; with the reconstruction at
                # hand the reported MSE counts the image's own pixels only (equal to get_psnr() on the cropped image)
                sse = (((out["recon"] - self._target) ** 2) * self._valid[:, None, :]).sum(dim=(1, 2))
            if update_reconstruction:
                self._stitch(out)
                # smoe.py:906-907,1768-1769: per-pixel sampling probabilities of the next sub-sampled passes
                err = ((out["recon"] - self._target) ** 2).mean(dim=1)                  # (B, N)
                self._sampl_prob = err / err.sum(dim=1, keepdim=True).clamp_min(1e-30)
        loss_val, mse_val, num_pi = self._global(loss, sse, valid_only=(not train) and update_reconstruction and self._valid is not None)
        self._last_block_loss = loss
        return loss_val, mse_val, num_pi, 0

    def _sample_pixels(self, sampling_percentage):
        """Pixel sub-sampling (smoe.py:1664-1667): every block trains on round(N * p / 100) pixels drawn without
        replacement with the error-proportional probabilities of the last reconstruction pass (smoe.py:906-907; uniform
        before the first one, smoe.py:270-272).  The
        reference feeds only the drawn pixels; here they get the loss weight N / n (the others 0), which gives the same
        loss and the same gradients -- `mean` over the n drawn pixels.  Draws: exponential-race keys (the successive-
        sampling law of numpy's `choice(replace=False, p=...)`, another random stream).  The engine is told that the weights are
        a sample (``loss_w_is_sample``): pixels that were not drawn do not vote in the kernel-list prune, as in the reference,
        which never sees them.  With a loss mask on top, drawn pixels the mask zeroes do not vote either (the reference
        feeds them with weight 0: they would)."""
        B, N = self._sampl_prob.shape
        n = max(1, int(round(N * sampling_percentage / 100.0)))
        gen = getattr(self, "_sample_gen", None)
        if gen is None:
            gen = self._sample_gen = torch.Generator(device=self._sampl_prob.device)
            gen.manual_seed(20260000 + self.rank)
        u = torch.rand((B, N), generator=gen, device=self._sampl_prob.device, dtype=torch.float32).clamp_min(1e-30)
        keys = -torch.log(u) / self._sampl_prob.clamp_min(1e-12)          # smallest keys win
        idx = torch.topk(keys, n, dim=1, largest=False).indices
        w = torch.zeros((B, N), dtype=torch.float32, device=self._sampl_prob.device)
        w.scatter_(1, idx, float(N) / float(n))
        if self._loss_w is not None:
            w = w * self._loss_w
        return w.contiguous()

    def _assemble_one(self, out, name):
        """One full-image product (reconstruction / argmax / gate) from a pass's device outputs (smoe.py:1719-1783)."""
        bs, d = self.batch_size_valued, self.dim_domain
        if name == "image":
            recon = blk.from_planar(out["recon"].cpu().numpy(), bs)                   # (B,*bs,C)
            recon = sdist.allgather_blocks(recon, self.num_blocks)
            return blk.blocks_to_image(recon, self.image.shape[:d], bs)
        if name == "argmax":
            am = out["argmax"].cpu().numpy().astype(np.int64).reshape((self.B,) + bs)
            am = am + (np.arange(self.lo, self.hi, dtype=np.int64) * self.kernels).reshape((-1,) + (1,) * d)
            am = sdist.allgather_blocks(am, self.num_blocks)
            return blk.blocks_to_image(am[..., None], self.image.shape[:d], bs)[..., 0]
        gate = out["gate_w"].cpu().numpy().reshape((self.B, self._kp) + bs)[:, :self.kernels]
        return sdist.allgather_blocks(gate, self.num_blocks)                          # gate: (B,K,*bs)

    def _assemble(self, out):
        return tuple(self._assemble_one(out, n) for n in ("image", "argmax", "gate"))

    def _stitch(self, out, quantised=False):
        """Keep the products of an update_reconstruction pass.  Single process: the device tensors are kept and the
        host copy + stitching (tens of ms for large images) happens per product when an attribute / getter asks; several ranks:
        assembled right away, because the gather is a collective every rank takes part in."""
        slot = "_qimages" if quantised else "_images"
        if self.world_size == 1:
            setattr(self, slot, {"pending": out})
        else:
            setattr(self, slot, dict(zip(("image", "argmax", "gate"), self._assemble(out))))
        if quantised:
            self.qvalid = True
        else:
            self.valid = True

    def _image_product(self, slot, name):
        st = getattr(self, slot)
        if st is None:
            return None
        if name not in st:                 # single process: each product is copied to the host and stitched when first asked for
            st[name] = self._assemble_one(st["pending"], name)
            if all(n in st for n in ("image", "argmax", "gate")):
                del st["pending"]
        return st[name]

    reconstruction_image = property(lambda self: self._image_product("_images", "image"))
    weight_matrix_argmax = property(lambda self: self._image_product("_images", "argmax"))
    weight_matrix = property(lambda self: self._image_product("_images", "gate"))
    qreconstruction_image = property(lambda self: self._image_product("_qimages", "image"))
    qweight_matrix_argmax = property(lambda self: self._image_product("_qimages", "argmax"))
    qweight_matrix = property(lambda self: self._image_product("_qimages", "gate"))

    # -- training loop (smoe.py:1485-1603) -------------------------------------------------
    def train(self, num_iter, val_iter=100, ukl_iter=None, optimizer1=None, optimizer2=None, optimizer3=None,
              grad_clip_value_abs=None, pis_l1=0, u_l1=0, sv_l1_sub_l2=0, sampling_percentage=100,
              callbacks=(), with_inc=False, train_inc=False, train_orig=True, use_loss_mask=False):
        if ukl_iter is None:
            ukl_iter = val_iter
        if optimizer1:
            self.set_optimizer(optimizer1, optimizer2, optimizer3, grad_clip_value_abs=grad_clip_value_abs)
        assert self.optimizer1 is not None, "no optimizer found, you have to specify one!"
        if with_inc or train_inc or not train_orig:
            raise NotImplementedError("kernel-adding options are outside the hot path")
        self._train_sampling = sampling_percentage if (sampling_percentage < 100 and not self.ssim_opt) else None
        self._make_engine(pis_l1, u_l1)
        _train_loop(self, num_iter, val_iter, ukl_iter, pis_l1, u_l1, callbacks,
                    chunk_limit=1 if self._train_sampling is not None else None)   # a fresh pixel draw per pass (smoe.py:1664-1667)

    # hooks of _train_loop
    def _quantize(self):
        from .quantizer import quantize_params, rescaler
        self.qparams = quantize_params(self, self.get_params())
        if self.quantization_mode == 1:
            self.rparams = rescaler(self, self.qparams)

    def _after_first_evaluation(self):
        if self._loss0 is None:
            self._loss0 = self._last_block_loss.clone()
        if self._best_loss_blocks is None:
            self._best_loss_blocks = self._last_block_loss.clone()
            for k in PARAM_NAMES:
                self._best[k].copy_(self._params[k])

    def _fit_iterations(self, n):
        sp = getattr(self, "_train_sampling", None)
        self._engine.fit(self._target, self._params, self._state, self._active, n,
                         loss_w=self._sample_pixels(sp) if sp is not None else self._loss_w,
                         diverged=self._diverged, loss0=self._loss0,
                         **({} if sp is None else {"loss_w_is_sample": True}))

    def _readmit_kernels(self):
        self._engine.update_kernel_list(self._params, self._active)

    def _keep_best(self, loss_val):
        self._engine.checkpoint_best(self._last_block_loss, self._best_loss_blocks, self._params, self._best)
        if not self.best_loss or loss_val < self.best_loss:
            self.best_loss = loss_val

    # -- getters (smoe.py:1795-1888) ------------------------------------------------------------
    def _gather_params(self, p: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
        out = {k: v.cpu().numpy().copy() for k, v in p.items()}
        if self._mus_grid is not None:                     # use_diff_center: report the trained offsets
            out["musX"] = out["musX"] - self._mus_grid
        out = {k: np.ascontiguousarray(v[:, :self.kernels]) for k, v in out.items()}     # without the padding kernels
        if self.radial_as:                                 # the reference's variable is (K,) per model
            out["A_diagonal"] = np.ascontiguousarray(out["A_diagonal"][:, :, 0, 0])
        return {k: sdist.allgather_blocks(v, self.num_blocks) for k, v in out.items()}

    def get_params(self):
        return self._gather_params(self._params)

    def get_best_params(self):
        return self._gather_params(self._best)

    def get_reconstruction(self):
        if not self.valid:
            self.run_batched(train=False, update_reconstruction=True)
        return self.reconstruction_image

    def get_qreconstruction(self):
        if not self.qvalid:
            self.run_batched(train=False, update_reconstruction=True, with_quantized_params=True)
        return self.qreconstruction_image

    def get_qlosses(self):
        return self.qlosses

    def get_qmses(self):
        return self.qmses

    def get_weight_matrix_argmax(self):
        if not self.valid:
            self.run_batched(train=False, update_reconstruction=True)
        return self.weight_matrix_argmax

    def get_weight_matrix(self):
        if not self.valid:
            self.run_batched(train=False, update_reconstruction=True)
        return self.weight_matrix

    def get_psnr(self) -> float:
        """PSNR of the current reconstruction over the valid pixels (plotter.py:14-15)."""
        rec = self.get_reconstruction()
        mse = float(np.mean((rec.astype(np.float64) - self.image.astype(np.float64)) ** 2))
        return float(-10.0 * np.log10(mse))

    def get_losses(self):
        return self.losses

    def get_mses(self):
        return self.mses

    def get_num_pis(self):
        return self.num_pis

    def get_num_svs(self):
        return self.num_svs

    def get_best_loss(self):
        return self.best_loss

    def get_best_mse(self):
        return self.best_mse

    def get_losses_history(self):
        return self.losses_history

    def get_mses_history(self):
        return self.mses_history

    def get_iter(self):
        return self.iter

    def get_original_image(self):
        return np.squeeze(self.image)

    # -- checkpoint / restore (replaces the tf.train.Saver of smoe.py:1066-1077) ---------------
    def checkpoint(self, path):
        st = {"params": {k: v.cpu().numpy().copy() for k, v in self._params.items()},
              "best": {k: v.cpu().numpy().copy() for k, v in self._best.items()},
              "m": {k: v.cpu().numpy().copy() for k, v in self._state.m.items()},
              "v": {k: v.cpu().numpy().copy() for k, v in self._state.v.items()},
              "beta_pow": (float(self._state.c.beta1_power), float(self._state.c.beta2_power)),
              "step": int(self._state.c.step), "active": self._active.cpu().numpy(),
              "diverged": self._diverged.cpu().numpy(), "iter": self.iter, "losses": self.losses,
              "mses": self.mses, "num_pis": self.num_pis, "shard": (self.lo, self.hi, self.num_blocks),
              # what train() needs to continue exactly where it stopped: the per-block best losses behind the best
              # snapshot, the iteration-0 losses of the per-block stop rule (smoe.py:1565-1570) and the best scalars
              "best_loss_blocks": None if self._best_loss_blocks is None else self._best_loss_blocks.cpu().numpy(),
              "loss0": None if self._loss0 is None else self._loss0.cpu().numpy(),
              "best_loss": self.best_loss, "best_mse": self.best_mse}
        with open(path if self.world_size == 1 else f"{path}.rank{self.rank}", "wb") as fd:
            pickle.dump(st, fd)
        return path

    def restore(self, path):
        with open(path if self.world_size == 1 else f"{path}.rank{self.rank}", "rb") as fd:
            st = pickle.load(fd)
        assert tuple(st["shard"]) == (self.lo, self.hi, self.num_blocks), "checkpoint belongs to another sharding"
        dev = self._engine.device
        for k in PARAM_NAMES:
            self._params[k].copy_(torch.from_numpy(st["params"][k]).to(dev))
            self._best[k].copy_(torch.from_numpy(st["best"][k]).to(dev))
            self._state.m[k].copy_(torch.from_numpy(st["m"][k]).to(dev))
            self._state.v[k].copy_(torch.from_numpy(st["v"][k]).to(dev))
        self._state.c.beta1_power, self._state.c.beta2_power = st["beta_pow"]
        self._state.c.step = st["step"]
        self._active.copy_(torch.from_numpy(st["active"]).to(dev))
        self._diverged.copy_(torch.from_numpy(st["diverged"]).to(dev))
        self.iter = st["iter"]
        self.losses, self.mses, self.num_pis = st["losses"], st["mses"], st["num_pis"]
        self._best_loss_blocks = None if st.get("best_loss_blocks") is None else torch.from_numpy(st["best_loss_blocks"]).to(dev)
        self._loss0 = None if st.get("loss0") is None else torch.from_numpy(st["loss0"]).to(dev)
        self.best_loss, self.best_mse = st.get("best_loss"), st.get("best_mse", [])
        self.valid = False


# =====================================================================================================
# shared-kernel image mode: the reference's whole-image fit
# =====================================================================================================
def _default_shared_factory(cfg, device):
    from .engine import SharedEngine
    return SharedEngine(cfg, device)


class SharedSmoe:
    """``Smoe`` with ONE global kernel set, as the reference fits whole images
    (``Smoe(image, kernels_per_dim=[12, 12], batch_size=[32, 32])``): ``kernels_per_dim`` is the
    global kernel grid, ``batch_size`` the pixel batch of a pass (smoe.py:231-247), every batch keeps
    its own kernel list (smoe.py:315,1763-1766,2287-2365), the gradients of all batches are
    accumulated and one Adam step follows (smoe.py:1148-1150,1788).  ``get_params()`` returns the
    reference's layout exactly (leading axis = kernels).  Multi-GPU: the BATCHES are sharded over
    ranks and the accumulated gradient buffer is all-reduced (RCCL) before the Adam step.
    ``overlap_of_batches`` > 0: the halo of every batch takes part in the kernel-list influence test
    only (its loss is cropped, smoe.py:909-923)."""

    def __init__(self, image, kernels_per_dim=None, train_pis=True, init_params=None, start_batches=1,
                 batch_size=None, train_gammas=True, train_musx=True, use_determinant=False, normalize_pis=True,
                 use_yuv=True, precision=8, iter_offset=0, margin=0.5, overlap_of_batches=0, device=None,
                 engine_factory=None, quantization_mode=0, quantize_pis=False, bit_depths=None, lower_bounds=None,
                 upper_bounds=None, only_y_gamma=False, use_diff_center=False, ssim_opt=False, train_inverse_cov=True,
                 radial_as=False, loss_mask=None, kernel_count_as_norm_l1=False, **unsupported):
        for name, val in unsupported.items():
            if val:
                raise NotImplementedError(f"SharedSmoe({name}=...) is outside the hot path (SURVEY section 8)")
        if quantization_mode not in (0, 1, 2, 3):
            raise ValueError("quantization_mode must be 0, 1, 2 or 3")                # smoe_test.py:298-301
        self.kernel_count_as_norm_l1 = bool(kernel_count_as_norm_l1)                  # smoe.py:1022-1027
        assert kernels_per_dim is not None or init_params is not None, \
            "You need to specify the kernel grid size or give initial parameters."
        image = np.asarray(image, dtype=np.float32)
        self.image = image
        d = self.dim_domain = image.ndim - 1
        self.num_pixel = int(np.prod(image.shape[:d]))
        self.precision, self.margin = precision, margin
        self.use_yuv = bool(use_yuv) and image.shape[-1] == 3
        self.use_determinant = use_determinant
        self.train_pis, self.train_gammas, self.train_musx = train_pis, train_gammas, train_musx
        self.quantization_mode = int(quantization_mode)
        self.quantize_pis = bool(quantize_pis) or quantization_mode >= 2           # smoe.py:474, smoe_test.py:36-37
        self.bit_depths = [20, 18, 6, 10, 10] if bit_depths is None else list(bit_depths)
        self.lower_bounds = [-2500, -.3, -5, 0, -32] if lower_bounds is None else list(lower_bounds)
        self.upper_bounds = [2500, 1.3, 5, 2, 32] if upper_bounds is None else list(upper_bounds)
        self.radial_as = bool(radial_as)                                   # smoe.py:429-434,714-719
        self.train_inverse_cov = bool(train_inverse_cov)                  # smoe.py:41: the constructor default is True
        self.ssim_opt = bool(ssim_opt)                                    # smoe.py:929,980-1011: 1 - SSIM per batch
        self.only_y_gamma = bool(only_y_gamma) and self.use_yuv          # smoe_test.py:43-44, smoe.py:725-729
        self.use_diff_center = bool(use_diff_center)
        self.overlap = int(overlap_of_batches)                            # smoe.py:244
        if batch_size is None or batch_size[0] is None:
            bs = blk.get_batch_shape(start_batches, tuple(image.shape[:d]) + (d + image.shape[-1],))[:-1]
        elif len(batch_size) == d:
            bs = tuple(int(b) for b in batch_size)
        elif len(batch_size) == 1:
            bs = tuple(int(batch_size[0]) for _ in range(d))
        else:
            raise ValueError("Required BatchSize doesn't fit to input dimension")
        for ii in range(d):                                               # smoe.py:239-241
            if image.shape[ii] % bs[ii] > 0:
                raise ValueError("Required BatchSize is not compatible to input dimensions")
        self.batch_size_valued = bs
        self.batch_size = tuple(int(b) + 2 * self.overlap for b in bs)    # smoe.py:245
        self.Nb = int(np.prod(bs))
        blocks_all, _ = blk.image_to_blocks(image, bs)
        self.num_batches = self.start_batches = blocks_all.shape[0]
        self.rank, self.world_size = sdist.world()
        self.lo, self.hi = sdist.shard_range(self.num_batches, self.rank, self.world_size)
        if init_params:
            p0 = {k: np.ascontiguousarray(init_params[k], dtype=np.float32) for k in PARAM_NAMES}
            self.musX_init = p0["musX"]
        else:
            kpd = list(kernels_per_dim)
            if len(kpd) == 1:
                kpd = kpd * d
            p0 = {k: v[0] for k, v in blk.init_block_params(image[None], kpd, normalize_pis, self.train_inverse_cov).items()}
            self.musX_init = blk.gen_domain_grid(kpd, d)
        if self.radial_as:                    # one steering value per kernel: A_init[:, 0, 0] tiled over the diagonal
            a0 = p0["A_diagonal"] if p0["A_diagonal"].ndim == 1 else p0["A_diagonal"][:, 0, 0]
            p0["A_diagonal"] = np.ascontiguousarray(a0[:, None, None] * np.eye(d, dtype=np.float32))
        self.kernels = self.start_pis = self.kernel_count = p0["pis"].shape[0]
        # use_diff_center (smoe.py:390-394,746-747): the trained variable is the offset from the kernel grid; the
        # engine works on grid + offset, the getters subtract the grid
        self._mus_grid = np.ascontiguousarray(p0["musX"]).copy() if self.use_diff_center else None
        self._factory = engine_factory or _default_shared_factory
        self._device = device
        self._engine, self._engine_key = None, None
        self.optimizer1 = self.optimizer2 = self.optimizer3 = None
        self.grad_clip_value_abs = None
        self._make_engine(0.0, 0.0)
        dev = self._engine.device
        self._target = torch.from_numpy(blk.to_planar(blocks_all[self.lo:self.hi])).to(dev)
        self._params = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in p0.items()}
        self._best = {k: v.clone() for k, v in self._params.items()}
        self._state = self._engine.new_adam_state(self._params)
        self._lists = self._engine.new_lists(self.hi - self.lo)           # smoe.py:315
        # loss_mask (smoe.py:1674-1677): per-pixel loss weights, cut into the batches like the image
        self.loss_mask = loss_mask
        self._loss_w = None
        if loss_mask is not None:
            if self.ssim_opt:
                raise NotImplementedError("ssim_opt ignores loss weights (as the reference does)")
            lm, _ = blk.image_to_blocks(np.asarray(loss_mask, dtype=np.float32)[..., None], bs)
            self._loss_w = torch.from_numpy(np.ascontiguousarray(lm.reshape(self.num_batches, -1))).to(dev)
            self._engine.set_loss_weights(self._loss_w)
        self.losses, self.mses, self.num_pis, self.num_svs = [], [], [], []
        self.losses_history, self.mses_history = [], []
        self.best_loss, self.best_mse = None, []
        self.iter = iter_offset
        self.valid = False
        self.reconstruction_image = self.weight_matrix_argmax = None
        self.qparams = self.rparams = None                                # quantizer.py products (leading axis 1 = the model)
        self.qreconstruction_image = None
        self.qvalid = False
        self.qlosses, self.qmses = [], []
        self.best_qloss, self.best_qmse = None, []

    def _make_engine(self, pis_l1, u_l1):
        from .engine import SharedConfig
        o1, o2, o3 = self.optimizer1, self.optimizer2, self.optimizer3
        cfg = SharedConfig(
            image_shape=tuple(self.image.shape[:self.dim_domain]), batch_shape=self.batch_size_valued,
            channels=self.image.shape[-1], kernels=self.kernels, precision=self.precision, margin=self.margin,
            use_determinant=bool(self.use_determinant), use_yuv=bool(self.use_yuv), train_pis=bool(self.train_pis),
            train_gammas=bool(self.train_gammas), train_musx=bool(self.train_musx),
            lr_expert=o1._lr if o1 else 0.0, lr_pis=o2._lr if o2 else 0.0, lr_steer=o3._lr if o3 else 0.0,
            beta1=o1._beta1 if o1 else 0.9, beta2=o1._beta2 if o1 else 0.999, adam_eps=o1._epsilon if o1 else 1e-8,
            grad_clip=float(self.grad_clip_value_abs or 0.0), pis_l1=float(pis_l1), u_l1=float(u_l1),
            start_pis=self.kernels, overlap=self.overlap, quantization_mode=self.quantization_mode,
            quantize_pis=self.quantize_pis, bit_depths=tuple(self.bit_depths), lower_bounds=tuple(self.lower_bounds),
            upper_bounds=tuple(self.upper_bounds), only_y_gamma=self.only_y_gamma, ssim_opt=self.ssim_opt,
            train_inverse_cov=self.train_inverse_cov, radial_as=self.radial_as,
            kernel_count_as_norm_l1=self.kernel_count_as_norm_l1)
        key = repr(sorted(cfg.__dict__.items()))
        if key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = self._factory(cfg, self._device)
            self._engine_key = key
            if self._mus_grid is not None and self.quantization_mode >= 2:   # the graph quantises the OFFSETS (smoe.py:746-747)
                self._mus_grid_dev = torch.from_numpy(np.ascontiguousarray(self._mus_grid, dtype=np.float32)).to(self._engine.device)
                self._engine.set_center_grid(self._mus_grid_dev)
            if getattr(self, "_loss_w", None) is not None:                # a re-created engine needs the weights again
                self._engine.set_loss_weights(self._loss_w)

    @property
    def kernel_list_per_batch(self):
        bits = self._lists.cpu().numpy().view(np.uint32)
        K = self.kernels
        return [np.array([(row[k >> 5] >> (k & 31)) & 1 for k in range(K)], dtype=bool) for row in bits]

    def set_optimizer(self, optimizer1, optimizer2=None, optimizer3=None, optimizer4=None, optimizer5=None,
                      grad_clip_value_abs=None):
        self.optimizer1 = optimizer1
        self.optimizer2 = optimizer1 if optimizer2 is None else optimizer2
        self.optimizer3 = optimizer1 if optimizer3 is None else optimizer3
        self.grad_clip_value_abs = grad_clip_value_abs
        self._make_engine(0.0, 0.0)
        self._state = self._engine.new_adam_state(self._params)

    def _global(self, loss, sse):
        s = torch.stack([loss.double().sum(), sse.double().sum()])
        sdist.allreduce_sum_(s)
        s = s.cpu().numpy()
        loss_val = float(s[0]) * self.Nb / self.num_pixel                              # smoe.py:1758
        mse_val = float(s[1]) / (self.num_pixel * self.image.shape[-1]) * (2 ** self.precision) ** 2
        pis = self._params["pis"]
        if self.quantize_pis:                                        # pis_mask = qpis > 0 (smoe.py:480,1012)
            pis = _fake_quant_fixed(pis, self.lower_bounds[3], self.upper_bounds[3], self.bit_depths[3])
        return loss_val, mse_val, int((pis > 0).sum().item())

    def _quantize(self):
        """quantize_params + rescaler (quantizer.py:4-145) on the global kernel set (one model = leading axis 1)."""
        from .quantizer import quantize_params, rescaler
        self.qparams = quantize_params(self, {k: v[None] for k, v in self.get_params().items()})
        if self.quantization_mode == 1:
            self.rparams = rescaler(self, self.qparams)

    def run_batched(self, pis_l1=0, u_l1=0, sv_l1_sub_l2=0, train=True, update_reconstruction=False,
                    with_quantized_params=False, **kw):
        for name, val in kw.items():
            if val not in (False, None, 100):
                raise NotImplementedError(f"run_batched({name}=...) is outside the hot path")
        self._make_engine(pis_l1, u_l1)
        eng, nb = self._engine, self.hi - self.lo
        dev = eng.device
        if with_quantized_params:                                         # smoe.py:1688-1689: the lists are not touched
            assert self.rparams is not None, "quantize_params + rescaler first (smoe.py:1499-1501)"
            self.qvalid = False
            rp = {k: torch.from_numpy(np.ascontiguousarray(self.rparams[k][0], dtype=np.float32)).to(dev) for k in PARAM_NAMES}
            out = eng.forward(self._target, rp, self._lists, first_batch=self.lo, want_recon=update_reconstruction,
                              want_argmax=False, update_lists=False)
            if update_reconstruction:
                bs, d = self.batch_size_valued, self.dim_domain
                rec = sdist.allgather_blocks(blk.from_planar(out["recon"].cpu().numpy(), bs), self.num_batches)
                self.qreconstruction_image = blk.blocks_to_image(rec, self.image.shape[:d], bs)
                self.qvalid = True
            loss_val, mse_val, num_pi = self._global(out["loss"], out["sse"])
            return loss_val, mse_val, num_pi, 0
        self.valid = False
        if train:
            assert self.optimizer1 is not None, "no optimizer found, you have to specify one!"
            loss = torch.zeros((nb,), dtype=torch.float32, device=dev)
            sse = torch.zeros((nb,), dtype=torch.float32, device=dev)
            eng.accumulate(self._target, self._params, self._lists, first_batch=self.lo, loss_out=loss, sse_out=sse)
            if self.world_size > 1:
                sdist.allreduce_sum_(eng.grad_buffer())                    # the gradient exchange of the pass
            eng.apply(self._params, self._state)
        else:
            out = eng.forward(self._target, self._params, self._lists, first_batch=self.lo,
                              want_recon=update_reconstruction, want_argmax=update_reconstruction)
            loss, sse = out["loss"], out["sse"]
            if update_reconstruction:
                bs, d = self.batch_size_valued, self.dim_domain
                rec = sdist.allgather_blocks(blk.from_planar(out["recon"].cpu().numpy(), bs), self.num_batches)
                self.reconstruction_image = blk.blocks_to_image(rec, self.image.shape[:d], bs)
                am = out["argmax"].cpu().numpy().astype(np.int64).reshape((nb,) + bs)
                am = sdist.allgather_blocks(am, self.num_batches)
                self.weight_matrix_argmax = blk.blocks_to_image(am[..., None], self.image.shape[:d], bs)[..., 0]
                self.valid = True
        loss_val, mse_val, num_pi = self._global(loss, sse)
        return loss_val, mse_val, num_pi, 0

    def train(self, num_iter, val_iter=100, ukl_iter=None, optimizer1=None, optimizer2=None, optimizer3=None,
              grad_clip_value_abs=None, pis_l1=0, u_l1=0, callbacks=(), **kw):
        if ukl_iter is None:
            ukl_iter = val_iter
        if optimizer1:
            self.set_optimizer(optimizer1, optimizer2, optimizer3, grad_clip_value_abs=grad_clip_value_abs)
        assert self.optimizer1 is not None, "no optimizer found, you have to specify one!"
        self._make_engine(pis_l1, u_l1)
        _train_loop(self, num_iter, val_iter, ukl_iter, pis_l1, u_l1, callbacks)

    # hooks of _train_loop
    def _after_first_evaluation(self):
        pass

    def _fit_iterations(self, n):
        eng = self._engine
        if self.world_size == 1:                                             # n x (accumulate; apply) in one call
            eng.fit(self._target, self._params, self._state, self._lists, n)
        else:
            for _ in range(n):
                eng.accumulate(self._target, self._params, self._lists, first_batch=self.lo)
                sdist.allreduce_sum_(eng.grad_buffer())                      # the gradient exchange of the pass
                eng.apply(self._params, self._state)

    def _readmit_kernels(self):
        self._engine.update_kernel_list(self._params, self._lists, first_batch=self.lo)

    def _keep_best(self, loss_val):
        if not self.best_loss or loss_val < self.best_loss:                  # smoe.py:1574-1576
            self.best_loss = loss_val
            for k in PARAM_NAMES:
                self._best[k].copy_(self._params[k])

    def _host_params(self, p):
        out = {k: v.cpu().numpy().copy() for k, v in p.items()}
        if self._mus_grid is not None:                     # use_diff_center: report the trained offsets
            out["musX"] = out["musX"] - self._mus_grid
        if self.radial_as:                                 # the reference's (K,) variable
            out["A_diagonal"] = np.ascontiguousarray(out["A_diagonal"][:, 0, 0])
        return out

    def get_params(self):
        return self._host_params(self._params)

    def get_best_params(self):
        return self._host_params(self._best)

    def get_reconstruction(self):
        if not self.valid:
            self.run_batched(train=False, update_reconstruction=True)
        return self.reconstruction_image

    def get_weight_matrix_argmax(self):
        if not self.valid:
            self.run_batched(train=False, update_reconstruction=True)
        return self.weight_matrix_argmax

    def get_qreconstruction(self):
        if not self.qvalid:
            self.run_batched(train=False, update_reconstruction=True, with_quantized_params=True)
        return self.qreconstruction_image

    def get_qlosses(self):
        return self.qlosses

    def get_qmses(self):
        return self.qmses

    def get_psnr(self):
        rec = self.get_reconstruction()
        return float(-10.0 * np.log10(np.mean((rec.astype(np.float64) - self.image.astype(np.float64)) ** 2)))

    def get_losses(self):
        return self.losses

    def get_mses(self):
        return self.mses

    def get_num_pis(self):
        return self.num_pis

    def get_num_svs(self):
        return self.num_svs

    def get_best_loss(self):
        return self.best_loss

    def get_best_mse(self):
        return self.best_mse

    def get_iter(self):
        return self.iter

    def get_original_image(self):
        return np.squeeze(self.image)
