"""BlockEngine: thin host wrapper around the C ABI (include/smoe_hip.h).

PyTorch is used only as plumbing: device allocations, ``data_ptr()`` and the current HIP
stream.  All arithmetic happens in libsmoe_hip.so; there is no eager/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, Optional, Sequence

import torch

from . import _lib

PARAM_NAMES = ("pis", "musX", "A_diagonal", "A_corr", "gamma_e", "nu_e")


@dataclasses.dataclass
class EngineConfig:
    """Mirror of ``smoe_config`` (include/smoe_hip.h); defaults = smoe_test.py CLI defaults
    with kernel adding off (smoe_test.py:262-352)."""
    block_shape: Sequence[int]
    channels: int
    kernels: int
    precision: int = 8
    margin: float = 0.5
    use_determinant: bool = True
    use_yuv: bool = False
    train_pis: bool = True
    train_gammas: bool = True
    train_musx: bool = True
    lr_expert: float = 1e-3
    lr_pis: float = 1e-5
    lr_steer: float = 1.0
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    grad_clip: float = 0.0
    pis_l1: float = 0.0
    u_l1: float = 0.0
    start_pis: int = 0
    only_y_gamma: bool = False
    ssim_opt: bool = False
    # fake-quantised variables in the graph (smoe.py:474-538); tuples ordered A, musX, nu_e, pis, gamma_e
    quantization_mode: int = 0
    quantize_pis: bool = False
    bit_depths: Sequence[int] = (20, 18, 6, 10, 10)
    lower_bounds: Sequence[float] = (-2500, -.3, -5, 0, -32)
    upper_bounds: Sequence[float] = (2500, 1.3, 5, 2, 32)
    train_inverse_cov: bool = False      # smoe.py:734-735,791-793 (reference ctor default True, CLI default False)
    radial_as: bool = False              # smoe.py:714-719: equal steering diagonals, tied gradient, A_corr untrained
    kernel_count_as_norm_l1: bool = False  # smoe.py:1022-1027

    @property
    def dim(self) -> int:
        return len(self.block_shape)

    @property
    def pixels(self) -> int:
        n = 1
        for s in self.block_shape:
            n *= int(s)
        return n


def param_shapes(B: int, K: int, d: int, Cc: int) -> Dict[str, tuple]:
    """get_params() layout (smoe.py:1795-1800) with a leading block axis."""
    return {"pis": (B, K), "musX": (B, K, d), "A_diagonal": (B, K, d, d), "A_corr": (B, K, d, d),
            "gamma_e": (B, K, d, Cc), "nu_e": (B, K, Cc)}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class AdamState:
    """TF1 Adam slots on the device + the running beta powers on the host."""

    def __init__(self, params: Dict[str, torch.Tensor], beta1: float, beta2: float):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.c = _lib.SmoeAdamState()
        self.c.beta1_power = beta1
        self.c.beta2_power = beta2
        self.c.step = 0

    @property
    def step(self) -> int:
        return int(self.c.step)


class BlockEngine:
    def __init__(self, cfg: EngineConfig, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("BlockEngine needs a HIP device; this package has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = cfg
        c = _lib.SmoeConfig()
        c.abi_version = _lib.SMOE_ABI_VERSION
        c.device = self.device.index or 0
        c.dim = cfg.dim
        shape = list(cfg.block_shape) + [1] * (3 - cfg.dim)
        for i in range(3):
            c.block_shape[i] = int(shape[i])
        c.channels, c.kernels, c.precision = cfg.channels, cfg.kernels, cfg.precision
        c.margin = cfg.margin
        c.use_determinant, c.use_yuv = int(cfg.use_determinant), int(cfg.use_yuv)
        c.train_pis, c.train_gammas, c.train_musx = int(cfg.train_pis), int(cfg.train_gammas), int(cfg.train_musx)
        c.lr_expert, c.lr_pis, c.lr_steer = cfg.lr_expert, cfg.lr_pis, cfg.lr_steer
        c.beta1, c.beta2, c.adam_eps = cfg.beta1, cfg.beta2, cfg.adam_eps
        c.grad_clip = cfg.grad_clip or 0.0
        c.pis_l1, c.u_l1 = cfg.pis_l1, cfg.u_l1
        c.start_pis = cfg.start_pis or cfg.kernels
        c.only_y_gamma = int(cfg.only_y_gamma)
        c.ssim_opt = int(cfg.ssim_opt)
        c.quantization_mode, c.quantize_pis = int(cfg.quantization_mode), int(cfg.quantize_pis)
        for i in range(5):
            c.bit_depths[i] = int(cfg.bit_depths[i])
            c.lower_bounds[i], c.upper_bounds[i] = float(cfg.lower_bounds[i]), float(cfg.upper_bounds[i])
        c.train_inverse_cov = int(cfg.train_inverse_cov)
        c.radial_as = int(cfg.radial_as)
        c.kernel_count_as_norm_l1 = int(cfg.kernel_count_as_norm_l1)
        self._c = c
        self._h = C.c_void_p()
        _lib.check(self.lib.smoe_create(C.byref(self._h), C.byref(c)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.smoe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_params(self, p: Dict[str, torch.Tensor], B: int):
        shapes = param_shapes(B, self.cfg.kernels, self.cfg.dim, self.cfg.channels)
        for name in PARAM_NAMES:
            t = p[name]
            if tuple(t.shape) != shapes[name] or t.dtype != torch.float32 or not t.is_contiguous() \
                    or t.device != self.device:
                raise ValueError(f"parameter {name}: expected contiguous float32 {shapes[name]} on {self.device}, "
                                 f"got {t.dtype} {tuple(t.shape)} on {t.device}")

    def _cparams(self, p: Dict[str, torch.Tensor]) -> _lib.SmoeParams:
        s = _lib.SmoeParams()
        for name in PARAM_NAMES:
            setattr(s, name, p[name].data_ptr())
        return s

    def _check_target(self, target: torch.Tensor, loss_w: Optional[torch.Tensor]) -> int:
        if target.dim() != 3 or target.shape[1] != self.cfg.channels or target.shape[2] != self.cfg.pixels \
                or target.dtype != torch.float32 or not target.is_contiguous() or target.device != self.device:
            raise ValueError(f"target must be contiguous float32 [B,{self.cfg.channels},{self.cfg.pixels}] on {self.device}")
        B = target.shape[0]
        if loss_w is not None and (tuple(loss_w.shape) != (B, self.cfg.pixels) or loss_w.dtype != torch.float32
                                   or not loss_w.is_contiguous() or loss_w.device != self.device):
            raise ValueError("loss_w must be contiguous float32 [B,N]")
        return B

    def new_params(self, B: int) -> Dict[str, torch.Tensor]:
        shapes = param_shapes(B, self.cfg.kernels, self.cfg.dim, self.cfg.channels)
        return {k: torch.zeros(s, dtype=torch.float32, device=self.device) for k, s in shapes.items()}

    def new_adam_state(self, params: Dict[str, torch.Tensor]) -> AdamState:
        return AdamState(params, self.cfg.beta1, self.cfg.beta2)

    def coords(self) -> torch.Tensor:
        out = torch.empty((self.cfg.dim, self.cfg.pixels), dtype=torch.float32)
        _lib.check(self.lib.smoe_get_coords(self._h, C.c_void_p(out.data_ptr())))
        return out

    def set_tiling(self, lanes_per_block: int):
        _lib.check(self.lib.smoe_set_tiling(self._h, lanes_per_block))

    def set_total_blocks(self, total_blocks: int):
        """Block count of the WHOLE job the calls of this engine are shards of (0 = each call's own count): the kernel
        tiling -- and with it the summation order inside a block -- is then the same for every split of the job over
        calls / ranks, so per-block results are bit-identical (include/smoe_hip.h: smoe_set_total_blocks)."""
        _lib.check(self.lib.smoe_set_total_blocks(self._h, int(total_blocks)))

    def set_center_grid(self, grid: Optional[torch.Tensor]):
        """use_diff_center with quantization_mode 2 / 3: the kernel-grid centres [B, K, d] (float32, on the device, laid
        out like musX) the trained offsets are relative to, or None to clear.  The engine keeps a reference."""
        if grid is not None:
            assert grid.dtype == torch.float32 and grid.is_contiguous() and grid.device == self.device
            assert grid.ndim == 3 and tuple(grid.shape[1:]) == (self.cfg.kernels, len(self.cfg.block_shape))
        self._center_grid = grid
        _lib.check(self.lib.smoe_set_center_grid(self._h, _ptr(grid)))

    def fit_occupancy(self, B: int) -> int:
        return int(self.lib.smoe_fit_occupancy(self._h, B))

    def fit_variant(self, B: int) -> str:
        return self.lib.smoe_fit_variant(self._h, B).decode()

    # -- the hot path ------------------------------------------------------------
    def forward(self, target, params, active, loss_w=None, want_recon=True, want_argmax=False,
                want_gate=False, update_active=True):
        B = self._check_target(target, loss_w)
        self._check_params(params, B)
        dev = self.device
        N, K, Cc = self.cfg.pixels, self.cfg.kernels, self.cfg.channels
        out = {
            "loss": torch.empty((B,), dtype=torch.float32, device=dev),
            "sse": torch.empty((B,), dtype=torch.float32, device=dev),
            "recon": torch.empty((B, Cc, N), dtype=torch.float32, device=dev) if want_recon else None,
            "argmax": torch.empty((B, N), dtype=torch.uint8, device=dev) if want_argmax else None,
            "gate_w": torch.empty((B, K, N), dtype=torch.float32, device=dev) if want_gate else None,
        }
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_forward(self._h, B, _ptr(target), _ptr(loss_w), C.byref(cp),
                                         _ptr(out["recon"]), _ptr(out["argmax"]), _ptr(out["gate_w"]),
                                         _ptr(out["loss"]), _ptr(out["sse"]), _ptr(active),
                                         int(update_active), self._stream()))
        return out

    def fit(self, target, params, state: AdamState, active, n_iters: int, loss_w=None, diverged=None,
            loss0=None, loss_out=None, sse_out=None, loss_w_is_sample=False):
        """loss_w_is_sample: ``loss_w`` is a pixel sub-sample (N / n for the drawn pixels, 0 otherwise; smoe.py:1664-1667):
        the pixels with weight 0 are "not fed" and do not vote in the kernel-list prune (include/smoe_hip.h: smoe_set_sampling)."""
        B = self._check_target(target, loss_w)
        _lib.check(self.lib.smoe_set_sampling(self._h, int(bool(loss_w_is_sample))))
        self._check_params(params, B)
        self._check_params(state.m, B)
        self._check_params(state.v, B)
        state.c.m = self._cparams(state.m)
        state.c.v = self._cparams(state.v)
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_fit(self._h, B, _ptr(target), _ptr(loss_w), C.byref(cp), C.byref(state.c),
                                     int(n_iters), _ptr(loss_out), _ptr(sse_out), _ptr(active), _ptr(diverged),
                                     _ptr(loss0), self._stream()))

    def prepare_fit(self, target, params, state: AdamState, active, loss_w=None, diverged=None, loss0=None, loss_out=None,
                    sse_out=None, loss_w_is_sample=False):
        """``fit`` with the argument checks and the ctypes marshalling done once: returns ``run(n_iters)`` that only makes the
        C call (a few microseconds of host time per launch instead of the ~50 of the checked path).  The tensors must stay
        alive and in place while ``run`` is used (measurement loops, bench.py)."""
        B = self._check_target(target, loss_w)
        self._check_params(params, B)
        self._check_params(state.m, B)
        self._check_params(state.v, B)
        state.c.m = self._cparams(state.m)
        state.c.v = self._cparams(state.v)
        cp = self._cparams(params)
        args = (_ptr(target), _ptr(loss_w), C.byref(cp), C.byref(state.c))
        tail = (_ptr(loss_out), _ptr(sse_out), _ptr(active), _ptr(diverged), _ptr(loss0), self._stream())
        fit, h, sample = self.lib.smoe_fit, self._h, int(bool(loss_w_is_sample))
        keep = (cp, target, loss_w, params, state, active, diverged, loss0, loss_out, sse_out)

        def run(n_iters: int, _keep=keep):
            _lib.check(self.lib.smoe_set_sampling(h, sample))
            _lib.check(fit(h, B, *args, int(n_iters), *tail))
        return run

    def update_kernel_list(self, params, active):
        B = active.shape[0]
        self._check_params(params, B)
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_update_kernel_list(self._h, B, C.byref(cp), _ptr(active), self._stream()))

    def checkpoint_best(self, loss, best_loss, params, best):
        B = loss.shape[0]
        self._check_params(params, B)
        self._check_params(best, B)
        cp, cb = self._cparams(params), self._cparams(best)
        _lib.check(self.lib.smoe_checkpoint_best(self._h, B, _ptr(loss), _ptr(best_loss), C.byref(cp),
                                                 C.byref(cb), self._stream()))

    def reduce_scalars(self, loss, sse, active) -> torch.Tensor:
        """[sum loss*N, sum sse, sum popcount(active)] as a float64 device tensor."""
        out = torch.empty((3,), dtype=torch.float64, device=self.device)
        B = 0
        for t in (loss, sse, active):
            if t is not None:
                B = t.shape[0]
        _lib.check(self.lib.smoe_reduce_scalars(self._h, B, _ptr(loss), _ptr(sse), _ptr(active), _ptr(out),
                                                self._stream()))
        return out


# =====================================================================================================
# shared-kernel image mode
# =====================================================================================================
@dataclasses.dataclass
class SharedConfig:
    """Mirror of ``smoe_shared_config``: ONE global kernel set over the image, batches of
    ``batch_shape`` pixels with per-batch kernel lists."""
    image_shape: Sequence[int]
    batch_shape: Sequence[int]
    channels: int
    kernels: int
    precision: int = 8
    margin: float = 0.5
    use_determinant: bool = True
    use_yuv: bool = False
    train_pis: bool = True
    train_gammas: bool = True
    train_musx: bool = True
    lr_expert: float = 1e-3
    lr_pis: float = 1e-5
    lr_steer: float = 1.0
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    grad_clip: float = 0.0
    pis_l1: float = 0.0
    u_l1: float = 0.0
    start_pis: int = 0
    only_y_gamma: bool = False
    overlap: int = 0
    ssim_opt: bool = False
    train_inverse_cov: bool = False
    radial_as: bool = False
    kernel_count_as_norm_l1: bool = False  # smoe.py:1022-1027: pis_l1 / count(qpis > 0) over the image
    quantization_mode: int = 0
    quantize_pis: bool = False
    bit_depths: Sequence[int] = (20, 18, 6, 10, 10)
    lower_bounds: Sequence[float] = (-2500, -.3, -5, 0, -32)
    upper_bounds: Sequence[float] = (2500, 1.3, 5, 2, 32)

    @property
    def dim(self) -> int:
        return len(self.image_shape)


class SharedEngine:
    """Host wrapper of the smoe_shared_* entry points.  Parameters: dict of contiguous float32
    device tensors in the get_params() layout with leading K (no block axis)."""

    def __init__(self, cfg: SharedConfig, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("SharedEngine needs a HIP device; this package has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = cfg
        c = _lib.SmoeSharedConfig()
        c.abi_version, c.device, c.dim = _lib.SMOE_ABI_VERSION, self.device.index or 0, cfg.dim
        for i in range(3):
            c.image_shape[i] = int(cfg.image_shape[i]) if i < cfg.dim else 1
            c.batch_shape[i] = int(cfg.batch_shape[i]) if i < cfg.dim else 1
        c.channels, c.kernels, c.precision, c.margin = cfg.channels, cfg.kernels, cfg.precision, cfg.margin
        c.use_determinant, c.use_yuv = int(cfg.use_determinant), int(cfg.use_yuv)
        c.train_pis, c.train_gammas, c.train_musx = int(cfg.train_pis), int(cfg.train_gammas), int(cfg.train_musx)
        c.lr_expert, c.lr_pis, c.lr_steer = cfg.lr_expert, cfg.lr_pis, cfg.lr_steer
        c.beta1, c.beta2, c.adam_eps = cfg.beta1, cfg.beta2, cfg.adam_eps
        c.grad_clip, c.pis_l1, c.u_l1 = cfg.grad_clip or 0.0, cfg.pis_l1, cfg.u_l1
        c.start_pis = cfg.start_pis or cfg.kernels
        c.only_y_gamma = int(cfg.only_y_gamma)
        c.overlap = int(cfg.overlap)
        c.ssim_opt = int(cfg.ssim_opt)
        c.train_inverse_cov = int(cfg.train_inverse_cov)
        c.radial_as = int(cfg.radial_as)
        c.kernel_count_as_norm_l1 = int(cfg.kernel_count_as_norm_l1)
        c.quantization_mode, c.quantize_pis = int(cfg.quantization_mode), int(cfg.quantize_pis)
        for i in range(5):
            c.bit_depths[i] = int(cfg.bit_depths[i])
            c.lower_bounds[i], c.upper_bounds[i] = float(cfg.lower_bounds[i]), float(cfg.upper_bounds[i])
        self._h = C.c_void_p()
        _lib.check(self.lib.smoe_shared_create(C.byref(self._h), C.byref(c)))
        self.num_batches = int(self.lib.smoe_shared_num_batches(self._h))
        self.list_words = int(self.lib.smoe_shared_list_words(self._h))
        self.batch_pixels = 1
        for b in cfg.batch_shape:
            self.batch_pixels *= int(b)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.smoe_shared_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _cparams(self, p):
        shapes = param_shapes(1, self.cfg.kernels, self.cfg.dim, self.cfg.channels)
        s = _lib.SmoeParams()
        for name in PARAM_NAMES:
            t = p[name]
            if tuple(t.shape) != shapes[name][1:] or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"parameter {name}: expected contiguous float32 {shapes[name][1:]} on {self.device}")
            setattr(s, name, t.data_ptr())
        return s

    def new_lists(self, nb: Optional[int] = None) -> torch.Tensor:
        """All kernels listed in every batch (smoe.py:315)."""
        nb = self.num_batches if nb is None else nb
        K, KW = self.cfg.kernels, self.list_words
        words = torch.full((nb, KW), -1, dtype=torch.int32, device=self.device)
        if K % 32:
            words[:, KW - 1] = (1 << (K % 32)) - 1
        return words

    def new_adam_state(self, params):
        return AdamState(params, self.cfg.beta1, self.cfg.beta2)

    def _check_target(self, target, nb):
        want = (nb, self.cfg.channels, self.batch_pixels)
        if tuple(target.shape) != want or target.dtype != torch.float32 or not target.is_contiguous():
            raise ValueError(f"target must be contiguous float32 {want}")

    def forward(self, target, params, lists, first_batch=0, want_recon=True, want_argmax=False, update_lists=True):
        nb = lists.shape[0]
        self._check_target(target, nb)
        dev = self.device
        out = {"loss": torch.empty((nb,), dtype=torch.float32, device=dev),
               "sse": torch.empty((nb,), dtype=torch.float32, device=dev),
               "recon": torch.empty_like(target) if want_recon else None,
               "argmax": torch.empty((nb, self.batch_pixels), dtype=torch.int32, device=dev) if want_argmax else None}
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_shared_forward(self._h, first_batch, nb, _ptr(target), C.byref(cp), _ptr(out["recon"]),
                                                _ptr(out["argmax"]), _ptr(out["loss"]), _ptr(out["sse"]), _ptr(lists),
                                                int(update_lists), self._stream()))
        return out

    def accumulate(self, target, params, lists, first_batch=0, loss_out=None, sse_out=None):
        nb = lists.shape[0]
        self._check_target(target, nb)
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_shared_accumulate(self._h, first_batch, nb, _ptr(target), C.byref(cp), _ptr(loss_out),
                                                   _ptr(sse_out), _ptr(lists), self._stream()))

    def apply(self, params, state: AdamState):
        state.c.m = self._cparams(state.m)
        state.c.v = self._cparams(state.v)
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_shared_apply(self._h, C.byref(cp), C.byref(state.c), self._stream()))

    def discard_gradients(self):
        """zero_op alone (smoe.py:1613): drop what accumulate() has gathered since the last apply()."""
        _lib.check(self.lib.smoe_shared_discard(self._h, self._stream()))

    def fit(self, target, params, state: AdamState, lists, n_iters, loss_out=None, sse_out=None):
        self._check_target(target, self.num_batches)
        state.c.m = self._cparams(state.m)
        state.c.v = self._cparams(state.v)
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_shared_fit(self._h, _ptr(target), C.byref(cp), C.byref(state.c), int(n_iters),
                                            _ptr(loss_out), _ptr(sse_out), _ptr(lists), self._stream()))

    def update_kernel_list(self, params, lists, first_batch=0):
        cp = self._cparams(params)
        _lib.check(self.lib.smoe_shared_update_kernel_list(self._h, first_batch, lists.shape[0], C.byref(cp), _ptr(lists),
                                                           self._stream()))

    def set_loss_weights(self, loss_w: Optional[torch.Tensor]):
        """[num_batches, Nb] float32 device tensor of per-pixel loss weights for the WHOLE image (or None to clear);
        the engine keeps a reference so the memory stays alive."""
        if loss_w is not None:
            assert loss_w.dtype == torch.float32 and loss_w.is_contiguous() and loss_w.device == self.device
            assert tuple(loss_w.shape) == (self.num_batches, self.batch_pixels)
        self._loss_w = loss_w
        _lib.check(self.lib.smoe_shared_set_loss_weights(self._h, _ptr(loss_w)))

    def set_center_grid(self, grid: Optional[torch.Tensor]):
        """use_diff_center with quantization_mode 2 / 3: the kernel-grid centres [K, d] (float32, on the device) the trained
        offsets are relative to, or None to clear.  The engine keeps a reference."""
        if grid is not None:
            assert grid.dtype == torch.float32 and grid.is_contiguous() and grid.device == self.device
            assert tuple(grid.shape) == (self.cfg.kernels, len(self.cfg.image_shape))
        self._center_grid = grid
        _lib.check(self.lib.smoe_shared_set_center_grid(self._h, _ptr(grid)))

    def grad_buffer(self) -> torch.Tensor:
        """The gradient accumulation buffer as a float64 device tensor view (for the all-reduce
        between accumulate() and apply() when batches are sharded over ranks)."""
        ptr, cnt = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.smoe_shared_grad_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        n = int(cnt.value)

        class _Arr:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr.value), False), "version": 2}
        return torch.as_tensor(_Arr(), device=self.device)

