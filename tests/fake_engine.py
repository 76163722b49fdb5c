"""Test double of ``BlockEngine`` built on the plain-C oracle (oracle/smoe_oracle.c), working on
torch CPU tensors.  TEST INFRASTRUCTURE: lets the CPU suite drive the ``Smoe`` facade's host
logic (iteration scheduling, validation cadence, best snapshot, sharding, scalar all-reduce)
without a GPU.  Never imported by the product."""
import numpy as np
import torch

from oracle import c_oracle as co
from oracle import smoe_oracle as o

NAMES = o.PARAM_NAMES


class _State:
    def __init__(self, params, b1, b2):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.beta_pow = np.array([b1, b2], np.float32)
        self._step = 0

        class _C:
            pass
        self.c = _C()
        self.c.beta1_power, self.c.beta2_power, self.c.step = b1, b2, 0

    @property
    def step(self):
        return self._step


class OracleEngine:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.device = torch.device("cpu")
        self.ocfg = o.OracleConfig(
            block_shape=tuple(cfg.block_shape), channels=cfg.channels, kernels=cfg.kernels,
            precision=cfg.precision, margin=cfg.margin, use_determinant=cfg.use_determinant, use_yuv=cfg.use_yuv,
            train_pis=cfg.train_pis, train_gammas=cfg.train_gammas, train_musx=cfg.train_musx,
            lr_expert=cfg.lr_expert, lr_pis=cfg.lr_pis, lr_steer=cfg.lr_steer, beta1=cfg.beta1, beta2=cfg.beta2,
            adam_eps=cfg.adam_eps, grad_clip=(cfg.grad_clip or None), pis_l1=cfg.pis_l1, u_l1=cfg.u_l1,
            start_pis=cfg.start_pis or cfg.kernels, only_y_gamma=getattr(cfg, 'only_y_gamma', False),
            ssim_opt=getattr(cfg, 'ssim_opt', False), train_inverse_cov=getattr(cfg, 'train_inverse_cov', False),
            radial_as=getattr(cfg, 'radial_as', False), kernel_count_as_norm_l1=getattr(cfg, 'kernel_count_as_norm_l1', False),
            quantization_mode=getattr(cfg, 'quantization_mode', 0),
            quantize_pis=getattr(cfg, 'quantize_pis', False), bit_depths=tuple(getattr(cfg, 'bit_depths', (20, 18, 6, 10, 10))),
            lower_bounds=tuple(getattr(cfg, 'lower_bounds', (-2500, -.3, -5, 0, -32))),
            upper_bounds=tuple(getattr(cfg, 'upper_bounds', (2500, 1.3, 5, 2, 32))))
        # what the plain-C restatement lacks
        self.numpy_only = (self.ocfg.ssim_opt or self.ocfg.quantization_mode >= 2 or self.ocfg.train_inverse_cov
                           or self.ocfg.radial_as or self.ocfg.kernel_count_as_norm_l1)
        self.coords = np.ascontiguousarray(o.block_coords(cfg.block_shape).T)

    def close(self):
        pass

    def set_center_grid(self, grid):
        self.ocfg.mus_grid = None if grid is None else grid.numpy()

    def new_adam_state(self, params):
        return _State(params, self.cfg.beta1, self.cfg.beta2)

    @staticmethod
    def _np(d):
        return {k: d[k].numpy() for k in NAMES}

    def forward(self, target, params, active, loss_w=None, want_recon=True, want_argmax=False, want_gate=False,
                update_active=True):
        B = target.shape[0]
        act = active.numpy().view(np.uint32)
        p = self._np(params)
        if want_argmax or want_gate or self.numpy_only:
            # the C oracle has no argmax / gate outputs and no SSIM loss: use the numpy one for those
            K = self.cfg.kernels
            mask = ((act[:, None] >> np.arange(K, dtype=np.uint32)[None, :]) & 1).astype(bool)
            tgt = np.ascontiguousarray(target.numpy().transpose(0, 2, 1))
            f = o.forward(p, tgt, self.coords.T, mask, self.ocfg, None if loss_w is None else loss_w.numpy(), np.float32)
            out = {"loss": torch.from_numpy(f["loss"].astype(np.float32)), "sse": torch.from_numpy(f["sse"].astype(np.float32)),
                   "recon": torch.from_numpy(np.ascontiguousarray(f["recon"].transpose(0, 2, 1))),
                   "argmax": torch.from_numpy(f["argmax"].astype(np.uint8)),
                   "gate_w": torch.from_numpy(np.ascontiguousarray(f["wt"]))}
            if update_active:
                bits = (f["active_new"].astype(np.uint32) << np.arange(K, dtype=np.uint32)[None, :]).sum(axis=1)
                act[:] = bits.astype(np.uint32)
            return out
        r = co.forward(self.ocfg, self.coords, target.numpy(), p, act, None if loss_w is None else loss_w.numpy(),
                       want_recon=want_recon, update_active=update_active)
        return {"loss": torch.from_numpy(r["loss"]), "sse": torch.from_numpy(r["sse"]),
                "recon": None if r["recon"] is None else torch.from_numpy(r["recon"]), "argmax": None, "gate_w": None}

    def fit(self, target, params, state, active, n_iters, loss_w=None, diverged=None, loss0=None, loss_out=None,
            sse_out=None, loss_w_is_sample=False):
        act = active.numpy().view(np.uint32)
        state.beta_pow[:] = [state.c.beta1_power, state.c.beta2_power]      # restore() writes the c fields
        if self.numpy_only or loss_w_is_sample:
            return self._fit_numpy(target, params, state, act, n_iters, diverged, loss0, loss_out, sse_out,
                                   loss_w=None if loss_w is None else loss_w.numpy(), sample=loss_w_is_sample)
        r = co.fit(self.ocfg, self.coords, target.numpy(), self._np(params), self._np(state.m), self._np(state.v), act,
                   n_iters, state.beta_pow, None if loss_w is None else loss_w.numpy(),
                   None if diverged is None else diverged.numpy().view(np.uint32),
                   None if loss0 is None else loss0.numpy())
        state._step = int(state.c.step) + n_iters
        state.c.beta1_power, state.c.beta2_power, state.c.step = float(state.beta_pow[0]), float(state.beta_pow[1]), state._step
        if loss_out is not None:
            loss_out.copy_(torch.from_numpy(r["loss"]))
        if sse_out is not None:
            sse_out.copy_(torch.from_numpy(r["sse"]))

    def _fit_numpy(self, target, params, state, act, n_iters, diverged, loss0, loss_out, sse_out, loss_w=None, sample=False):
        """The iteration body of oracle.fit (pass -> prune -> Adam -> divergence test) on the numpy oracle."""
        K = self.cfg.kernels
        shifts = np.arange(K, dtype=np.uint32)[None, :]
        tgt = np.ascontiguousarray(target.numpy().transpose(0, 2, 1))
        p = self._np(params)
        st = {"m": self._np(state.m), "v": self._np(state.v), "t": 0,
              "b1p": np.float32(state.beta_pow[0]), "b2p": np.float32(state.beta_pow[1])}
        stopped = np.zeros((tgt.shape[0],), bool) if diverged is None else diverged.numpy().view(np.uint32).astype(bool)
        f = None
        for _ in range(n_iters):
            mask = ((act[:, None] >> shifts) & 1).astype(bool)
            f = o.forward(p, tgt, self.coords.T, mask, self.ocfg, None if self.ocfg.ssim_opt else loss_w, np.float32, want_grads=True,
                          fed=(loss_w > 0) if (sample and loss_w is not None) else None)
            new = np.where(stopped[:, None], mask, f["active_new"])
            act[:] = (new.astype(np.uint32) << shifts).sum(axis=1).astype(np.uint32)
            p = o.adam_step(p, f["grads"], st, self.ocfg, np.float32, frozen=stopped)
            if loss0 is not None:
                l0 = loss0.numpy()
                stopped = stopped | np.isnan(f["loss"]) | (f["loss"] + 1 > (l0 + 100) * 10)
        for k in NAMES:
            params[k].copy_(torch.from_numpy(np.ascontiguousarray(p[k], dtype=np.float32)))
            state.m[k].copy_(torch.from_numpy(np.ascontiguousarray(st["m"][k], dtype=np.float32)))
            state.v[k].copy_(torch.from_numpy(np.ascontiguousarray(st["v"][k], dtype=np.float32)))
        state.beta_pow[:] = [st["b1p"], st["b2p"]]
        state._step = int(state.c.step) + n_iters
        state.c.beta1_power, state.c.beta2_power, state.c.step = float(st["b1p"]), float(st["b2p"]), state._step
        if diverged is not None:
            diverged.numpy().view(np.uint32)[:] = stopped.astype(np.uint32)
        if loss_out is not None and f is not None:
            loss_out.copy_(torch.from_numpy(f["loss"].astype(np.float32)))
        if sse_out is not None and f is not None:
            sse_out.copy_(torch.from_numpy(f["sse"].astype(np.float32)))

    def update_kernel_list(self, params, active):
        K = self.cfg.kernels
        act = active.numpy().view(np.uint32)
        mask = ((act[:, None] >> np.arange(K, dtype=np.uint32)[None, :]) & 1).astype(bool)
        new = o.readmit(self._np(params), mask, self.ocfg, np.float32)
        act[:] = (new.astype(np.uint32) << np.arange(K, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)

    def checkpoint_best(self, loss, best_loss, params, best):
        better = loss < best_loss
        for k in NAMES:
            bm = better.reshape((-1,) + (1,) * (params[k].dim() - 1))
            best[k].copy_(torch.where(bm, params[k], best[k]))
        best_loss.copy_(torch.where(better, loss, best_loss))

    def reduce_scalars(self, loss, sse, active):
        N = self.cfg.pixels
        out = torch.zeros(3, dtype=torch.float64)
        if loss is not None:
            out[0] = loss.double().sum() * N
        if sse is not None:
            out[1] = sse.double().sum()
        if active is not None:
            out[2] = float(sum(bin(int(x) & 0xFFFFFFFF).count("1") for x in active.numpy().view(np.uint32)))
        return out

    def fit_variant(self, B):
        return "oracle"

    def set_total_blocks(self, total_blocks):
        self.total_blocks = int(total_blocks)       # the oracle has one summation order: nothing to choose


class OracleSharedEngine:
    """Test double of ``SharedEngine`` on the numpy oracle (shared-kernel mode), torch CPU tensors."""

    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.device = torch.device("cpu")
        self.ocfg = o.OracleConfig(
            block_shape=tuple(cfg.batch_shape), channels=cfg.channels, kernels=cfg.kernels, precision=cfg.precision,
            margin=cfg.margin, use_determinant=cfg.use_determinant, use_yuv=cfg.use_yuv, train_pis=cfg.train_pis,
            train_gammas=cfg.train_gammas, train_musx=cfg.train_musx, lr_expert=cfg.lr_expert, lr_pis=cfg.lr_pis,
            lr_steer=cfg.lr_steer, beta1=cfg.beta1, beta2=cfg.beta2, adam_eps=cfg.adam_eps,
            grad_clip=(cfg.grad_clip or None), pis_l1=cfg.pis_l1, u_l1=cfg.u_l1, start_pis=cfg.start_pis or cfg.kernels,
            quantization_mode=getattr(cfg, 'quantization_mode', 0), quantize_pis=getattr(cfg, 'quantize_pis', False),
            bit_depths=tuple(getattr(cfg, 'bit_depths', (20, 18, 6, 10, 10))),
            lower_bounds=tuple(getattr(cfg, 'lower_bounds', (-2500, -.3, -5, 0, -32))),
            upper_bounds=tuple(getattr(cfg, 'upper_bounds', (2500, 1.3, 5, 2, 32))),
            only_y_gamma=getattr(cfg, 'only_y_gamma', False), ssim_opt=getattr(cfg, 'ssim_opt', False),
            train_inverse_cov=getattr(cfg, 'train_inverse_cov', False), radial_as=getattr(cfg, 'radial_as', False),
            kernel_count_as_norm_l1=getattr(cfg, 'kernel_count_as_norm_l1', False))
        self.coords = o.global_batch_coords(cfg.image_shape, cfg.batch_shape)
        ov = int(getattr(cfg, "overlap", 0))
        self.halo = o.global_halo_coords(cfg.image_shape, cfg.batch_shape, ov) if ov > 0 else None
        self.num_batches = self.coords.shape[0]
        self.list_words = (cfg.kernels + 31) // 32
        self.batch_pixels = self.coords.shape[1]
        self._sizes = None
        self._buf = None

    def close(self):
        pass

    def new_adam_state(self, params):
        return _State(params, self.cfg.beta1, self.cfg.beta2)

    def new_lists(self, nb=None):
        nb = self.num_batches if nb is None else nb
        K, KW = self.cfg.kernels, self.list_words
        w = torch.full((nb, KW), -1, dtype=torch.int32)
        if K % 32:
            w[:, KW - 1] = (1 << (K % 32)) - 1
        return w

    def _mask(self, lists):
        bits = lists.numpy().view(np.uint32)
        K = self.cfg.kernels
        return np.stack([(bits[:, k >> 5] >> np.uint32(k & 31)) & 1 for k in range(K)], axis=1).astype(bool)

    def _setbits(self, lists, mask):
        bits = lists.numpy().view(np.uint32)
        bits[:] = 0
        for k in range(self.cfg.kernels):
            bits[:, k >> 5] |= (mask[:, k].astype(np.uint32) << np.uint32(k & 31))

    def _p(self, params):
        return {k: params[k].numpy()[None] for k in NAMES}

    def set_loss_weights(self, loss_w):
        self._loss_w = None if loss_w is None else loss_w.numpy()

    def set_center_grid(self, grid):
        self.ocfg.mus_grid = None if grid is None else grid.numpy()[None]

    def _lw(self, first_batch, nb):
        lw = getattr(self, "_loss_w", None)
        return None if lw is None else lw[first_batch:first_batch + nb]

    def _halo(self, first_batch, nb):
        return None if self.halo is None else self.halo[first_batch:first_batch + nb]

    def forward(self, target, params, lists, first_batch=0, want_recon=True, want_argmax=False, update_lists=True):
        nb = lists.shape[0]
        tgt = np.ascontiguousarray(target.numpy().transpose(0, 2, 1))
        f = o.shared_pass(self._p(params), tgt, self.coords[first_batch:first_batch + nb], self._mask(lists), self.ocfg,
                          halo_coords=self._halo(first_batch, nb), loss_w=self._lw(first_batch, nb))
        if update_lists:
            self._setbits(lists, f["lists_new"])
        return {"loss": torch.from_numpy(f["loss"].astype(np.float32)), "sse": torch.from_numpy(f["sse"].astype(np.float32)),
                "recon": torch.from_numpy(np.ascontiguousarray(f["recon"].transpose(0, 2, 1))) if want_recon else None,
                "argmax": torch.from_numpy(f["argmax"].astype(np.int32)) if want_argmax else None}

    def grad_buffer(self):
        if self._buf is None:
            n = sum(int(np.prod(s)) for s in self._shapes().values())
            self._buf = torch.zeros(n, dtype=torch.float64)
        return self._buf

    def _shapes(self):
        K, d, C = self.cfg.kernels, len(self.cfg.image_shape), self.cfg.channels
        return {"pis": (K,), "musX": (K, d), "A_diagonal": (K, d, d), "A_corr": (K, d, d), "gamma_e": (K, d, C), "nu_e": (K, C)}

    def accumulate(self, target, params, lists, first_batch=0, loss_out=None, sse_out=None):
        nb = lists.shape[0]
        tgt = np.ascontiguousarray(target.numpy().transpose(0, 2, 1))
        f = o.shared_pass(self._p(params), tgt, self.coords[first_batch:first_batch + nb], self._mask(lists), self.ocfg,
                          np.float32, want_grads=True, halo_coords=self._halo(first_batch, nb), loss_w=self._lw(first_batch, nb))
        self._setbits(lists, f["lists_new"])
        buf = self.grad_buffer()
        flat = np.concatenate([f["grads"][k][0].astype(np.float64).ravel() for k in NAMES])
        buf += torch.from_numpy(flat)
        if loss_out is not None:
            loss_out.copy_(torch.from_numpy(f["loss"].astype(np.float32)))
        if sse_out is not None:
            sse_out.copy_(torch.from_numpy(f["sse"].astype(np.float32)))

    def apply(self, params, state):
        buf = self.grad_buffer().numpy()
        grads, off = {}, 0
        for k, shp in self._shapes().items():
            n = int(np.prod(shp))
            grads[k] = buf[off:off + n].reshape((1,) + shp).astype(np.float32)
            off += n
        st = {"m": {k: state.m[k].numpy()[None] for k in NAMES}, "v": {k: state.v[k].numpy()[None] for k in NAMES},
              "t": state._step, "b1p": np.float32(state.c.beta1_power), "b2p": np.float32(state.c.beta2_power)}
        newp = o.adam_step(self._p(params), grads, st, self.ocfg, np.float32)
        for k in NAMES:
            params[k].copy_(torch.from_numpy(np.ascontiguousarray(newp[k][0])))
            state.m[k].copy_(torch.from_numpy(np.ascontiguousarray(st["m"][k][0])))
            state.v[k].copy_(torch.from_numpy(np.ascontiguousarray(st["v"][k][0])))
        state._step += 1
        state.c.beta1_power, state.c.beta2_power, state.c.step = float(st["b1p"]), float(st["b2p"]), state._step
        self._buf.zero_()

    def fit(self, target, params, state, lists, n_iters, loss_out=None, sse_out=None):
        for _ in range(n_iters):
            self.accumulate(target, params, lists, 0, loss_out, sse_out)
            self.apply(params, state)

    def update_kernel_list(self, params, lists, first_batch=0):
        nb = lists.shape[0]
        win = self.coords if self.halo is None else self.halo
        new = o.shared_readmit(self._p(params), self._mask(lists), win[first_batch:first_batch + nb], self.ocfg)
        self._setbits(lists, new)
