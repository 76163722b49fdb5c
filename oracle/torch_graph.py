"""Independent torch restatement of the reference TF graph for ONE block, written
op-for-op from smoe.py:732-937,1012-1053 (band_part assembly, the 4-operand
einsum, boolean_mask of every parameter, straight-through fake-quant), used only
to cross-check the numpy oracle's forward values and analytic gradients through
``torch.autograd``.  Test infrastructure; not part of the product path.
"""
import math

import torch


class _FakeQuant01(torch.autograd.Function):
    """tf.quantization.fake_quant_with_min_max_args(x, 0, 1, num_bits) with its
    registered gradient (pass-through inside [nudged_min, nudged_max])."""

    @staticmethod
    def forward(ctx, x, bits):
        levels = float(2 ** bits - 1)
        scale = torch.tensor(1.0, dtype=x.dtype) / levels
        inv = 1.0 / scale
        nmax = levels * scale
        ctx.save_for_backward((x >= 0) & (x <= nmax))
        cl = torch.clamp(x, min=0.0, max=float(nmax))
        return torch.floor(cl * inv + 0.5) * scale

    @staticmethod
    def backward(ctx, g):
        (inside,) = ctx.saved_tensors
        return g * inside.to(g.dtype), None


def _nudge(mn, mx, bits):
    """TF fake_quant Nudge(): scale, nudged_min, nudged_max for the range [mn, mx] (tensors or floats)."""
    qmin, qmax = 0.0, float(2 ** bits - 1)
    scale = (mx - mn) / (qmax - qmin)
    zp = qmin - mn / scale
    nzp = torch.where(zp < qmin, torch.full_like(zp, qmin),
                      torch.where(zp > qmax, torch.full_like(zp, qmax), torch.sign(zp) * torch.floor(zp.abs() + 0.5)))
    return scale, (qmin - nzp) * scale, (qmax - nzp) * scale


class _FakeQuantVars(torch.autograd.Function):
    """tf.quantization.fake_quant_with_min_max_vars(x, min, max, num_bits) and its registered gradient: the
    input gradient passes inside [nudged_min, nudged_max]; what falls below goes to ``min``, above to ``max``.
    min == max == 0 gives zeros (fake_quant_ops_functor.h)."""

    @staticmethod
    def forward(ctx, x, mn, mx, bits):
        if float(mn) == 0.0 and float(mx) == 0.0:
            ctx.zero = True
            ctx.save_for_backward(x)
            return torch.zeros_like(x)
        ctx.zero = False
        scale, nmin, nmax = _nudge(mn.detach(), mx.detach(), bits)
        ctx.save_for_backward(x, nmin, nmax)
        cl = torch.minimum(torch.maximum(x, nmin), nmax)
        return torch.floor((cl - nmin) * (1.0 / scale) + 0.5) * scale + nmin

    @staticmethod
    def backward(ctx, g):
        if ctx.zero:                      # "If min and max are both zero, we propagate everything to inputs."
            return g, torch.zeros(()), torch.zeros(()), None
        x, nmin, nmax = ctx.saved_tensors
        below, above = x < nmin, x > nmax
        between = ~(below | above)
        return g * between.to(g.dtype), (g * below.to(g.dtype)).sum(), (g * above.to(g.dtype)).sum(), None


def _fake_quant_args(x, mn, mx, bits):
    """fake_quant_with_min_max_args: constant range (no gradient to it)."""
    return _FakeQuantVars.apply(x, torch.tensor(float(mn), dtype=x.dtype), torch.tensor(float(mx), dtype=x.dtype), bits)


def quantize_graph_params(params, *, quantization_mode, quantize_pis, bit_depths, lower_bounds, upper_bounds,
                          train_musx=True, radial_as=False):
    """smoe.py:474-538: the fake-quantised views of the variables the graph is built on.
    bit_depths / bounds order: A, musX, nu_e, pis, gamma_e (smoe_test.py:302-309)."""
    pis_v, musX_v = params["pis"], params["musX"]
    Ad_v, Ac_v, gam_v, nu_v = params["A_diagonal"], params["A_corr"], params["gamma_e"], params["nu_e"]
    lb, ub, bd = lower_bounds, upper_bounds, bit_depths
    if quantization_mode >= 2 or quantize_pis:
        qpis = _fake_quant_args(pis_v, lb[3], ub[3], bd[3])
    else:
        qpis = pis_v
    pis_mask = qpis > 0
    if quantization_mode == 2:
        qAd = _fake_quant_args(Ad_v, lb[0], ub[0], bd[0])
        qAc = _fake_quant_args(Ac_v, lb[0], ub[0], bd[0])
        qmu = _fake_quant_args(musX_v, lb[1], ub[1], bd[1])
        qnu = _fake_quant_args(nu_v, lb[2], ub[2], bd[2])
        qga = _fake_quant_args(gam_v, lb[4], ub[4], bd[4])
    elif quantization_mode == 3:
        if radial_as:                     # smoe.py:498-504: the (K,) variable, NOT shifted by its minimum
            mn, mx = Ad_v[pis_mask].min(), Ad_v[pis_mask].max()
            qAd = _FakeQuantVars.apply(Ad_v, torch.zeros_like(mn), mx - mn, bd[0]) + mn
        else:
            dg = torch.diagonal(Ad_v[pis_mask], dim1=-2, dim2=-1)
            mn, mx = dg.min(), dg.max()
            qAd = _FakeQuantVars.apply(Ad_v - mn, torch.zeros_like(mn), mx - mn, bd[0]) + mn
        qAc = _FakeQuantVars.apply(Ac_v, Ac_v[pis_mask].min(), Ac_v[pis_mask].max(), bd[0])
        qmu = _FakeQuantVars.apply(musX_v, musX_v[pis_mask].min(), musX_v[pis_mask].max(), bd[1]) if train_musx else musX_v
        mn, mx = nu_v[pis_mask].min(), nu_v[pis_mask].max()
        qnu = _FakeQuantVars.apply(nu_v - mn, torch.zeros_like(mn), mx - mn, bd[2]) + mn
        qga = _FakeQuantVars.apply(gam_v, gam_v[pis_mask].min(), gam_v[pis_mask].max(), bd[4])
    else:
        qAd, qAc, qmu, qnu, qga = Ad_v, Ac_v, musX_v, nu_v, gam_v
    return {"pis": qpis, "musX": qmu, "A_diagonal": qAd, "A_corr": qAc, "gamma_e": qga, "nu_e": qnu}, pis_mask


def custom_ssim_2d(img1, img2):
    """ops/image_ops_impl.py:77-233 for (H,W,C) inputs with max_val = 1: 11x11 Gaussian (sigma 1.5, built with
    a softmax), VALID depthwise correlation, mean of luminance * contrast-structure per channel."""
    dt = img1.dtype
    c = torch.arange(11, dtype=dt) - 5.0
    g = (c ** 2) * (-0.5 / 1.5 ** 2)
    g = (g.reshape(1, -1) + g.reshape(-1, 1)).reshape(1, -1)
    kernel = torch.softmax(g, dim=1).reshape(1, 1, 11, 11)
    C = img1.shape[-1]
    kern = kernel.repeat(C, 1, 1, 1)

    def reducer(x):
        return torch.nn.functional.conv2d(x.permute(2, 0, 1).unsqueeze(0), kern, groups=C)[0].permute(1, 2, 0)

    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mean0, mean1 = reducer(img1), reducer(img2)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1 = reducer(img1 * img2) * 2.0
    den1 = reducer(img1 ** 2 + img2 ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return torch.mean(luminance * cs, dim=(0, 1))


def custom_ssim_3d(img1, img2):
    """ops/image_ops_impl.py:77-233 with ndim = 3 for (T0,T1,T2,C) inputs: the 11x11x11 Gaussian (softmax over the sum of
    the three squared offsets), conv3d per channel (the reference moves the channels to the batch axis,
    image_ops_impl.py:214-215), VALID, mean of luminance * contrast-structure over the three axes per channel."""
    dt = img1.dtype
    c = torch.arange(11, dtype=dt) - 5.0
    g = (c ** 2) * (-0.5 / 1.5 ** 2)
    g3 = (g.reshape(1, 1, -1) + g.reshape(1, -1, 1) + g.reshape(-1, 1, 1)).reshape(1, -1)
    kernel = torch.softmax(g3, dim=1).reshape(1, 1, 11, 11, 11)

    def reducer(x):
        return torch.nn.functional.conv3d(x.permute(3, 0, 1, 2).unsqueeze(1), kernel)[:, 0].permute(1, 2, 3, 0)

    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mean0, mean1 = reducer(img1), reducer(img2)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1 = reducer(img1 * img2) * 2.0
    den1 = reducer(img1 ** 2 + img2 ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return torch.mean(luminance * cs, dim=(0, 1, 2))


def _symmetric_pad(img, pad):
    """tf.pad(..., "SYMMETRIC") on the two leading axes of (H,W,C)."""
    import numpy as np
    ri = torch.from_numpy(np.pad(np.arange(img.shape[0]), pad, mode="symmetric"))
    ci = torch.from_numpy(np.pad(np.arange(img.shape[1]), pad, mode="symmetric"))
    return img[ri][:, ci]


def tf_graph_block(params, coords, target, kernel_list, *, precision=8, margin=0.5,
                   use_determinant=True, use_yuv=False, train_gammas=True,
                   pis_l1=0.0, u_l1=0.0, start_pis=None, loss_w=None, ssim_opt=False, block_shape=None,
                   quantization_mode=0, quantize_pis=False, bit_depths=(20, 18, 6, 10, 10),
                   lower_bounds=(-2500, -.3, -5, 0, -32), upper_bounds=(2500, 1.3, 5, 2, 32), train_musx=True,
                   train_inverse_cov=False, radial_as=False, kernel_count_as_norm_l1=False, musX_grid=None):
    """params: dict of torch tensors (K,), (K,d), (K,d,d), (K,d,d), (K,d,C), (K,C)
    with requires_grad; coords (N,d); target (N,C); kernel_list (K,) bool.
    Returns dict(loss, mse_op, res (N,C), w_e (Ka,N), indices)."""
    qp, pis_mask = quantize_graph_params(params, quantization_mode=quantization_mode, quantize_pis=quantize_pis,
                                         bit_depths=bit_depths, lower_bounds=lower_bounds, upper_bounds=upper_bounds,
                                         train_musx=train_musx, radial_as=radial_as)
    pis_v, musX_v = qp["pis"], qp["musX"]
    if musX_grid is not None:                                                    # use_diff_center, smoe.py:390-394,746-747
        musX_v = musX_v + musX_grid
    Ad_v, Ac_v = qp["A_diagonal"], qp["A_corr"]
    gam_v, nu_v = qp["gamma_e"], qp["nu_e"]
    K, d = musX_v.shape
    C = nu_v.shape[1]
    N = coords.shape[0]
    dt = coords.dtype
    # smoe.py:732-733  band_part(A_diagonal,0,0) + band_part(set_diag(A_corr,0),-1,0)
    if radial_as:                                                                # smoe.py:714-719: A_diagonal is (K,)
        Ad_v = Ad_v.reshape(-1, 1, 1).repeat(1, d, d)
    A = torch.diag_embed(torch.diagonal(Ad_v, dim1=-2, dim2=-1)) + torch.tril(Ac_v, diagonal=-1)
    if train_inverse_cov:                                                        # smoe.py:734-735
        A = A + torch.tril(Ac_v, diagonal=-1).transpose(-1, -2)
    # smoe.py:480,738-753
    bool_mask = kernel_list & pis_mask
    indices = torch.arange(K)[bool_mask]
    musX, nu_e, gamma_e, A, pis = musX_v[bool_mask], nu_v[bool_mask], gam_v[bool_mask], A[bool_mask], pis_v[bool_mask]
    # smoe.py:777-782,796
    x_sub_mu = (coords.unsqueeze(0) - musX.unsqueeze(1)).unsqueeze(-1)          # (K,N,d,1)
    if train_inverse_cov:                                                        # smoe.py:791-793
        maha = torch.einsum("abli,alm,abmj->ab", x_sub_mu, A, x_sub_mu)
    else:
        maha = torch.einsum("abli,alm,anm,abnj->ab", x_sub_mu, A, A, x_sub_mu)
    n_exp = torch.exp(-0.5 * maha)                                               # smoe.py:807
    if use_determinant:                                                          # smoe.py:809-815
        n_div = torch.prod(torch.diagonal(A, dim1=-2, dim2=-1), dim=-1)
        n_quo = n_div / math.sqrt((2 * math.pi) ** d)
        Nk = n_quo.unsqueeze(1) * n_exp
    else:
        Nk = n_exp
    n_w = Nk * pis.unsqueeze(-1)                                                 # smoe.py:819
    n_w_norm = torch.sum(n_w, dim=0)
    n_w_norm = torch.maximum(torch.tensor(10e-12, dtype=dt), n_w_norm)           # smoe.py:821
    w_e = n_w / n_w_norm
    infl = (w_e > 0.5 / (2 ** precision)).to(dt)                                 # smoe.py:825-826
    w_e = w_e * infl
    klb = infl.sum(dim=1) > 0                                                    # smoe.py:829
    nu_t = nu_e.t().unsqueeze(-1)                                                # (C,K,1)
    if train_gammas:                                                             # smoe.py:841-846
        dom = coords.t().unsqueeze(0).repeat(C, 1, 1)                            # (C,d,N)
        sloped = torch.matmul(gamma_e.permute(2, 0, 1), dom)                     # (C,K,N)
        res = torch.sum(w_e * (sloped + nu_t), dim=1)
    else:
        res = torch.sum(w_e * nu_t, dim=1)
    pre = res.t()
    res = torch.clamp(res, 0.0, 1.0).t()                                         # smoe.py:857-858
    res = _FakeQuant01.apply(res, precision)                                     # smoe.py:899
    diff = res - target
    mse = torch.mean(diff ** 2)
    eps = margin / (2 ** precision)
    lw = torch.ones((N, 1), dtype=dt) if loss_w is None else loss_w.reshape(N, 1)
    lp = torch.clamp((diff.abs() - eps) ** 2, min=0.0) * lw                      # smoe.py:932
    if use_yuv:
        loss_pixel = 6 / 8 * lp[:, 0].mean() + 1 / 8 * lp[:, 1:].mean(dim=0).sum()
    else:
        loss_pixel = lp.mean()
    if ssim_opt:                                                                 # smoe.py:980-1011
        if len(block_shape) == 3:                                                # smoe.py:999-1003
            import numpy as np
            r3, t3 = res.reshape(tuple(block_shape) + (C,)), target.reshape(tuple(block_shape) + (C,))
            for ax, b in enumerate(block_shape):                                 # tf.pad(..., "SYMMETRIC") on the three axes
                ii = torch.from_numpy(np.pad(np.arange(b), 5, mode="symmetric"))
                r3, t3 = r3.index_select(ax, ii), t3.index_select(ax, ii)
            ssim_per_channel = custom_ssim_3d(r3, t3)
        else:
            bh, bw = block_shape
            r2 = _symmetric_pad(res.reshape(bh, bw, C), 5)
            t2 = _symmetric_pad(target.reshape(bh, bw, C), 5)
            ssim_per_channel = custom_ssim_2d(r2, t2)
        if use_yuv:
            ssim = torch.sum(ssim_per_channel * torch.tensor([6.0, 1.0, 1.0], dtype=dt)) / 8
        else:
            ssim = torch.mean(ssim_per_channel)
        loss_pixel = 1 - ssim
    k0 = K if start_pis is None else start_pis
    if kernel_count_as_norm_l1:                                                  # smoe.py:1012,1022-1023
        k0 = float(torch.count_nonzero(pis_mask))
    loss = loss_pixel + pis_l1 * pis.sum() / k0 + u_l1 * torch.diagonal(A, dim1=-2, dim2=-1).sum()
    return {"loss": loss, "mse_op": mse * (2 ** precision) ** 2, "res": res, "pre": pre,
            "w_e": w_e, "indices": indices[klb], "bool_mask": bool_mask}
