"""Host-side block plumbing: image <-> independent blocks, the reference's initialisers,
deterministic synthetic data.  NumPy only; runs once per fit (not the hot path).

Block order is the reference's ``sliding_window`` order (smoe.py:18-35): y outer, x inner,
t innermost.  Every block is its own [0,1]^d domain (smoe.py:2412), i.e. ``blocks[b]`` is what
``Smoe(block_b, ...)`` would see.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------
# tiling
# ------------------------------------------------------------------------------------
def padded_shape(domain_shape: Sequence[int], block_shape: Sequence[int]) -> Tuple[int, ...]:
    return tuple(int(-(-s // b) * b) for s, b in zip(domain_shape, block_shape))


def grid_shape(domain_shape: Sequence[int], block_shape: Sequence[int]) -> Tuple[int, ...]:
    return tuple(int(-(-s // b)) for s, b in zip(domain_shape, block_shape))


def image_to_blocks(image: np.ndarray, block_shape: Sequence[int]):
    """(H,W[,T],C) -> blocks (B, *block_shape, C) in sliding_window order, plus a validity
    mask (B, N) that is 0 on padding.  The reference raises when the image is not a multiple
    of the block (smoe.py:239-241); here the image is edge-replicated up to the next multiple
    and padded pixels get loss weight 0 (the graph's per-pixel ``loss_weights``, smoe.py:550,932)."""
    d = image.ndim - 1
    bs = tuple(int(b) for b in block_shape)
    assert len(bs) == d
    dom = image.shape[:d]
    pad = padded_shape(dom, bs)
    if pad != tuple(dom):
        widths = [(0, p - s) for s, p in zip(dom, pad)] + [(0, 0)]
        img = np.pad(image, widths, mode="edge")
        valid = np.pad(np.ones(dom, dtype=np.float32), widths[:-1], mode="constant")
    else:
        img = image
        valid = np.ones(dom, dtype=np.float32)
    g = grid_shape(dom, bs)
    C = image.shape[-1]
    # (g0,b0,g1,b1[,g2,b2],C) -> (g0,g1[,g2],b0,b1[,b2],C)
    split = []
    for gi, bi in zip(g, bs):
        split += [gi, bi]
    perm = list(range(0, 2 * d, 2)) + list(range(1, 2 * d, 2))
    blocks = img.reshape(split + [C]).transpose(perm + [2 * d]).reshape((-1,) + bs + (C,))
    vblk = valid.reshape(split).transpose(perm).reshape(-1, int(np.prod(bs)))
    return np.ascontiguousarray(blocks), np.ascontiguousarray(vblk)


def blocks_to_image(blocks: np.ndarray, domain_shape: Sequence[int], block_shape: Sequence[int]) -> np.ndarray:
    """Inverse of image_to_blocks (crops the padding): (B, *block_shape, X) -> (*domain_shape, X).
    Mirrors the stitch of smoe.py:1719-1744."""
    d = len(block_shape)
    bs = tuple(int(b) for b in block_shape)
    g = grid_shape(domain_shape, bs)
    tail = blocks.shape[1 + d:]
    arr = blocks.reshape(tuple(g) + bs + tail)
    perm = []
    for i in range(d):
        perm += [i, d + i]
    perm += list(range(2 * d, 2 * d + len(tail)))
    arr = arr.transpose(perm).reshape(tuple(gi * bi for gi, bi in zip(g, bs)) + tail)
    sl = tuple(slice(0, s) for s in domain_shape)
    return arr[sl]


def to_planar(blocks: np.ndarray) -> np.ndarray:
    """(B, *block_shape, C) -> (B, C, N): the C-ABI target layout (include/smoe_hip.h)."""
    B, C = blocks.shape[0], blocks.shape[-1]
    N = int(np.prod(blocks.shape[1:-1]))           # explicit: a rank may hold no blocks at all (B = 0)
    return np.ascontiguousarray(blocks.reshape(B, N, C).transpose(0, 2, 1))


def from_planar(arr: np.ndarray, block_shape: Sequence[int]) -> np.ndarray:
    """(B, C, N) -> (B, *block_shape, C)."""
    B, C = arr.shape[0], arr.shape[1]
    return np.ascontiguousarray(arr.transpose(0, 2, 1)).reshape((B,) + tuple(block_shape) + (C,))


def get_batch_shape(desired_batches: int, joint_domain_shape: Sequence[int]) -> Tuple[int, ...]:
    """Smoe.get_batch_shape (smoe.py:2459-2543): the block shape used when no ``batch_size`` is given.
    Every axis is divided by one of its divisors so that the number of batches is the smallest
    possible count >= ``desired_batches``; among those the divisor tuple with the smallest sum
    (most cube-like split) wins, first hit in the reference's enumeration order on ties.
    ``joint_domain_shape`` = image shape incl. the trailing (d + C) axis, which is never split."""
    def divisors(n):
        factors = {}
        nn, i = n, 2
        while i * i <= nn:
            while nn % i == 0:
                factors[i] = factors.get(i, 0) + 1
                nn //= i
            i += 1
        if nn > 1:
            factors[nn] = 1
        primes = list(factors.keys())

        def generate(k):
            if k == len(primes):
                yield 1
            else:
                for factor in generate(k + 1):
                    p_i = 1
                    for _ in range(factors[primes[k]] + 1):
                        yield factor * p_i
                        p_i *= primes[k]
        return list(generate(0))

    import itertools
    shape = [int(v) for v in joint_domain_shape]
    factors = [divisors(n) for n in shape[:-1]] + [[1]]
    if len(shape) > 4:                                   # light-field hack of the reference (smoe.py:2507-2509)
        factors[0] = [1]
        factors[1] = [1]
    shapes = list(itertools.product(*factors))
    possible = np.array([float(np.prod(sh[:-1])) for sh in shapes])
    diff = possible - desired_batches
    diff[diff < 0] = np.inf
    aimed = possible[int(np.argmin(diff))]
    cand = [shapes[i] for i in np.where(possible == aimed)[0]]
    sums = [np.sum(divs[2:3]) if len(divs) > 4 else np.sum(divs) for divs in cand]
    divs = cand[int(np.argmin(sums))]
    return tuple(int(shape[i] / divs[i]) for i in range(len(shape)))


# ------------------------------------------------------------------------------------
# the reference's initialisers, per block
# ------------------------------------------------------------------------------------
def gen_domain_grid(kernels_per_dim: Sequence[int], dim: int) -> np.ndarray:
    """Smoe.gen_domain for a list input (smoe.py:2402-2415,2424): kernel centres with equal
    spacing between positions and the border."""
    kpd = [int(k) for k in kernels_per_dim]
    if len(kpd) == 1:
        kpd = kpd * dim
    coord = [np.linspace((1 / n) / 2, 1 - (1 / n) / 2, n) for n in kpd]
    grids = np.meshgrid(*coord, indexing="ij")
    return np.reshape(np.stack(grids, axis=-1), (int(np.prod(kpd)), dim))


def generate_kernel_grid(kernels_per_dim: Sequence[int], dim: int, train_inverse_cov: bool = False):
    """smoe.py:2146-2163 -> (musX_init (K,d), A_init (K,d,d))."""
    kpd = [int(k) for k in kernels_per_dim]
    mus = gen_domain_grid(kpd, dim)
    if len(kpd) > 1:
        A_proto = np.diag([2.0 * (k + 1) for k in kpd])
        K = int(np.prod(kpd))
    else:
        A_proto = np.zeros((dim, dim))
        np.fill_diagonal(A_proto, 2 * (kpd[0] + 1))
        K = kpd[0] ** dim
    A = np.tile(A_proto, (K, 1, 1))
    if train_inverse_cov:
        A = A ** 2
    return mus, A


def generate_experts(blocks: np.ndarray, musX_init: np.ndarray) -> np.ndarray:
    """smoe.py:2165-2235 per block -> nu_e_init (B,K,C): mean of the block's pixels inside
    the [mu - mu0, mu + mu0) window of every kernel (Python banker's ``round``)."""
    B = blocks.shape[0]
    shape = blocks.shape[1:-1]
    d = len(shape)
    K = musX_init.shape[0]
    stride = musX_init[0]
    nu = np.empty((B, K, blocks.shape[-1]), dtype=np.float32)
    for k in range(K):
        sl: List[slice] = [slice(None)]
        for ax in range(d):
            lo = int(round((musX_init[k, ax] - stride[ax]) * shape[ax]))
            hi = int(round((musX_init[k, ax] + stride[ax]) * shape[ax]))
            sl.append(slice(lo, hi))
        nu[:, k, :] = np.mean(blocks[tuple(sl)], axis=tuple(range(1, d + 1)))
    return nu


def init_block_params(blocks: np.ndarray, kernels_per_dim: Sequence[int], normalize_pis: bool = True,
                      train_inverse_cov: bool = False) -> Dict[str, np.ndarray]:
    """What Smoe.__init__ builds when no init_params are given (smoe.py:260-262), for every
    block: float32 arrays in the get_params() layout with a leading block axis."""
    B = blocks.shape[0]
    d = blocks.ndim - 2
    C = blocks.shape[-1]
    mus, A = generate_kernel_grid(kernels_per_dim, d, train_inverse_cov)
    K = mus.shape[0]
    pis = np.ones((K,), dtype=np.float32)          # generate_pis, smoe.py:2237-2242
    if normalize_pis:
        pis = pis / K
    return {
        "pis": np.ascontiguousarray(np.tile(pis, (B, 1))),
        "musX": np.ascontiguousarray(np.tile(mus.astype(np.float32), (B, 1, 1))),
        "A_diagonal": np.ascontiguousarray(np.tile(A.astype(np.float32), (B, 1, 1, 1))),
        "A_corr": np.zeros((B, K, d, d), dtype=np.float32),          # smoe.py:437
        "gamma_e": np.zeros((B, K, d, C), dtype=np.float32),         # smoe.py:2170
        "nu_e": generate_experts(blocks, mus),
    }


# ------------------------------------------------------------------------------------
# deterministic synthetic inputs
# ------------------------------------------------------------------------------------
def synthetic_blocks(B: int, block_shape: Sequence[int], C: int, seed: int) -> np.ndarray:
    """Random oriented step edge + linear ramp + N(0,(2/255)^2) noise per block, clipped,
    rounded to uint8 and divided by 255 (as utils.py:126-128 does).  (B, *block_shape, C) float32."""
    rng = np.random.default_rng(seed)
    d = len(block_shape)
    axes = [np.linspace(0, 1, s) for s in block_shape]
    grids = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)
    normal = rng.normal(size=(B, d))
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    offset = rng.uniform(0.25, 0.75, size=(B,))
    lo = rng.uniform(0.1, 0.9, size=(B, C))
    hi = rng.uniform(0.1, 0.9, size=(B, C))
    slope = rng.uniform(-0.3, 0.3, size=(B, d, C))
    out = np.empty((B,) + tuple(block_shape) + (C,), dtype=np.float32)
    step = 4096
    for s in range(0, B, step):
        e = min(B, s + step)
        nb = e - s
        proj = np.tensordot(grids - 0.5, normal[s:e].T, axes=([d], [0]))          # (*shape, nb)
        proj = np.moveaxis(proj, -1, 0) + 0.5
        side = (proj > offset[s:e].reshape((nb,) + (1,) * d)).astype(np.float64)
        ex = (nb,) + (1,) * d + (C,)
        img = lo[s:e].reshape(ex) * (1 - side[..., None]) + hi[s:e].reshape(ex) * side[..., None]
        img = img + np.einsum("...l,blc->b...c", grids - 0.5, slope[s:e])
        img = img + rng.normal(scale=2 / 255, size=img.shape)
        img = np.clip(img, 0, 1)
        out[s:e] = np.round(img * 255).astype(np.uint8).astype(np.float32) / np.float32(255.)
    return out


def psnr(mse, precision):
    """plotter.py:14-15."""
    return 10 * np.log10((2 ** precision) ** 2 / mse)
