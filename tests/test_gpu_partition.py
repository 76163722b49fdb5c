"""Partition invariance on the HIP path.

The reference walks all blocks of an image in ONE host loop (smoe.py:1643-1702): what a block comes out as cannot
depend on how many other blocks the pass holds.  Here the lanes-per-block tiling -- hence the order in which a block's
gradient terms are summed -- follows the number of blocks, so a shard of an image would round differently from the whole
image.  ``smoe_set_total_blocks`` makes every call choose its kernels for the WHOLE job: with AUTOMATIC tiling the fit of
an image as one batch and as R sequential shards (dist.shard_range, what R ranks would hold) must be bit-identical in
parameters, Adam slots, reconstruction, loss and kernel lists.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.dist import shard_range
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fit(eng, T, p0, K, n_iters):
    dev = "cuda"
    dp = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in p0.items()}
    st = eng.new_adam_state(dp)
    act = torch.full((T.shape[0],), (1 << K) - 1, dtype=torch.int32, device=dev)
    f0 = eng.forward(T, dp, act, want_recon=False)                    # iteration-0 pass (prunes the lists)
    eng.fit(T, dp, st, act, n_iters, loss0=f0["loss"])
    eng.update_kernel_list(dp, act)
    out = eng.forward(T, dp, act, want_recon=True)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in dp.items()}
    res.update({"m_" + k: v.cpu().numpy() for k, v in st.m.items()})
    res.update({"v_" + k: v.cpu().numpy() for k, v in st.v.items()})
    res.update(recon=out["recon"].cpu().numpy(), loss=out["loss"].cpu().numpy(), sse=out["sse"].cpu().numpy(),
               active=act.cpu().numpy())
    return res


@pytest.mark.parametrize("name,B,shape,C,kpd,ranks,n_iters", [
    ("cfg4", 32400, (16, 16), 3, [2, 2], 8, 12),            # 4K RGB: one batch -> 16 lanes per block; 4 050-block shards alone -> 32
    ("cfg2", 1024, (16, 16), 1, [2, 2], 8, 25),             # ONE 512x512 image: block over two wavefronts; 128-block shards too
    ("mid", 6000, (16, 16), 1, [2, 2], 4, 12),              # 32-lane batch whose shards alone would take the 64-lane kernel
    ("cfg5-part", 4080, (16, 16, 4), 3, [2, 2, 1], 4, 6),   # 1 024-pixel blocks: shards of 1 020 would run the two-wavefront form
    ("ragged-ranks", 1030, (16, 16), 1, [2, 2], 7, 10),     # ceil split with a short last shard
])
def test_one_batch_equals_sequential_shards_with_automatic_tiling(name, B, shape, C, kpd, ranks, n_iters):
    K = int(np.prod(kpd))
    b = blk.synthetic_blocks(B, shape, C, 20260400 + B)
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    p0 = blk.init_block_params(b, kpd)
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=(C == 3), quantize_pis=True))   # CLI defaults
    per = shard_range(B, 0, ranks)[1]
    # the premise: left to themselves the shards would take another kernel than the whole job
    local_variant, whole_variant = eng.fit_variant(per), eng.fit_variant(B)
    local_occ, whole_occ = eng.fit_occupancy(per), eng.fit_occupancy(B)
    eng.set_total_blocks(B)
    assert eng.fit_variant(per) == whole_variant == eng.fit_variant(1)
    whole = _fit(eng, T, p0, K, n_iters)
    parts = []
    for r in range(ranks):
        lo, hi = shard_range(B, r, ranks)
        parts.append(_fit(eng, T[lo:hi].contiguous(), {k: v[lo:hi] for k, v in p0.items()}, K, n_iters))
    for key in whole:
        got = np.concatenate([p[key] for p in parts], axis=0)
        assert got.shape == whole[key].shape
        assert np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got,
                              whole[key].view(np.uint32) if got.dtype == np.float32 else whole[key]), (name, key)
    assert np.isfinite(whole["loss"]).all()
    # ... and it is this setting that does it: the same shards with the per-call choice differ in the last bits (where the
    # per-call choice is another kernel at all)
    eng.set_total_blocks(0)
    if (local_variant, local_occ) != (whole_variant, whole_occ) or name in ("cfg2", "cfg5-part"):
        lo, hi = shard_range(B, 0, ranks)
        loc = _fit(eng, T[lo:hi].contiguous(), {k: v[lo:hi] for k, v in p0.items()}, K, n_iters)
        same = all(np.array_equal(loc[k], whole[k][lo:hi]) for k in ("nu_e", "A_diagonal", "musX"))
        if name == "cfg4":
            assert local_variant != whole_variant and not same
    eng.close()


def test_facade_tells_the_engine_the_whole_image():
    """``Smoe`` hands the engine the block count of the whole image, so a rank's shard runs the whole image's kernels."""
    from steered_mixture_of_experts_amd.smoe import Smoe
    img = np.random.default_rng(3).uniform(size=(64, 64, 1)).astype(np.float32)
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], train_inverse_cov=False, use_determinant=True)
    assert s.num_blocks == 16
    # 16 blocks in all: whatever count a call holds, the kernel is the one 16 blocks take
    assert s._engine.fit_variant(100000) == s._engine.fit_variant(16)
    s._engine.set_total_blocks(0)
    assert s._engine.fit_variant(100000) != s._engine.fit_variant(16)


def _bench(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra +
                         ["--steps", "10", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--clock-warm-iters", "0", "--no-reps"],
                         env=env, timeout=900, stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stdout
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_strong_scaling_is_bit_identical_at_one_and_two_ranks():
    """`bench.py --scaling strong` on the HIP engine at 1 rank and at 2 ranks sharing this box's GPU (gloo rendezvous):
    the digest of the fitted state (wrapping sum of all parameter bit patterns and kernel lists over all ranks) and the
    all-reduced PSNR are equal; with `--tiling-scope local` every rank takes the kernel of its own shard instead."""
    img = ["--scaling", "strong", "--image", "1040", "1600"]       # 65 x 100 = 6 500 blocks -> 32 lanes; halves of 3 250 too, quarters not
    one = _bench(["--gpus", "1"] + img)
    two = _bench(["--gpus", "2", "--backend", "gloo"] + img, {"SMOE_BENCH_SHARE_GPU": "1"})
    assert one["config"]["total_blocks"] == two["config"]["total_blocks"] == 6500
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and len(two["config"]["kernel_variant_per_rank"]) == 2
    assert two["config"]["kernel_variant_per_rank"][0].split()[0] == one["config"]["kernel_variant_per_rank"][0].split()[0]
    assert one["state_digest"] == two["state_digest"]
    assert one["final_psnr_db"] == two["final_psnr_db"] and one["diverged_blocks"] == two["diverged_blocks"]
    small = ["--scaling", "strong", "--image", "512", "1024"]       # 2 048 blocks: 64 lanes; halves of 1 024: block over two wavefronts (duo)
    one_s = _bench(["--gpus", "1"] + small)
    two_s = _bench(["--gpus", "2", "--backend", "gloo"] + small, {"SMOE_BENCH_SHARE_GPU": "1"})
    two_l = _bench(["--gpus", "2", "--backend", "gloo", "--tiling-scope", "local"] + small, {"SMOE_BENCH_SHARE_GPU": "1"})
    assert one_s["state_digest"] == two_s["state_digest"]
    assert "duo64w2" in two_l["config"]["kernel_variant_per_rank"][0] and "duo" not in two_s["config"]["kernel_variant_per_rank"][0]
    assert two_l["state_digest"] != one_s["state_digest"]          # the two-wavefront tiling of a 1 024-block shard sums in another order
