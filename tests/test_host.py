"""CPU tests of the host-side plumbing (steered_mixture_of_experts_amd.blocks / dist): block
tiling in the reference's sliding_window order, the reference's initialisers, sharding."""
import numpy as np

from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd import dist as sdist


def _sliding_window(image, bs):
    """The reference's iteration order (smoe.py:18-35) with Overlap=0: y outer, x inner, t innermost."""
    out = []
    if image.ndim == 3:
        for y in range(0, image.shape[0], bs[0]):
            for x in range(0, image.shape[1], bs[1]):
                out.append(image[y:y + bs[0], x:x + bs[1], :])
    else:
        for y in range(0, image.shape[0], bs[0]):
            for x in range(0, image.shape[1], bs[1]):
                for z in range(0, image.shape[2], bs[2]):
                    out.append(image[y:y + bs[0], x:x + bs[1], z:z + bs[2], :])
    return np.stack(out)


def test_tiling_matches_sliding_window_order_and_roundtrips():
    rng = np.random.default_rng(0)
    img = rng.random((48, 32, 3)).astype(np.float32)
    b, v = blk.image_to_blocks(img, (16, 16))
    assert np.array_equal(b, _sliding_window(img, (16, 16))) and v.all()
    assert np.array_equal(blk.blocks_to_image(b, (48, 32), (16, 16)), img)
    vid = rng.random((32, 16, 8, 3)).astype(np.float32)
    b, v = blk.image_to_blocks(vid, (16, 16, 4))
    assert np.array_equal(b, _sliding_window(vid, (16, 16, 4)))
    assert np.array_equal(blk.blocks_to_image(b, (32, 16, 8), (16, 16, 4)), vid)
    pl = blk.to_planar(b)
    assert pl.shape == (4, 3, 1024) and np.array_equal(blk.from_planar(pl, (16, 16, 4)), b)


def test_padding_is_edge_replicated_with_zero_weight():
    rng = np.random.default_rng(1)
    img = rng.random((20, 37, 1)).astype(np.float32)          # 1080-like: not a multiple of the block
    b, v = blk.image_to_blocks(img, (16, 16))
    assert b.shape == (2 * 3, 16, 16, 1) and v.shape == (6, 256)
    assert v.sum() == 20 * 37
    full = blk.blocks_to_image(b, (32, 48), (16, 16))
    assert np.array_equal(full[:20, :37], img)
    assert np.array_equal(full[20:, :37], np.repeat(img[19:20], 12, axis=0))
    assert np.array_equal(blk.blocks_to_image(b, (20, 37), (16, 16)), img)
    vm = blk.blocks_to_image(v.reshape(6, 16, 16, 1), (32, 48), (16, 16))[..., 0]
    assert vm[:20, :37].all() and not vm[20:].any() and not vm[:, 37:].any()


def test_initialisers_match_the_oracle_restatement():
    for shape, C, kpd in [((16, 16), 1, [2, 2]), ((32, 32), 3, [2, 4]), ((16, 16, 4), 3, [2, 2, 1]), ((16, 16), 1, [3])]:
        b = blk.synthetic_blocks(5, shape, C, 3)
        p = blk.init_block_params(b, kpd)
        q = o.init_params(b, kpd)
        for k in p:
            assert p[k].dtype == np.float32 and np.array_equal(p[k], q[k]), k
    mus, A = blk.generate_kernel_grid([12], 2)
    assert mus.shape == (144, 2) and np.allclose(A[0], np.diag([26., 26.]))      # smoe.py:2158-2160


def test_synthetic_blocks_are_deterministic_uint8_lattice():
    a = blk.synthetic_blocks(300, (16, 16), 1, 20260002)
    b = blk.synthetic_blocks(300, (16, 16), 1, 20260002)
    assert np.array_equal(a, b) and a.dtype == np.float32
    k = np.round(a * 255)
    assert np.array_equal(a, (k.astype(np.uint8).astype(np.float32) / np.float32(255.)))
    assert a.min() >= 0 and a.max() <= 1


def test_shard_ranges_partition_the_blocks():
    for B in (0, 1, 7, 1024, 32400):
        for R in (1, 2, 3, 8):
            spans = [sdist.shard_range(B, r, R) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(R - 1))
            assert max(hi - lo for lo, hi in spans) <= -(-B // R) if B else True
    assert sdist.world() == (0, 1)


def test_get_batch_shape_divisor_search():
    # smoe.py:2459-2543: smallest batch count >= desired, most cube-like split, trailing axis untouched
    assert blk.get_batch_shape(1, (512, 512, 3)) == (512, 512, 3)
    assert blk.get_batch_shape(1024, (512, 512, 3)) == (16, 16, 3)
    assert blk.get_batch_shape(1000, (512, 512, 3)) == (16, 16, 3)          # 1024 is the next reachable count
    assert blk.get_batch_shape(2, (512, 512, 3)) in ((256, 512, 3), (512, 256, 3))
    assert blk.get_batch_shape(4, (512, 512, 3)) == (256, 256, 3)
    assert blk.get_batch_shape(6, (48, 36, 5)) == (16, 18, 5) or np.prod(np.array((48, 36)) / np.array(blk.get_batch_shape(6, (48, 36, 5))[:2])) == 6
    s = blk.get_batch_shape(8, (64, 64, 16, 6))
    assert s[-1] == 6 and (64 // s[0]) * (64 // s[1]) * (16 // s[2]) == 8 and s == (32, 32, 8, 6)
    s = blk.get_batch_shape(7, (30, 20, 4))                                   # 7 is not reachable: next is 8
    assert (30 // s[0]) * (20 // s[1]) == 8 and 30 % s[0] == 0 and 20 % s[1] == 0
