"""A fixed-seed slice of the randomised sweeps in scripts/fuzz_parity.py / scripts/fuzz_shared.py: random block / image
shapes and option combinations, one evaluation pass and one fit step per case against the fp64 restatement."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def _sweep(mod, n, seed):
    rng = np.random.default_rng(seed)
    failed, refused, checked = [], 0, 0
    for i in range(n):
        desc, msg = mod.one_case(rng, i)
        if msg and msg.startswith("refused"):
            refused += 1
            assert "ssim_opt" in desc and "do not fit" in msg, (desc, msg)      # the only legitimate refusal in this space
        elif msg:
            failed.append((desc, msg))
        else:
            checked += 1
    assert not failed, failed
    assert checked >= n - 3


def test_block_mode_random_configurations():
    import fuzz_parity
    _sweep(fuzz_parity, 60, 2026)


def test_shared_mode_random_configurations():
    import fuzz_shared
    _sweep(fuzz_shared, 60, 2027)
