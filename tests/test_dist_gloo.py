"""world_size-2 CPU test (gloo) of the multi-GPU path: blocks are sharded across ranks with no
data-path collective; the 3-scalar all-reduce gives the same global loss / MSE / kernel count
as a single process, and every block's fitted parameters are bit-identical to the
single-process fit (block-partition invariance)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from fake_engine import OracleEngine
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.smoe import Adam, Smoe
ws = int(os.environ.get("WORLD_SIZE", "1"))
if ws > 1:
    dist.init_process_group(backend="gloo")
b = blk.synthetic_blocks(15, (16, 16), 1, 99)
img = blk.blocks_to_image(b, (48, 80), (16, 16))
s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine)
s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
s.train(6, val_iter=3)
out = {"params": s.get_params(), "losses": s.get_losses(), "mses": s.get_mses(), "num_pis": s.get_num_pis(),
       "recon": s.get_reconstruction(), "span": (s.lo, s.hi), "argmax": s.get_weight_matrix_argmax()}
# shared-kernel mode: batches sharded, accumulated gradient buffer all-reduced before the Adam step
from fake_engine import OracleSharedEngine
from steered_mixture_of_experts_amd.smoe import SharedSmoe
g = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True, engine_factory=OracleSharedEngine)
g.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
g.train(4, val_iter=2)
out["shared"] = {"params": g.get_params(), "losses": g.get_losses(), "recon": g.get_reconstruction(),
                 "lists": np.array(g.kernel_list_per_batch), "span": (g.lo, g.hi)}
# the same with image-wide quantities: quantization_mode 3 ranges / routing and pis_l1 / count(qpis > 0)
q = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True,
               engine_factory=OracleSharedEngine, quantization_mode=3, quantize_pis=True, kernel_count_as_norm_l1=True,
               bit_depths=[14, 12, 8, 10, 10], lower_bounds=[-60, -.3, -1, 0, -4], upper_bounds=[60, 1.3, 2, 2, 4])
q.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
q.train(4, val_iter=2, pis_l1=0.5)
out["shared_q3"] = {"params": q.get_params(), "losses": q.get_losses()}
if ws == 1 or dist.get_rank() == 0:
    pickle.dump(out, open(sys.argv[2], "wb"))
if ws > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    one = str(tmp_path / "one.pkl")
    subprocess.check_call([sys.executable, str(w), ROOT, one], env=env, timeout=300)
    two = str(tmp_path / "two.pkl")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(w), ROOT, two],
                          env=env, timeout=600)
    a, b = pickle.load(open(one, "rb")), pickle.load(open(two, "rb"))
    assert a["span"] == (0, 15) and b["span"] == (0, 8)
    for k in a["params"]:
        assert np.array_equal(a["params"][k], b["params"][k]), k
    assert np.array_equal(a["recon"], b["recon"]) and np.array_equal(a["argmax"], b["argmax"])
    assert [i for i, _ in a["losses"]] == [i for i, _ in b["losses"]] == [0, 3, 6]
    assert np.allclose([v for _, v in a["losses"]], [v for _, v in b["losses"]], rtol=1e-12)
    assert np.allclose([v for _, v in a["mses"]], [v for _, v in b["mses"]], rtol=1e-12)
    assert a["num_pis"] == b["num_pis"]
    sa, sb = a["shared"], b["shared"]
    assert sa["span"] == (0, 15) and sb["span"] == (0, 8)
    for k in sa["params"]:                       # the all-reduced gradient sum differs only in summation order
        assert np.allclose(sa["params"][k], sb["params"][k], rtol=1e-5, atol=1e-6), k
    assert np.allclose([v for _, v in sa["losses"]], [v for _, v in sb["losses"]], rtol=1e-5)
    assert (np.abs(sa["recon"] - sb["recon"]) < 1.5 / 255).mean() > 0.999
    qa, qb = a["shared_q3"], b["shared_q3"]
    for k in qa["params"]:
        assert np.allclose(qa["params"][k], qb["params"][k], rtol=1e-5, atol=1e-6), k
    assert np.allclose([v for _, v in qa["losses"]], [v for _, v in qb["losses"]], rtol=1e-5)
