"""End-to-end wall time of the facade on BASELINE configs[1] (one 512x512 grayscale image, 16x16 blocks, K=4,
200 iterations, validation every 100) split into its phases, next to the device time of the fit kernels alone."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.smoe import Adam, Smoe


def main(side=512, iters=200, val=100):
    b = blk.synthetic_blocks((side // 16) ** 2, (16, 16), 1, 7)
    img = blk.blocks_to_image(b, (side, side), (16, 16))
    torch.zeros(1).cuda()
    torch.cuda.synchronize()
    for rep in range(2):                      # second repetition: warm caches / loaded library
        t = [time.perf_counter()]
        s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, train_inverse_cov=False,
                 quantize_pis=True)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
        torch.cuda.synchronize(); t.append(time.perf_counter())
        devnull = open(os.devnull, "w")
        old = sys.stdout
        sys.stdout = devnull
        s.train(iters, val_iter=val)
        sys.stdout = old
        torch.cuda.synchronize(); t.append(time.perf_counter())
        rec = s.get_reconstruction()
        p = s.get_params()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        names = ["construct", "set_optimizer", f"train({iters}, val_iter={val})", "get_reconstruction+get_params"]
        print(f"rep {rep}: " + ", ".join(f"{n} {1e3 * (t[i + 1] - t[i]):.1f} ms" for i, n in enumerate(names)),
              f"| total {1e3 * (t[-1] - t[0]):.1f} ms | psnr {s.get_psnr():.2f} dB")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
