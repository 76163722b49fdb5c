"""Decode-side entry point (reference smoe_reconstruction.py:15-79): parameter pickle ->
``Smoe(init_params=...)`` -> ONE forward pass -> reconstructed image.

    python -m steered_mixture_of_experts_amd.smoe_reconstruction -i IMG -r OUT -p params.pkl
"""
import argparse
import os
import re

import numpy as np

from . import blocks as blk
from .smoe import SharedSmoe, Smoe
from .utils import load_checkpoint, read_image, write_image


def _decode_batch_shape(image_shape, channels, batches):
    """Batch shape for decoding a whole-image (shared-kernel) model: the reference's divisor search
    (smoe.py:2459-2543) with the smallest batch count >= ``batches`` whose batches fit the kernels' LDS tile.  Every
    batch lists every kernel at the start (smoe.py:315), so the split does not change the reconstruction."""
    limit = 2048 if channels == 1 else 1024
    d = len(image_shape)
    n = max(1, int(batches))
    total = int(np.prod(image_shape))
    while n <= total:
        bs = blk.get_batch_shape(n, tuple(image_shape) + (d + channels,))[:-1]
        if int(np.prod(bs)) <= limit:
            return [int(b) for b in bs]
        n = int(np.prod([s // b for s, b in zip(image_shape, bs)])) + 1
    raise ValueError("no batch shape fits")


_shared_engine_factory = None      # tests put the oracle-backed engine here; None = the HIP engine


def main(image_path, results_path, params_file, batches=1, bit_depths=(20, 18, 6, 10, 10), quant_params=False):
    if len(bit_depths) != 5:
        raise ValueError("Number of bit depths must be five!")           # smoe_reconstruction.py:17-18
    orig, precision, _ = read_image(image_path)
    cp = load_checkpoint(params_file)
    init_params = cp['params']
    if results_path is not None and not os.path.exists(results_path):
        os.mkdir(results_path)
    # the graph the model was trained on (smoe_reconstruction.py:32-43 copies these from the pickle onto the model; here
    # they go through the constructor so that the kernels are built for them).  Absent keys keep the defaults.
    qmode = int(cp.get('quantization_mode') or 0)
    common = dict(init_params=init_params, bit_depths=list(bit_depths), precision=precision,
                  use_determinant=bool(cp.get('use_determinant', True)), use_yuv=bool(cp.get('use_yuv', False)),
                  train_inverse_cov=bool(cp.get('train_inverse_cov', False)),    # absent key: trained by the CLI (False)
                  radial_as=bool(cp.get('radial_as', False)), quantization_mode=qmode,
                  quantize_pis=bool(cp.get('quantized_pis', False)))
    for key in ('lower_bounds', 'upper_bounds'):
        if cp.get(key) is not None:
            common[key] = list(cp[key])
    if np.asarray(init_params['pis']).ndim == 1:
        # ONE model for the whole image = the reference's own checkpoint layout (utils.save_model, kernels leading;
        # with reduce=True only the kernels with pis > 0) and this package's --mode shared pickles
        d = orig.ndim - 1
        bs = cp.get('batch_size') or _decode_batch_shape(orig.shape[:d], orig.shape[-1], batches)
        smoe = SharedSmoe(orig, batch_size=list(bs), only_y_gamma=bool(cp.get('only_y_gamma', False)),
                          use_diff_center=False, engine_factory=_shared_engine_factory, **common)
    else:
        smoe = Smoe(orig, start_batches=batches, batch_size=list(cp['batch_size']), **common)
    with_q = bool(quant_params) and smoe.quantization_mode <= 0         # smoe_reconstruction.py:46-51
    if with_q:
        from .quantizer import quantize_params, rescaler
        smoe.qparams = quantize_params(smoe, smoe.get_params())
        smoe.rparams = rescaler(smoe, smoe.qparams)
    loss, mse, _, _ = smoe.run_batched(train=False, update_reconstruction=True, with_quantized_params=with_q)
    found = re.findall(r'\d+', os.path.basename(params_file))
    iter_str = found[-1] if found else "0"
    reconstruction_path = results_path + '/' + iter_str + "_reconstruction"
    if with_q:                                                          # smoe_reconstruction.py:58-75
        reconstruction = smoe.get_qreconstruction()
        reconstruction_path += "_{0:1d}_{1:1d}_{2:1d}_{3:1d}_{4:1d}".format(*bit_depths)
    else:
        reconstruction = smoe.get_reconstruction()
    write_image(reconstruction, reconstruction_path, smoe.dim_domain, smoe.use_yuv, precision)
    return reconstruction, loss, mse


def _cli():
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--image_path', type=str, required=True, help="input image")
    parser.add_argument('-r', '--results_path', type=str, required=True, help="results path")
    parser.add_argument('-p', '--params_file', type=str, required=True, help="parameter file for model initialization.")
    parser.add_argument('-b', '--batches', type=int, default=1)
    parser.add_argument('-bd', '--bit_depths', type=int, default=[20, 18, 6, 10, 10], nargs='+')
    parser.add_argument('-qp', '--quant_params', action='store_true')
    args = parser.parse_args()
    main(**vars(args))


if __name__ == '__main__':
    _cli()
