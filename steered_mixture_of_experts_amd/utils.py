"""Model / image IO in the reference's formats (reference utils.py:18-162).  The pickle
schema written by ``save_model`` and read by ``load_params`` is the reference's
(``{'params', 'mses', 'losses', 'num_pis', flags...}``, utils.py:30-59) with the per-block
leading axis kept; image codecs (cv2, hdf5storage) are not rebuilt -- ``read_image`` handles
.npy / .npz arrays and, when Pillow is importable, ordinary image files."""
import pickle

import numpy as np


def psnr(mse, precision):
    """plotter.py:14-15: mse is mse_op, i.e. mean(diff^2) * (2^p)^2 (smoe.py:1053)."""
    return 10 * np.log10((2 ** precision) ** 2 / mse)


def save_model(smoe, path, best=False, reduce=False, quantize=False):
    """utils.py:18-59.  ``reduce`` (dropping pis<=0 kernels) would make the per-block arrays
    ragged and is therefore off by default (the quantiser keeps a ``used_kernels`` mask
    instead); ``quantize`` stores ``smoe.qparams`` with the reference's metadata keys."""
    params = smoe.get_best_params() if best else smoe.get_params()
    shared = np.asarray(params['pis']).ndim == 1                      # ONE model for the image (SharedSmoe)
    if reduce and not shared:
        raise NotImplementedError("reduce=True would drop the block structure of per-block parameters")
    if reduce:                                                       # utils.py:21-28: keep the kernels with pis > 0
        keep = np.asarray(params['pis']) > 0
        params = {k: np.asarray(v)[keep] for k, v in params.items()}
    cp = {'params': params, 'mses': smoe.get_mses(), 'losses': smoe.get_losses(), 'num_pis': smoe.get_num_pis(),
          'quantization_mode': smoe.quantization_mode, 'quantized_pis': smoe.quantize_pis,
          'lower_bounds': smoe.lower_bounds, 'upper_bounds': smoe.upper_bounds,
          'use_yuv': smoe.use_yuv, 'only_y_gamma': smoe.only_y_gamma, 'ssim_opt': smoe.ssim_opt,
          'use_determinant': smoe.use_determinant, 'use_diff_center': smoe.use_diff_center,
          # additions needed to rebuild the block tiling and the form of the kernels (the reference's Smoe defaults to
          # train_inverse_cov=True while its CLI trains with False, smoe.py:41 / smoe_test.py:342)
          'train_inverse_cov': bool(getattr(smoe, 'train_inverse_cov', False)),
          'radial_as': bool(getattr(smoe, 'radial_as', False)), 'bit_depths': list(smoe.bit_depths),
          'batch_size': tuple(smoe.batch_size_valued), 'shape_of_img': tuple(smoe.image.shape),
          'mode': 'shared' if shared else 'blocks'}
    if quantize and smoe.qparams is not None:                        # utils.py:37-56
        qparams = dict(smoe.qparams)
        qparams.update({'dim_of_domain': smoe.dim_domain, 'dim_of_output': smoe.image.shape[-1],
                        'shape_of_img': smoe.image.shape[:-1], 'used_ranges': False,
                        'quantized_tria_params': True, 'trained_gamma': smoe.train_gammas,
                        'trained_musx': smoe.train_musx, 'radial_as': smoe.radial_as,
                        'trained_pis': smoe.train_pis, 'use_yuv': smoe.use_yuv,
                        'only_y_gamma': smoe.only_y_gamma, 'use_determinant': smoe.use_determinant,
                        'use_diff_center': smoe.use_diff_center})
        cp.update({'qparams': qparams})
    if smoe.rank == 0:
        with open(path, 'wb') as fd:
            pickle.dump(cp, fd)
    return cp


def load_params(path):
    """utils.py:61-65."""
    with open(path, 'rb') as fd:
        return pickle.load(fd)['params']


def load_checkpoint(path):
    with open(path, 'rb') as fd:
        return pickle.load(fd)


def read_image(path, use_yuv=True):
    """utils.py:68-134 for array inputs: returns (float32 image in [0,1], precision, affines)."""
    if path.lower().endswith('.npy'):
        orig = np.load(path)
    elif path.lower().endswith('.npz'):
        orig = np.load(path)["imgs"]
    else:
        try:
            from PIL import Image
        except ImportError as e:                                     # pragma: no cover
            raise ValueError("Unknown data format (Pillow is not available for image files)") from e
        orig = np.asarray(Image.open(path))
    if orig.ndim == 2:
        orig = orig[..., None]
    if orig.dtype == np.uint8:
        orig = orig.astype(np.float32) / 255.                        # utils.py:126-128
        precision = 8
    elif orig.dtype == np.uint16:
        orig = orig.astype(np.float32) / 2 ** 16.                    # utils.py:129-131
        precision = 16
    else:
        orig = orig.astype(np.float32)
        precision = 8
    return orig, precision, None


def write_image(img, path, type=2, yuv=False, precision=8):
    """utils.py:136-162, array output: the rounded integer image as .npy (and .png when
    Pillow is importable and the image is 2-D)."""
    if precision == 8:
        out = np.uint8(np.round(img * 255))
    else:
        out = np.uint16(np.round(img * 2 ** precision))
    np.save(path + ".npy", out)
    if type == 2:
        try:
            from PIL import Image
            Image.fromarray(np.squeeze(out)).save(path + ".png")
        except Exception:                                            # pragma: no cover
            pass
    return out
