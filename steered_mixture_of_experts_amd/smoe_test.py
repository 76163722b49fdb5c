"""Training driver mirroring the reference's CLI (reference smoe_test.py:19-356; despite its name it
is the experiment driver, not a test): same flags and defaults for everything on the hot path,
three Adam optimizers with ``base_lr``, ``base_lr/lr_div``, ``base_lr*lr_mult``
(smoe_test.py:84-88), ``train``, then ``params_best.pkl`` / ``params_last.pkl`` (smoe_test.py:248-249).

    python -m steered_mixture_of_experts_amd.smoe_test -i IMG.npy -r OUT -k 2 -bz 16 16 -n 200

``--mode blocks`` (default): every ``-bz`` block is an independent model with ``-k`` kernels per axis
(the per-block hot path).  ``--mode shared``: the reference's whole-image fit -- ``-k`` is the GLOBAL
kernel grid and ``-bz`` the pixel batch of a pass.  Flags of features that are not built (kernel
adding, support vectors, motion models; batch overlap outside
``--mode shared``, SSIM outside ``--mode blocks``) are
accepted for command-line compatibility but must keep their inactive values.
"""
import argparse
import os
import shutil

import numpy as np

from .smoe import Adam, SharedSmoe, Smoe
from .utils import load_params, read_image, save_model, write_image


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('-i', '--image_path', type=str, required=True, help="input image (.npy/.npz array or an image file)")
    p.add_argument('-r', '--results_path', type=str, required=True, help="results path")
    p.add_argument('-n', '--iterations', type=int, default=10000)
    p.add_argument('-v', '--validation_iterations', type=int, default=100)
    p.add_argument('-k', '--kernels_per_dim', type=int, default=[12], nargs='+')
    p.add_argument('-p', '--params_file', type=str, default=None)
    p.add_argument('-reg', '--l1reg', type=float, default=0)
    p.add_argument('-lr', '--base_lr', type=float, default=0.001)
    p.add_argument('-b', '--batches', type=int, default=1)
    p.add_argument('-bz', '--batch_size', type=int, default=[None], nargs='+')
    p.add_argument('-d', '--lr_div', type=float, default=100)
    p.add_argument('-m', '--lr_mult', type=float, default=1000)
    p.add_argument('-dp', '--disable_train_pis', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-dg', '--disable_train_gammas', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-dm', '--disable_train_musx', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-udc', '--use_diff_center', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-ud', '--use_determinant', type=str2bool, nargs='?', const=True, default=True)
    p.add_argument('-np', '--normalize_pis', type=str2bool, nargs='?', const=True, default=True)
    p.add_argument('-qm', '--quantization_mode', type=int, default=0)
    p.add_argument('-bd', '--bit_depths', type=int, default=[20, 18, 6, 10, 10], nargs='+')
    p.add_argument('-qp', '--quantize_pis', type=str2bool, nargs='?', const=True, default=True)
    p.add_argument('-lb', '--lower_bounds', type=float, default=[-2500, -.3, -5, 0, -32], nargs='+')
    p.add_argument('-ub', '--upper_bounds', type=float, default=[2500, 1.3, 5, 2, 32], nargs='+')
    p.add_argument('-yuv', '--use_yuv', type=str2bool, nargs='?', const=True, default=True)
    p.add_argument('-oyg', '--only_y_gamma', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-ukl', '--update_kernel_list_iterations', type=int, default=None)
    # --- accepted for compatibility, must stay inactive --------------------------------------------
    p.add_argument('-ni', '--iterations_inc', type=int, default=1000)
    p.add_argument('-na', '--iterations_all', type=int, default=1000)
    p.add_argument('-is', '--inc_steps', type=int, default=0, help="kernel adding is not built (reference default 100)")
    p.add_argument('-tr', '--threshold_rel', type=float, default=0.2)
    p.add_argument('-c', '--checkpoint_path', type=str, default=None)
    p.add_argument('-msv', '--lr_mult_sv', type=float, default=1)
    p.add_argument('-ra', '--radial_as', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-ssim', '--ssim_opt', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-sp', '--sampling_percentage', type=int, default=100)
    p.add_argument('-ovl', '--overlap_of_batches', type=int, default=0)
    p.add_argument('-svreg', '--svreg', type=float, default=0)
    p.add_argument('-hpc', '--hpc_mode', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-cis', '--current_inc_step', type=int, default=0)
    p.add_argument('-kcn', '--kernel_count_norm_l1', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-tvs', '--train_svs', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-tt', '--train_trafo', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-npm', '--num_params_model', type=int, default=6)
    p.add_argument('-tiv', '--train_inverse_cov', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-if', '--init_flag', type=float, default=1)
    p.add_argument('-orfc', '--only_rec_from_checkpoint', type=str2bool, nargs='?', const=False, default=False)
    p.add_argument('-mask', '--loss_mask_path', type=str, default=None)
    # --- this implementation ---------------------------------------------------------------------
    p.add_argument('--mode', choices=['blocks', 'shared'], default='blocks',
                   help="blocks: independent per-block models (hot path); shared: global kernels (reference image fit)")
    return p


def main(args):
    if len(args.bit_depths) != 5:
        raise ValueError("Number of bit depths must be five!")                        # smoe_test.py:24-25
    inactive = {"inc_steps": 0,
                "svreg": 0, "hpc_mode": False,
                "train_svs": False, "train_trafo": False,
                "only_rec_from_checkpoint": False, "checkpoint_path": None}
    for name, val in inactive.items():
        if getattr(args, name) != val:
            raise NotImplementedError(f"--{name}={getattr(args, name)!r}: this feature is outside the per-block hot path "
                                      "(SURVEY section 8) and is not built")
    if args.sampling_percentage != 100 and args.mode == 'shared':
        raise NotImplementedError("--sampling_percentage is built for --mode blocks")
    if args.overlap_of_batches and args.mode != 'shared':
        raise NotImplementedError("--overlap_of_batches needs --mode shared (independent blocks have no neighbours)")
    if args.quantization_mode >= 2:                                                   # smoe_test.py:36-37
        args.quantize_pis = True
    orig, precision, _ = read_image(args.image_path, args.use_yuv)                    # smoe_test.py:39
    use_yuv = args.use_yuv and orig.shape[-1] == 3                                    # smoe_test.py:41-44
    only_y_gamma = args.only_y_gamma and use_yuv
    init_params = load_params(args.params_file) if args.params_file is not None else None
    if args.results_path is not None:                                                 # smoe_test.py:51-54
        if os.path.exists(args.results_path):
            shutil.rmtree(args.results_path)
        os.mkdir(args.results_path)
    loss_mask = np.load(args.loss_mask_path)["loss_mask"] if args.loss_mask_path else None
    kpd = list(args.kernels_per_dim)
    if len(kpd) == 1:
        kpd = [kpd[0]] * len(orig.shape[:-1])                                         # smoe_test.py:62-63
    common = dict(init_params=init_params, train_pis=not args.disable_train_pis,
                  train_gammas=not args.disable_train_gammas, train_musx=not args.disable_train_musx,
                  start_batches=args.batches, batch_size=args.batch_size, use_determinant=args.use_determinant,
                  normalize_pis=args.normalize_pis, use_yuv=use_yuv, precision=precision)
    if args.mode == 'blocks':
        smoe = Smoe(orig, kpd, use_diff_center=args.use_diff_center, quantization_mode=args.quantization_mode,
                    bit_depths=args.bit_depths, quantize_pis=args.quantize_pis,
                    lower_bounds=args.lower_bounds, upper_bounds=args.upper_bounds, only_y_gamma=only_y_gamma,
                    loss_mask=loss_mask, ssim_opt=args.ssim_opt, train_inverse_cov=args.train_inverse_cov,
                    radial_as=args.radial_as, kernel_count_as_norm_l1=args.kernel_count_norm_l1, **common)
    else:
        smoe = SharedSmoe(orig, kpd, overlap_of_batches=args.overlap_of_batches, only_y_gamma=only_y_gamma,
                          use_diff_center=args.use_diff_center, ssim_opt=args.ssim_opt, quantization_mode=args.quantization_mode,
                          quantize_pis=args.quantize_pis, bit_depths=args.bit_depths, lower_bounds=args.lower_bounds,
                          upper_bounds=args.upper_bounds, train_inverse_cov=args.train_inverse_cov,
                          radial_as=args.radial_as, loss_mask=loss_mask,
                          kernel_count_as_norm_l1=args.kernel_count_norm_l1, **common)
    optimizer1 = Adam(args.base_lr)                                                   # smoe_test.py:84-86
    optimizer2 = Adam(args.base_lr / args.lr_div)
    optimizer3 = Adam(args.base_lr * args.lr_mult)
    smoe.set_optimizer(optimizer1, optimizer2, optimizer3)
    if args.iterations != 0:
        extra = {"sampling_percentage": args.sampling_percentage} if args.mode == 'blocks' else {}
        smoe.train(args.iterations, val_iter=args.validation_iterations, ukl_iter=args.update_kernel_list_iterations,
                   pis_l1=args.l1reg, **extra)                                        # smoe_test.py:119-121
    # both modes write the reference's checkpoint schema (utils.save_model; smoe_test.py:248-249)
    quant = args.quantization_mode != 0
    save_model(smoe, args.results_path + "/params_best.pkl", best=True, quantize=quant)
    save_model(smoe, args.results_path + "/params_last.pkl", best=False, quantize=quant)
    rec = smoe.get_reconstruction()
    if smoe.rank == 0:
        write_image(rec, os.path.join(args.results_path, "reconstruction"), smoe.dim_domain, use_yuv, precision)
        print("PSNR: %.3f dB" % smoe.get_psnr())
    return smoe


if __name__ == '__main__':
    main(build_parser().parse_args())
