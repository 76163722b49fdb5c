"""CPU restatement of the reference's parameter quantiser (quantize_params + rescaler, quantizer.py:4-145;
reduce_params, utils.py:7-16) for ONE block at a time in plain numpy float64.  TEST INFRASTRUCTURE ONLY: the checker
of the vectorised ``steered_mixture_of_experts_amd.quantizer``; nothing in the product imports it.  Pinned: bit-exact
against vectors produced by the reference's own quantize_params / rescaler (tests/golden/ref_quantizer.npz,
tests/test_reference_golden.py)."""
import numpy as np


def quantize_block(params, bit_depths, mode=0, quantize_pis=False, lower_bounds=None, upper_bounds=None):
    idx = params['pis'] > 0                                              # reduce_params
    p = {k: np.asarray(v, np.float64)[idx] for k, v in params.items()}
    lb, ub = {}, {}
    for name, b in (('A_diagonal', 0), ('A_corr', 0), ('musX', 1), ('nu_e', 2), ('gamma_e', 4)):
        if mode <= 1 or mode == 3:
            lb[name] = np.amin(p[name], axis=0, keepdims=True)
            ub[name] = np.amax(p[name], axis=0, keepdims=True)
        else:
            lb[name] = np.ones((1,) + p[name].shape[1:]) * lower_bounds[b]
            ub[name] = np.ones((1,) + p[name].shape[1:]) * upper_bounds[b]
    if mode <= 1 and not quantize_pis:
        lb['pis'] = np.amin(p['pis'], axis=0, keepdims=True)
        ub['pis'] = np.amax(p['pis'], axis=0, keepdims=True)
    else:
        lb['pis'] = np.ones((1,)) * lower_bounds[3]
        ub['pis'] = np.ones((1,)) * upper_bounds[3]
    steps = {'A_diagonal': 2 ** bit_depths[0] - 1, 'A_corr': 2 ** bit_depths[0] - 1, 'musX': 2 ** bit_depths[1] - 1,
             'nu_e': 2 ** bit_depths[2] - 1, 'pis': 2 ** bit_depths[3] - 1, 'gamma_e': 2 ** bit_depths[4] - 1}
    q = {k: np.round((p[k] - lb[k]) / (ub[k] - lb[k] + 10e-12) * steps[k]) for k in steps}
    r = {k: q[k] / steps[k] * (ub[k] - lb[k]) + lb[k] for k in steps}
    r['A'] = r['A_diagonal'] + r['A_corr']
    return idx, q, r
