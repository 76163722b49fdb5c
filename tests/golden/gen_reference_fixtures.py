"""Generator of tests/golden/ref_*.npz: golden vectors produced by the REFERENCE'S OWN CODE.

Run in the build container only (it reads /root/reference, which never travels):

    python tests/golden/gen_reference_fixtures.py

The reference cannot be imported (``import smoe`` needs tensorflow / skimage / cv2, ``import utils`` needs cv2 /
hdf5storage -- all absent, nothing is stubbed).  Its host-side numerics, however, are plain numpy functions that do not
touch those libraries.  This script parses the reference files, takes the definitions of exactly those functions out
of the syntax tree, compiles them UNCHANGED and calls them on seeded inputs; inputs and outputs are stored as data.

Functions executed (reference file:line at the surveyed revision):
    smoe.py:18-35        sliding_window
    smoe.py:2146-2163    Smoe.generate_kernel_grid
    smoe.py:2165-2235    Smoe.generate_experts
    smoe.py:2237-2242    Smoe.generate_pis
    smoe.py:2395-2426    Smoe.gen_domain
    smoe.py:2428-2438    Smoe.calc_intervals
    smoe.py:2459-2543    Smoe.get_batch_shape
    quantizer.py:4-86    quantize_params
    quantizer.py:88-144  rescaler
    utils.py:7-16        reduce_params
    utils.py:18-59       save_model  (writes tests/golden/ref_checkpoint.pkl through the reference's own pickle schema)
    plotter.py:14-15     psnr
and, WITHOUT executing anything, the argparse defaults of smoe_test.py:262-352 (read off the syntax tree into
tests/golden/ref_cli_defaults.json).

What stays unpinned: everything that runs inside TensorFlow (the graph, its gradients, Adam) -- see DESIGN.md section 5.
"""
import ast
import itertools
import os
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _functions(path, names, cls=None):
    """{name: function} compiled from the definitions found in ``path`` (module level, or inside class ``cls``)."""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    defs = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {d.name for d in defs} == set(names), (path, names)
    for d in defs:
        d.decorator_list = []                      # staticmethod: called as plain functions here
    mod = ast.Module(body=defs, type_ignores=[])
    ns = {"np": np, "product": itertools.product}
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)
    return {n: ns[n] for n in names}, ns


def _image(shape, C, seed):
    """8-bit-valued smooth-plus-noise test picture in [0, 1] (inputs are data, stored in the fixture)."""
    rng = np.random.default_rng(seed)
    axes = np.meshgrid(*[np.linspace(0, 1, s) for s in shape], indexing="ij")
    base = sum((i + 1) * a for i, a in enumerate(axes)) / sum(range(1, len(shape) + 1))
    img = np.stack([np.clip(0.15 + 0.7 * base * (0.6 + 0.4 * np.cos(3.0 * (c + 1) * axes[0])) +
                            0.05 * rng.standard_normal(shape), 0, 1) for c in range(C)], axis=-1)
    return (np.round(img * 255) / 255).astype(np.float64)


def gen_init():
    f, _ = _functions("smoe.py", ["gen_domain", "generate_kernel_grid", "generate_experts", "generate_pis"], cls="Smoe")
    out = {}
    cases = [("g2", (32, 48), 1, [2, 2], True, False), ("g34", (32, 48), 1, [3, 4], False, False),
             ("rgb", (48, 32), 3, [4], True, True), ("vid", (16, 16, 8), 3, [2, 2, 2], True, False),
             ("b16", (16, 16), 1, [2, 2], True, False), ("odd", (30, 50), 3, [3, 5], True, False)]
    for name, shape, C, kpd, norm, ic in cases:
        img = _image(shape, C, 11 + len(out))
        me = types.SimpleNamespace(image=img, dim_domain=len(shape), train_inverse_cov=ic, musX_init=None)
        me.gen_domain = f["gen_domain"]
        joint = f["gen_domain"](img, len(shape))
        f["generate_kernel_grid"](me, kpd)
        f["generate_experts"](me)
        f["generate_pis"](me, norm)
        out.update({f"{name}.image": img, f"{name}.kpd": np.array(kpd), f"{name}.normalize_pis": np.array(norm),
                    f"{name}.train_inverse_cov": np.array(ic), f"{name}.joint_domain": joint,
                    f"{name}.musX_init": me.musX_init, f"{name}.A_init": me.A_init, f"{name}.nu_e_init": me.nu_e_init,
                    f"{name}.gamma_e_init": me.gamma_e_init, f"{name}.pis_init": me.pis_init})
    np.savez_compressed(os.path.join(OUT, "ref_init.npz"), **out)


def gen_windows():
    f, _ = _functions("smoe.py", ["sliding_window"])
    g, _ = _functions("smoe.py", ["gen_domain", "calc_intervals", "get_batch_shape"], cls="Smoe")
    out = {}
    for name, shape, C, ov, bs in [("img", (32, 48), 1, 0, (16, 16)), ("halo", (32, 48), 3, 4, (16, 24)),
                                   ("vid", (16, 16, 8), 1, 0, (8, 8, 4)), ("vidhalo", (16, 16, 8), 1, 2, (8, 16, 4))]:
        joint = g["gen_domain"](_image(shape, C, 5), len(shape))
        coords, wins = zip(*f["sliding_window"](joint, ov, bs))
        out.update({f"{name}.joint_domain": joint, f"{name}.overlap": np.array(ov), f"{name}.batch": np.array(bs),
                    f"{name}.coords": np.stack(coords), f"{name}.windows": np.stack(wins)})
    shapes = [(512, 512, 3), (1080, 1920, 5), (2160, 3840, 5), (1080, 1920, 30, 6), (48, 32, 3), (30, 50, 5), (17, 19, 3)]
    want = [1, 2, 4, 7, 16, 60, 1024, 2040]
    res = np.array([[list(g["get_batch_shape"](w, s)) + [0] * (4 - len(s)) for w in want] for s in shapes])
    out.update({"gbs.shapes": np.array([list(s) + [0] * (4 - len(s)) for s in shapes]), "gbs.want": np.array(want), "gbs.result": res})
    iv = [(n, b) for n in (16, 30, 1080) for b in (1, 3, 4, 7)]
    out.update({"ci.args": np.array(iv), "ci.result": np.array([g["calc_intervals"](n, b) + [(0, 0)] * (7 - max(2, b + 1) + 1) for n, b in iv])})
    np.savez_compressed(os.path.join(OUT, "ref_windows.npz"), **out)


def gen_quantizer():
    fq, ns = _functions("quantizer.py", ["quantize_params", "rescaler"])
    fu, _ = _functions("utils.py", ["reduce_params"])
    ns["reduce_params"] = fu["reduce_params"]
    fp, _ = _functions("plotter.py", ["psnr"])
    out = {}
    rng = np.random.default_rng(2026)
    i = 0
    for d, C, K in ((2, 1, 9), (2, 3, 16), (3, 3, 8)):
        for mode, qpis in ((0, False), (1, True), (2, True), (3, True), (1, False)):
            p = {"pis": rng.uniform(-0.02, 0.4, K), "musX": rng.uniform(-0.1, 1.1, (K, d)),
                 "A_diagonal": np.stack([np.diag(rng.uniform(2, 40, d)) for _ in range(K)]),
                 "A_corr": np.stack([np.tril(rng.normal(0, 5, (d, d)), -1) for _ in range(K)]),
                 "nu_e": rng.uniform(-0.2, 1.2, (K, C)), "gamma_e": rng.normal(0, 1.5, (K, d, C))}
            p["pis"][1] = 0.0
            smoe = types.SimpleNamespace(quantization_mode=mode, quantize_pis=qpis, radial_as=False, dim_domain=d,
                                         image=np.zeros((4,) * d + (C,)), bit_depths=[20, 18, 6, 10, 10],
                                         lower_bounds=[-2500, -.3, -5, 0, -32], upper_bounds=[2500, 1.3, 5, 2, 32],
                                         use_diff_center=False)
            raw = {k: v.copy() for k, v in p.items()}
            q = fq["quantize_params"](smoe, p)
            r = fq["rescaler"](smoe, q)
            tag = f"q{i}"
            out.update({f"{tag}.mode": np.array(mode), f"{tag}.quantize_pis": np.array(qpis)})
            out.update({f"{tag}.in.{k}": v for k, v in raw.items()})
            out.update({f"{tag}.q.{k}": q[k] for k in ("A_diagonal", "A_corr", "musX", "nu_e", "pis", "gamma_e")})
            out.update({f"{tag}.lb.{k}": np.asarray(v) for k, v in q["lower_bounds"].items()})
            out.update({f"{tag}.ub.{k}": np.asarray(v) for k, v in q["upper_bounds"].items()})
            out.update({f"{tag}.steps.{k}": np.asarray(v) for k, v in q["steps"].items()})
            out.update({f"{tag}.r.{k}": v for k, v in r.items()})
            i += 1
    out["ncases"] = np.array(i)
    # radial_as: A_diagonal is a (K,) vector, A_corr is not quantised, rescaler returns A = a * I (quantizer.py:12-14,128-133)
    j = 0
    for d, C, K in ((2, 1, 9), (3, 3, 8)):
        for mode, qpis in ((0, False), (1, True), (2, True)):
            p = {"pis": rng.uniform(-0.02, 0.4, K), "musX": rng.uniform(-0.1, 1.1, (K, d)),
                 "A_diagonal": rng.uniform(2, 40, K), "A_corr": np.zeros((K, d, d)),
                 "nu_e": rng.uniform(-0.2, 1.2, (K, C)), "gamma_e": rng.normal(0, 1.5, (K, d, C))}
            p["pis"][2] = 0.0
            smoe = types.SimpleNamespace(quantization_mode=mode, quantize_pis=qpis, radial_as=True, dim_domain=d,
                                         image=np.zeros((4,) * d + (C,)), bit_depths=[20, 18, 6, 10, 10],
                                         lower_bounds=[-2500, -.3, -5, 0, -32], upper_bounds=[2500, 1.3, 5, 2, 32],
                                         use_diff_center=False)
            raw = {k: v.copy() for k, v in p.items()}
            q = fq["quantize_params"](smoe, p)
            r = fq["rescaler"](smoe, q)
            assert "A_corr" not in q
            tag = f"rq{j}"
            out.update({f"{tag}.mode": np.array(mode), f"{tag}.quantize_pis": np.array(qpis)})
            out.update({f"{tag}.in.{k}": v for k, v in raw.items()})
            out.update({f"{tag}.q.{k}": q[k] for k in ("A_diagonal", "musX", "nu_e", "pis", "gamma_e")})
            out.update({f"{tag}.r.{k}": v for k, v in r.items()})
            j += 1
    out["nradial"] = np.array(j)
    mse = np.array([0.5, 12.25, 650.0, 4000.0])
    out.update({"psnr.mse": mse, "psnr.p8": fp["psnr"](mse, 8), "psnr.p10": fp["psnr"](mse, 10)})
    np.savez_compressed(os.path.join(OUT, "ref_quantizer.npz"), **out)


def gen_checkpoint():
    """A checkpoint written by the reference's save_model (reduce=True, quantize=True as smoe_test.py:248-249 call it)
    for a whole-image model built by the reference's own initialisers; the image is stored next to it."""
    import pickle
    fi, _ = _functions("smoe.py", ["gen_domain", "generate_kernel_grid", "generate_experts", "generate_pis"], cls="Smoe")
    fq, nsq = _functions("quantizer.py", ["quantize_params", "rescaler"])
    fu, nsu = _functions("utils.py", ["reduce_params", "save_model"])
    nsq["reduce_params"] = fu["reduce_params"]
    nsu["pickle"] = pickle
    img = _image((32, 48), 3, 77)
    me = types.SimpleNamespace(image=img, dim_domain=2, train_inverse_cov=False, musX_init=None, gen_domain=fi["gen_domain"])
    fi["generate_kernel_grid"](me, [3, 4])
    fi["generate_experts"](me)
    fi["generate_pis"](me, True)
    rng = np.random.default_rng(99)
    K = me.musX_init.shape[0]
    params = {"pis": me.pis_init.copy(), "musX": (me.musX_init + rng.normal(0, 0.01, me.musX_init.shape)).astype(np.float32),
              "A_diagonal": (me.A_init * rng.uniform(0.8, 1.3, (K, 1, 1))).astype(np.float32),
              "A_corr": np.stack([np.tril(rng.normal(0, 2.0, (2, 2)), -1) for _ in range(K)]).astype(np.float32),
              "nu_e": me.nu_e_init.astype(np.float32), "gamma_e": rng.normal(0, 0.3, (K, 2, 3)).astype(np.float32)}
    params["pis"][5] = 0.0                       # dropped by reduce_params
    params["pis"][7] = -0.01
    smoe = types.SimpleNamespace(
        quantization_mode=1, quantize_pis=True, radial_as=False, dim_domain=2, image=img, bit_depths=[20, 18, 6, 10, 10],
        lower_bounds=[-2500, -.3, -5, 0, -32], upper_bounds=[2500, 1.3, 5, 2, 32], use_diff_center=False, use_yuv=True,
        only_y_gamma=False, ssim_opt=False, use_determinant=True, train_trafo=False, affines=None, train_gammas=True,
        train_musx=True, train_pis=True,
        get_params=lambda: {k: v.copy() for k, v in params.items()}, get_best_params=lambda: {k: v.copy() for k, v in params.items()},
        get_mses=lambda: [(0, 812.5), (100, 95.25)], get_losses=lambda: [(0, 0.031), (100, 0.0042)],
        get_num_pis=lambda: [(0, 12), (100, 10)])
    smoe.qparams = fq["quantize_params"](smoe, smoe.get_params())
    fu["save_model"](smoe, os.path.join(OUT, "ref_checkpoint.pkl"), best=False, reduce=True, quantize=True)
    np.savez_compressed(os.path.join(OUT, "ref_checkpoint_inputs.npz"), image=img, **{"p." + k: v for k, v in params.items()})


def gen_cli_defaults():
    """dest -> default of every add_argument call in the reference's smoe_test.py, literal-evaluated from the source."""
    import json
    tree = ast.parse(open(os.path.join(REF, "smoe_test.py")).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            flags = [a.value for a in node.args if isinstance(a, ast.Constant)]
            dest = max(flags, key=len).lstrip("-")
            kw = {k.arg: k.value for k in node.keywords}
            entry = {"flags": flags}
            if "default" in kw:
                try:
                    entry["default"] = ast.literal_eval(kw["default"])
                except ValueError:
                    entry["default"] = ast.unparse(kw["default"])
            if "required" in kw:
                entry["required"] = ast.literal_eval(kw["required"])
            if "type" in kw:
                entry["type"] = ast.unparse(kw["type"])
            out[dest] = entry
    with open(os.path.join(OUT, "ref_cli_defaults.json"), "w") as fd:
        json.dump(out, fd, indent=1, sort_keys=True)


if __name__ == "__main__":
    gen_init()
    gen_windows()
    gen_quantizer()
    gen_checkpoint()
    gen_cli_defaults()
    for n in ("ref_init.npz", "ref_windows.npz", "ref_quantizer.npz", "ref_checkpoint.pkl", "ref_checkpoint_inputs.npz",
              "ref_cli_defaults.json"):
        print(n, os.path.getsize(os.path.join(OUT, n)), "bytes")
