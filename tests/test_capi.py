"""CPU checks of the drop-in boundary: libsmoe_hip.so loads, exports every entry point that
include/smoe_hip.h declares, the ctypes structures have the C layout, and (without a GPU)
the library refuses to run instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smoe_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smoe_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from steered_mixture_of_experts_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in smoe_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert lib.smoe_abi_version() == _lib.SMOE_ABI_VERSION


def test_ctypes_structs_match_the_c_layout(tmp_path):
    from steered_mixture_of_experts_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smoe_hip.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(smoe_config), sizeof(smoe_params),'
                    ' sizeof(smoe_adam_state), offsetof(smoe_config, start_pis), offsetof(smoe_config, lr_expert),'
                    ' offsetof(smoe_adam_state, beta1_power), offsetof(smoe_adam_state, step));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.SmoeConfig), C.sizeof(_lib.SmoeParams), C.sizeof(_lib.SmoeAdamState),
            _lib.SmoeConfig.start_pis.offset, _lib.SmoeConfig.lr_expert.offset,
            _lib.SmoeAdamState.beta1_power.offset, _lib.SmoeAdamState.step.offset]
    assert got == want


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from steered_mixture_of_experts_amd import _lib
    lib = _lib.load()
    c = _lib.SmoeConfig()
    c.abi_version, c.dim, c.channels, c.kernels, c.precision = _lib.SMOE_ABI_VERSION, 2, 1, 4, 8
    c.block_shape[0] = c.block_shape[1] = 16
    c.block_shape[2] = 1
    h = C.c_void_p()
    rc = lib.smoe_create(C.byref(h), C.byref(c))
    assert rc == _lib.SMOE_ERR_NO_DEVICE and b"no CPU path" in lib.smoe_last_error()
    from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
    with pytest.raises(RuntimeError):
        BlockEngine(EngineConfig(block_shape=(16, 16), channels=1, kernels=4))


def test_argument_validation_messages():
    from steered_mixture_of_experts_amd import _lib
    lib = _lib.load()
    assert lib.smoe_is_supported(2, 1, 4) == 1 and lib.smoe_is_supported(2, 2, 5) == 0
    h = C.c_void_p()
    c = _lib.SmoeConfig()
    c.abi_version = 99
    assert lib.smoe_create(C.byref(h), C.byref(c)) == _lib.SMOE_ERR_INVALID
    assert b"abi_version" in lib.smoe_last_error()
    c.abi_version, c.dim = _lib.SMOE_ABI_VERSION, 5
    assert lib.smoe_create(C.byref(h), C.byref(c)) == _lib.SMOE_ERR_INVALID
    assert lib.smoe_forward(None, 1, None, None, None, None, None, None, None, None, None, 0, None) == _lib.SMOE_ERR_INVALID


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "steered_mixture_of_experts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("smoe_oracle", "oracle") or "oracle" not in re.sub(r'""".*?"""|#.*|//.*', "", txt, flags=re.S), f
    out = subprocess.check_output([sys.executable, "-c",
                                   "import sys; import steered_mixture_of_experts_amd.smoe, steered_mixture_of_experts_amd.engine;"
                                   "print(any(m.startswith('oracle') for m in sys.modules))"], cwd=ROOT)
    assert out.strip() == b"False"
