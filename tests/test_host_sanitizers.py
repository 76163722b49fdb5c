"""The host layer of libsmoe_hip (csrc/smoe_capi.hip: argument validation, kernel-variant selection, tiling rules incl. the
partition-invariant choice, LDS-size arithmetic, workspace set-up) under AddressSanitizer + UndefinedBehaviorSanitizer on the
CPU box (SURVEY section 5 "sanitizers": GPU sanitizers are not available on the pool).

`make hostcheck` compiles every translation unit host-only with -fsanitize=address,undefined and -DSMOE_HOST_TEST=1 (handles
without a device, workspace on the host heap, HIP calls of the entry points compiled out) and links tests/host/
hostcheck_driver.cpp, which walks ~1.4 million checked calls over every instantiated (dim, channels, kernels) triple, block
shape, graph variant, tiling code and batch size.  Any sanitizer report aborts the driver (-fno-sanitize-recover)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "steered_mixture_of_experts_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_layer_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", CSRC, "hostcheck", "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert build.returncode == 0, build.stdout[-4000:]
    exe = os.path.join(CSRC, "hostcheck", "smoe_hostcheck")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    tail = run.stdout[-4000:]
    assert run.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in run.stdout and "runtime error" not in run.stdout, tail
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith("hostcheck:") and last.endswith(" 0 failed"), tail
    assert int(last.split()[1]) > 1_000_000
