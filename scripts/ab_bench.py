"""A/B timing of several builds of libsmoe_hip.so with bench.py in ONE session (alternating, so that clock / thermal
drift hits every build alike).  usage: ab_bench.py [--rounds R] lib1.so lib2.so ... -- <bench.py args>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
import torch
sys.path.insert(0, sys.argv[1])
from steered_mixture_of_experts_amd import _lib
_lib.LIB_PATH = sys.argv[2]
import ctypes
have = ctypes.CDLL(sys.argv[2])
_lib.EXPORTS = tuple(n for n in _lib.EXPORTS if hasattr(have, n))
import bench
bench.main(sys.argv[3:])
'''


def main():
    av = sys.argv[1:]
    rounds = 3
    if av[0] == "--rounds":
        rounds = int(av[1]); av = av[2:]
    k = av.index("--")
    libs, bargs = av[:k], av[k + 1:]
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, "-c", CHILD, ROOT, os.path.abspath(l)] + bargs + ["--no-cpu-baseline", "--no-extras"],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not line:
                res[l].append(float("nan")); print(out.stderr[-600:]); var = "?"; continue
            d = json.loads(line[-1])
            res[l].append(d["value"])
            var = d["config"]["kernel_variant"]
    for l in libs:
        v = res[l]
        print(f"{l:60s} {var:22s} " + " ".join(f"{x:9.1f}" for x in v) + f"   median {sorted(v)[len(v)//2]:9.1f}")


if __name__ == "__main__":
    main()
