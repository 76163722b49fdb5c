"""VGPR / scratch / LDS table of the fit and forward kernels in `hipcc -S` dumps.  usage: vgpr_table.py a.s [b.s] (b: compare)"""
import re
import sys


def table(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", s):
        name = m.group(1)
        k = re.search(r"(fit_kernel|forward_kernel)I(.*?)EEv", name)
        if not k:
            continue
        args = re.findall(r"L[ib](\d+)E", k.group(2))
        out[k.group(1) + "<" + ",".join(args) + ">"] = (int(m.group(3)), int(m.group(4)), int(m.group(2)))
    return out


a = table(sys.argv[1])
b = table(sys.argv[2]) if len(sys.argv) > 2 else None
for k in sorted(a):
    if b is None:
        print(f"{k:48s} vgpr {a[k][0]:4d} spill {a[k][1]:3d} sgpr {a[k][2]:4d}")
    elif k in b:
        flag = "  <-- waves/SIMD change" if (512 // ((a[k][0] + 7) // 8 * 8)) != (512 // ((b[k][0] + 7) // 8 * 8)) else ""
        print(f"{k:48s} vgpr {a[k][0]:4d} -> {b[k][0]:4d}  spill {a[k][1]:3d} -> {b[k][1]:3d}{flag}")
