"""Per-basic-block register high-water marks of one kernel in a `hipcc -S` dump: highest VGPR / AGPR index referenced and
the number of v_accvgpr moves (AGPRs as spill space).  usage: asm_regs.py file.s <kernel-symbol-substring> [min_instr]"""
import re
import sys


def main(path, pat, min_n=20):
    s = open(path).read().split('\n')
    start = next(i for i, l in enumerate(s) if pat in l and re.match(r"^[A-Za-z_][\w$.]*:", l))
    end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
    cur = ['entry', 0, -1, -1, 0, '']
    blocks = []
    for l in s[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r'^(\.LBB[0-9_]+):\s*(;.*)?$', t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, -1, -1, 0, m.group(2) or '']
            continue
        if t.startswith(';') and 'Depth' in t:
            cur[5] += ' ' + t
            continue
        if not t or t.startswith(('.', ';')):
            continue
        cur[1] += 1
        for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', t):
            cur[2] = max(cur[2], int(b))
        for a in re.findall(r'\bv(\d+)\b', t):
            cur[2] = max(cur[2], int(a))
        for a, b in re.findall(r'\ba\[(\d+):(\d+)\]', t):
            cur[3] = max(cur[3], int(b))
        for a in re.findall(r'\ba(\d+)\b', t):
            cur[3] = max(cur[3], int(a))
        if t.startswith('v_accvgpr'):
            cur[4] += 1
    blocks.append(cur)
    for b in blocks:
        if b[1] >= min_n:
            d = re.search(r'Depth=(\d)', b[5])
            print(f"{b[0]:12s} n={b[1]:5d} max_v={b[2]:4d} max_a={b[3]:4d} accvgpr_moves={b[4]:4d} depth={d.group(1) if d else '-'}")


main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
