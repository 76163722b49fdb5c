"""Per-basic-block instruction census of one kernel in a `hipcc -S` dump (VALU / SALU / LDS / VMEM / waits / branches),
with the loop-depth comments LLVM leaves at block heads.  usage: asm_blocks.py file.s <kernel-symbol-substring>"""
import collections
import re
import sys


def main(path, pat):
    s = open(path).read().split('\n')
    start = next(i for i, l in enumerate(s) if pat in l and re.match(r"^[A-Za-z_][\w$.]*:", l))
    end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
    blocks = []
    cur = ['entry', '', collections.Counter()]
    for l in s[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r'^(\.LBB[0-9_]+):\s*(;.*)?$', t)
        if m:
            blocks.append(cur)
            cur = [m.group(1), (m.group(2) or ''), collections.Counter()]
            continue
        if t.startswith(';') and ('Loop' in t or 'Depth' in t):
            cur[1] += ' ' + t
            continue
        if not t or t.startswith(('.', ';')):
            continue
        op = t.split()[0]
        if op.startswith('v_'):
            kind = 'valu'
            if op.startswith(('v_exp', 'v_rcp', 'v_sqrt', 'v_rsq', 'v_log')):
                cur[2]['trans'] += 1
            if op.startswith('v_mfma'):
                kind = 'mfma'
        elif op.startswith('ds_'):
            kind = 'lds'
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            kind = 'vmem'
        elif op.startswith('s_waitcnt'):
            kind = 'wait'
        elif op.startswith(('s_cbranch', 's_branch')):
            kind = 'br'
        elif op.startswith('s_'):
            kind = 'salu'
        else:
            kind = 'other'
        cur[2][kind] += 1
    blocks.append(cur)
    tot = collections.Counter()
    for name, cm, c in blocks:
        n = sum(v for k, v in c.items() if k != 'trans')
        tot.update(c)
        d = re.findall(r'Depth=(\d)', cm)
        print(f"{name:12s} n={n:5d} valu={c['valu']:5d} trans={c['trans']:3d} salu={c['salu']:4d} lds={c['lds']:4d} "
              f"vmem={c['vmem']:3d} wait={c['wait']:3d} br={c['br']:2d}  {'depth ' + d[-1] if d else ''} {cm[:60]}")
    print('total', dict(tot))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
