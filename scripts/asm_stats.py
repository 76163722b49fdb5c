"""Instruction mix per loop region of a kernel in a `hipcc -S` dump, weighted by the issue
costs measured with scripts/ubench/valu_rate*.hip on MI355X (nominal cycles per
wave-instruction per SIMD)."""
import collections
import re
import sys

COST = {'v_exp_f32': 8.3, 'v_rcp_f32': 8.3, 'v_sqrt_f32': 8.3, 'v_rsq_f32': 8.3,
        'v_cndmask_b32': 4.5, 'v_max_f32': 4.5, 'v_min_f32': 4.5, 'v_med3_f32': 4.5, 'v_floor_f32': 4.5,
        'v_min3_f32': 4.5, 'v_max3_f32': 4.5,
        'v_fma_f32': 3.2, 'v_pk_fma_f32': 5.0, 'v_pk_mul_f32': 5.0, 'v_pk_add_f32': 5.0}


def cost(op):
    base = op.replace('_e32', '').replace('_e64', '').replace('_dpp', '')
    if base.startswith('v_cmp'):
        return 4.5
    if base in COST:
        return COST[base]
    if base.startswith('v_'):
        return 2.6
    return 1.0


def main(path, kernel):
    s = open(path).read()
    i = s.index(kernel + ':')
    body = s[i:s.index('s_endpgm', i)]
    lines = body.split('\n')

    def stats(a, b, name):
        seg = [l.strip() for l in lines[a:b]]
        ins = [l.split()[0] for l in seg if l and not l.startswith(('.', ';')) and not l.endswith(':')]
        c = collections.Counter(ins)
        tot = sum(cost(k) * v for k, v in c.items())
        print(f"{name}: {len(ins)} instrs, est {tot:.0f} cycles; top:", c.most_common(16))

    d2 = [n for n, l in enumerate(lines) if 'Depth=2' in l]
    hdr = [n for n, l in enumerate(lines) if 'This Loop Header: Depth=1' in l]
    if not d2 or not hdr:
        stats(0, len(lines), 'whole kernel')
        return
    lo = min(d2)
    nxt = [n for n, l in enumerate(lines) if n > max(d2) and re.match(r'^\.LBB', l) and 'Depth=2' not in l]
    hi = nxt[0] if nxt else max(d2) + 1
    stats(hdr[-1], lo, 'iteration-pre')
    stats(lo, hi, 'pixel-loop body')
    stats(hi, len(lines), 'iteration-post + epilogue')
    stats(0, hdr[-1], 'prologue')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
