#!/usr/bin/env python3
"""tools/asm_mix.py build/plaac_kernels.s <mangled-name-substring> [min-block-size]
Instruction mix per basic block of one kernel of the `make asm` listing (what the hot loops issue)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 100


def cat(op):
    if op.startswith('v_add_f64'):
        return 'add64'
    if op.startswith(('v_mul_f64', 'v_fma_f64', 'v_rcp_f64', 'v_div', 'v_max_f64', 'v_min_f64', 'v_cmp', 'v_floor',
                      'v_cvt', 'v_fmac', 'v_fract')):
        return 'fp_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('v_'):
        return 'valu_other'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and sub in l.split(':')[0] and ':' in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blocks, cur = [], ['entry', []]
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
    else:
        t = l.strip()
        if t and not t.startswith(('.', ';', '//')):
            cur[1].append(t)
blocks.append(cur)
print(lines[start].split(':')[0], end - start, 'lines')
for nm, ins in blocks:
    if len(ins) < minsz:
        continue
    c = collections.Counter(cat(i.split()[0]) for i in ins)
    waits = [i.split(None, 1)[1] for i in ins if i.startswith('s_waitcnt')]
    br = [i for i in ins if i.startswith(('s_cbranch', 's_branch'))]
    print('  %-12s %4d %s' % (nm, len(ins), dict(c)), '| waits:', waits[:5], '|', br[-1:] )
