"""Instruction mix per basic block of one kernel in a hipcc -S listing (which loop bodies carry what).
    python scripts/asm_blocks.py file.s <mangled-name-substring> [min_instructions]"""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('v_exp') or op.startswith('v_log') or op.startswith('v_rcp') or op.startswith('v_rsq'):
        return 'trans'
    if op.startswith('v_pk_'):
        return 'vpk'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
        return 'sync'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_load_lds') or op.startswith('buffer_load') and 'lds' in op:
        return 'dma'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_') or op.startswith('flat_'):
        return 'vmem'
    return 'other'


def main(path, key, min_n=8):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E') or (l.startswith('_Z') and key in l.split(':')[0]))
    blocks, cur, name = [], {}, 'entry'
    order = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('s_endpgm') and False:
            break
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append((name, cur, order))
            name, cur, order = m.group(1), {}, []
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
        order.append(s)
    blocks.append((name, cur, order))
    for name, cur, order in blocks:
        n = sum(cur.values())
        if n < int(min_n):
            continue
        tgt = [o.split()[-1] for o in order if o.startswith('s_cbranch') or o.startswith('s_branch')]
        print(f'{name:12s} n={n:4d} ' + ' '.join(f'{k}={v}' for k, v in sorted(cur.items())) + '  -> ' + ','.join(tgt))


if __name__ == '__main__':
    main(*sys.argv[1:])
