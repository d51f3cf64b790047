"""Instruction pattern of a kernel's loops: python scripts/lib/kpat.py <file.s> <mangled-name substring> ...
M = MFMA, r = LDS read, w = LDS write, D = LDS-DMA, g = global load, s = global store, [..] = s_waitcnt, |B| = barrier,
. = other VALU, , = SALU; one line per basic block that contains an MFMA (loop headers marked)."""
import re
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end\d+:', txt, flags=re.S):
    name, body = m.group(1), m.group(2)
    if not all(p in name for p in sys.argv[2:]):
        continue
    print(name[:100])
    blocks, cur, label = [], [], 'entry'
    for l in body.split('\n'):
        if l.startswith('.LBB'):
            blocks.append((label, cur))
            cur, label = [], l.strip()
            continue
        t = l.strip().split()
        if not t:
            continue
        op = t[0]
        if op.startswith('v_mfma'): cur.append('M')
        elif op.startswith('ds_read'): cur.append('r')
        elif op.startswith('ds_write'): cur.append('w')
        elif op == 's_waitcnt': cur.append('[' + l.strip().split(' ', 1)[1].replace('lgkmcnt', 'l').replace('vmcnt', 'v').replace(' ', '') + ']')
        elif op == 's_barrier': cur.append('|B|')
        elif op.startswith('buffer_load') and 'lds' in l: cur.append('D')
        elif op.startswith(('buffer_load', 'global_load')): cur.append('g')
        elif op.startswith(('buffer_store', 'global_store')): cur.append('s')
        elif op.startswith('scratch_'): cur.append('!')
        elif op.startswith('v_'): cur.append('.')
        elif op.startswith('s_'): cur.append(',')
    blocks.append((label, cur))
    for label, cur in blocks:
        if 'M' in cur:
            print('  ', label[:60])
            print('     ', ''.join(cur))
