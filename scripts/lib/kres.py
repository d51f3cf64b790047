"""Per-kernel resources from the compiler's metadata: python scripts/lib/kres.py <file.s> [substring ...]
(LDS bytes, VGPRs, AGPRs, spills; names demangled with c++filt)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
meta = txt[txt.index('amdhsa.kernels:'):]
pats = sys.argv[2:]
rows = []
for blk in meta.split('  - .agpr_count:')[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    rows.append((g('name'), g('group_segment_fixed_size'), g('vgpr_count'), blk.split('\n')[0].strip(), g('vgpr_spill_count')))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for (n, lds, vg, ag, sp), dn in zip(rows, names):
    dn = re.sub(r'\(.*', '', dn)
    if not pats or any(p in dn for p in pats):
        print(f'{dn[:64]:64s} lds {lds:>7s} vgpr {vg:>4s} agpr {ag:>4s} spill {sp}')
