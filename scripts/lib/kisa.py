"""Instruction census of one kernel's ISA: python scripts/lib/kisa.py <file.s> <mangled-name substring>
(counted vmcnt waits by value, barriers, MFMAs, LDS / scratch operations) -- the check that a counted-wait K loop did not
acquire a compiler-inserted `s_waitcnt vmcnt(0)` or a spill."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end\d+:', txt, flags=re.S):
    name, body = m.group(1), m.group(2)
    if all(p in name for p in sys.argv[2:]):
        waits = Counter(re.findall(r's_waitcnt[^\n]*vmcnt\((\d+)\)', body))
        print(name[:72])
        print('   lines', body.count('\n'), 'vmcnt', dict(sorted(waits.items(), key=lambda kv: int(kv[0]))), 'barriers', body.count('s_barrier'),
              'mfma', body.count('v_mfma'), 'ds_read_b128', body.count('ds_read_b128'), 'ds_write_b128', body.count('ds_write_b128'),
              'buffer_load_lds', len(re.findall(r'buffer_load_dwordx4[^\n]*lds', body)), 'scratch', body.count('scratch_'))
