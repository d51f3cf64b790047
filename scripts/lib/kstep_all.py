"""Kernel time of the LAST bench step of a rocprofv3 kernel-trace database, aggregated by kernel name (model
initialisation and warm-up excluded).  usage: python scripts/kstep_all.py <results.db> [top]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute('select name, start, end from kernels order by start').fetchall()
marks = [r[2] for r in rows if 'sgd_step_kernel' in r[0]]
lo, hi = marks[-2], marks[-1]
agg = {}
for n, s, e in rows:
    if not (lo < e <= hi): continue
    n = re.sub(r'\(.*', '', n.replace('void ', ''))[:58]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(t for _, t in agg.values())
print('one step: %d launches, %.3f ms of kernel time' % (sum(c for c, _ in agg.values()), tot / 1e3))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%-58s %5d  %8.1f us  %6.1f avg  %5.1f %%' % (n, c, t, t / c, 100 * t / tot))
