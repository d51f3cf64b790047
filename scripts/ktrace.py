"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace rocpd database, in launch order.
usage: python scripts/ktrace.py <results.db> <name substring> [group size]
Prints the median duration (us) of every consecutive group of `group size` matching dispatches."""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
grp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
try:
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
except sqlite3.Error as e:
    print('schema?', e)
    print([r[0] for r in db.execute("select name from sqlite_master").fetchall()])
    sys.exit(1)
d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
print(len(d), 'dispatches of', pat)
for i in range(0, len(d), grp):
    g = d[i:i + grp]
    print('%4d: median %.1f us  min %.1f  max %.1f' % (i // grp, statistics.median(g), min(g), max(g)))
