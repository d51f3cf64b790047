"""Every dispatch of kernels matching a substring inside the last bench step of a rocprofv3 kernel-trace database:
launch order, grid, duration.  usage: python scripts/kstep.py <results.db> <substring>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
rows = db.execute('select name, grid_x, grid_y, grid_z, workgroup_x, start, end from kernels order by start').fetchall()
marks = [r[6] for r in rows if 'sgd_step_kernel' in r[0]]
lo, hi = marks[-2], marks[-1]
tot = 0.0
for i, (n, gx, gy, gz, wx, s, e) in enumerate(r for r in rows if lo <= r[5] <= hi):
    if pat in n:
        tot += (e - s) / 1e3
        print('%5d grid %6d x %3d x %2d wgs  %7.1f us  %s' % (i, gx // wx, gy, gz, (e - s) / 1e3, n[:60]))
print('total %.1f us' % tot)
