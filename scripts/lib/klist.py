"""Every kernel launch of the LAST bench step of a rocprofv3 kernel-trace database, in start order:
start offset (us), gap to the previous kernel's end, duration, grid / workgroup size, LDS, kernel name.
usage: python scripts/klist.py <results.db>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
want = [c for c in ('grid_x', 'grid_size_x', 'workgroup_x', 'workgroup_size_x', 'lds_size', 'lds_block_size', 'queue_id', 'stream_id') if c in cols]
rows = db.execute('select name, start, end%s from kernels order by start' % ''.join(', ' + c for c in want)).fetchall()
marks = [r[2] for r in rows if 'sgd_step_kernel' in r[0]]
lo, hi = marks[-2], marks[-1]
print('# columns:', cols)
print('# start_us gap_us dur_us ' + ' '.join(want) + ' name')
prev = None
for r in rows:
    n, s, e = r[:3]
    if not (lo < e <= hi):
        continue
    n = re.sub(r'\(.*', '', n.replace('void ', ''))[:70]
    gap = 0.0 if prev is None else (s - prev) / 1e3
    print('%9.1f %6.1f %7.1f %s %s' % ((s - lo) / 1e3, gap, (e - s) / 1e3, ' '.join(str(x) for x in r[3:]), n))
    prev = e
