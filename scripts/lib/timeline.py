"""Timeline statistics of an overlapped step from a rocprofv3 --kernel-trace rocpd database:
per-queue busy time, union busy time, idle time, and the time two queues run concurrently.
usage: python scripts/timeline.py <results.db> <steps> [skip_fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
print('columns:', cols)
rows = db.execute('select name, start, end, %s from kernels order by start' % (qcol or '0')).fetchall()
# window: the last `steps` steps, delimited by the optimizer kernel that ends each step
marks = [e for n, s, e, q in rows if 'sgd_step_kernel' in n]
steps = int(min(steps, len(marks) - 1))
t_lo, t_hi = marks[-1 - steps], marks[-1]
rows = [r for r in rows if r[1] >= t_lo and r[2] <= t_hi]
span = t_hi - t_lo
print('%d steps, %.2f ms/step' % (steps, span / 1e6 / steps))
per_q = {}
for n, s, e, q in rows:
    per_q.setdefault(q, []).append((s, e))
ev = []
for n, s, e, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = conc = 0
depth = 0
last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: conc += t - last
    depth += d
    last = t
print('span %.2f ms  union-busy %.2f ms (%.1f%%)  idle %.2f ms  >=2 kernels in flight %.2f ms' % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, conc / 1e6))
for q, iv in per_q.items():
    b = sum(e - s for s, e in iv)
    print('queue %s: %d kernels, busy %.2f ms (%.1f%% of span)' % (q, len(iv), b / 1e6, 100.0 * b / span))
# gaps on the busiest queue
q0 = max(per_q, key=lambda q: len(per_q[q]))
iv = sorted(per_q[q0])
gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
gaps = [g for g in gaps if g > 0]
import statistics
print('busiest queue: %d gaps, median %.1f us, mean %.1f us, total %.2f ms' % (len(gaps), statistics.median(gaps) / 1e3, statistics.mean(gaps) / 1e3, sum(gaps) / 1e6))
big = sorted(((iv[i + 1][0] - iv[i][1], i) for i in range(len(iv) - 1)), reverse=True)[:25]
names = {(s, e): n for n, s, e, q in rows if q == q0}
for g, i in big:
    print('  gap %.1f us after %s' % (g / 1e3, names[iv[i]][:70]))
# context of the largest gap: what both queues do around it
g, i = big[0]
t0, t1 = iv[i][1], iv[i + 1][0]
print('--- around the largest gap on queue %s (%.1f us): kernels of all queues from -200 us to +200 us' % (q0, g / 1e3))
for n, s, e, q in rows:
    if e >= t0 - 200e3 and s <= t1 + 200e3:
        print('  q%s  start %+9.1f us  dur %7.1f us  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, n[:60]))
