"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite) as text: per-kernel calls, total ms,
average us, share.  usage: python scripts/prof_summary.py <results.db> [steps]  (steps: only used when the
run has no sgd_step_kernel to count them by)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '')
    if 'at::native' in name:
        m = re.search(r'(\w+Functor|\w+_kernel\w*|direct_copy\w*)', name)
        name = 'torch:' + (m.group(1) if m else name[:40])
    return name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = db.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += calls
        a[1] += tot
    # the run executes more steps than it times (warm-up, plan recording, the step behind it): count them by the kernel
    # that runs exactly once per step
    if 'sgd_step_kernel' in agg:
        steps = float(agg['sgd_step_kernel'][0])
    total = sum(v[1] for v in agg.values())
    print('total kernel time %.3f ms over %.0f steps = %.3f ms/step' % (total / 1e3, steps, total / 1e3 / steps))
    print('%-60s %8s %10s %9s %6s' % ('kernel', 'calls/st', 'ms/step', 'avg us', '%'))
    for k, (calls, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%-60s %8.1f %10.3f %9.2f %6.2f' % (k, calls / steps, tot / 1e3 / steps, tot / calls, 100 * tot / total))


if __name__ == '__main__':
    main()
