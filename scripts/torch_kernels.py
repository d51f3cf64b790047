"""List the PyTorch-native kernels (at::native...) of a rocprofv3 kernel-trace database with counts per step."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
allr = db.execute('select name, grid_x, workgroup_x, (end - start), start, end from kernels order by start').fetchall()
marks = [r[5] for r in allr if 'sgd_step_kernel' in r[0]]
steps = min(int(steps), len(marks) - 1)
rows = [r[:4] for r in allr if r[4] >= marks[-1 - steps] and r[5] <= marks[-1]]
agg = {}
for n, gx, wx, d in rows:
    if 'at::native' not in n and 'rocclr' not in n: continue
    m = re.search(r'(\w+Functor\w*|CUDAFunctor\w*|\w+_kernel_cuda\w*|copyBuffer|fillBuffer\w*|direct_copy\w*|distribution\w*|CatArray\w*)', n)
    fn = re.search(r'at::native::(\w+)<', n)
    key = ((m.group(1) if m else '?') + ' / ' + (fn.group(1) if fn else n[:30]), gx)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for (k, gx), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%-70s grid %-9d %6.1f/step %8.1f us/step' % (k[:70], gx, c / steps, t / 1e3 / steps))
