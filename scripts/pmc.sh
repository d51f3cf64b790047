#!/bin/bash
# usage: bash scripts/pmc.sh "<counters>" <cmd...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ctr=$1; shift
rm -rf gpurun_out/pmc
timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc -o p -- "$@" > gpurun_out/pmc.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter csv', glob.glob('gpurun_out/pmc/**/*', recursive=True)[:10]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'conv' not in k: continue
    print(k, {c: '%.3g' % (v / max(1, cnt[(k, c)])) for c, v in d.items()})
PY
rm -rf gpurun_out/pmc
