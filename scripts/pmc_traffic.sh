#!/bin/bash
# HBM traffic of the conv kernels over one bench run: two separate PMC passes (FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --serial --eager --no-h2d > gpurun_out/pmc_$ctr.log 2>&1
done
python - > gpurun_out/pmc_traffic.txt <<'PY'
import csv, glob, collections
out = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % ctr, recursive=True)
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != ctr: continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
        agg[k] += float(r['Counter_Value']); cnt[k] += 1
    out[ctr] = (agg, cnt)
keys = sorted(out['FETCH_SIZE'][0], key=lambda k: -out['FETCH_SIZE'][0][k])
print('%-50s %8s %14s %14s' % ('kernel', 'launches', 'FETCH_SIZE/launch', 'WRITE_SIZE/launch'))
import json
traffic = {}
for k in keys[:30]:
    n = out['FETCH_SIZE'][1][k]
    f = out['FETCH_SIZE'][0][k] / n
    w = out['WRITE_SIZE'][0].get(k, 0) / max(1, out['WRITE_SIZE'][1].get(k, 1))
    print('%-50s %8d %14.1f %14.1f' % (k, n, f, w))
    # bytes per launch: counters are in KiB; gfx950's FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    if k.startswith('conv_'):
        traffic[k] = (2.0 * f + w) * 1024.0
json.dump(traffic, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/pmc_traffic.txt
