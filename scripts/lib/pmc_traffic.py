"""Aggregates the two rocprofv3 --pmc passes FETCH_SIZE / WRITE_SIZE (gpurun_out/pmc_<ctr>/**/counter_collection.csv) by
kernel name: KiB per launch (text on stdout) and bytes per launch of the conv kernels (gpurun_out/pmc_traffic.json, read
by bench.py for roofline.traffic).  gfx950's FETCH_SIZE tallies 128-byte requests at 64 B: doubled in the JSON, as
MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes.  Called by `scripts/tune.sh profiles`."""
import collections
import csv
import glob
import json

out = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % ctr, recursive=True)
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != ctr:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:72]
        agg[k] += float(r['Counter_Value'])
        cnt[k] += 1
    out[ctr] = (agg, cnt)
keys = sorted(out['FETCH_SIZE'][0], key=lambda k: -out['FETCH_SIZE'][0][k])
print('%-58s %8s %14s %14s' % ('kernel', 'launches', 'FETCH_SIZE/launch', 'WRITE_SIZE/launch'))
traffic = {}
for k in keys[:30]:
    n = out['FETCH_SIZE'][1][k]
    f = out['FETCH_SIZE'][0][k] / n
    w = out['WRITE_SIZE'][0].get(k, 0) / max(1, out['WRITE_SIZE'][1].get(k, 1))
    print('%-58s %8d %14.1f %14.1f' % (k, n, f, w))
    if k.startswith('conv_'):       # bytes per launch: the counters are in KiB
        traffic[k] = (2.0 * f + w) * 1024.0
json.dump(traffic, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
