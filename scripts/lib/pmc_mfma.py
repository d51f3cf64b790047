"""Aggregates the two rocprofv3 --pmc passes of `scripts/tune.sh profiles` (gpurun_out/pmc_m1, pmc_m2) by kernel name:
matrix-pipe utilisation, wave-state shares and LDS bank conflicts per kernel."""
import collections
import csv
import glob

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for i in (1, 2):
    f = glob.glob('gpurun_out/pmc_m%d/**/*counter_collection.csv' % i, recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
        keys = [k]
        if k.startswith('conv3x3_halo_kernel') and r.get('Grid_Size') and r.get('Workgroup_Size'):
            # the 3x3 halo kernels also per launch size: the student's launches fill the chip (256 workgroups), the
            # teacher's 8-image launches of the same kernel are 128 workgroups -- half of the SIMDs have no wave at all
            keys.append('  %s @ %d workgroups' % (k[len('conv3x3_halo_kernel'):], int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)))
        for k in keys:
            agg[k][r['Counter_Name'] + ('' if r['Counter_Name'] != 'GRBM_GUI_ACTIVE' else str(i))] += float(r['Counter_Value'])
            if i == 1 and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE1', 0))
print('# MFMA util = SQ_VALU_MFMA_BUSY_CYCLES (= 32 cycles x MFMA instructions, summed over the 1024 SIMDs) / (1024 x kernel cycles), kernel cycles =')
print('# GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); wave-state shares of SQ_WAVE_CYCLES (quad-cycles);')
print('# LDS: bank-conflict cycles / LDS-active cycles.  Counters summed over all launches of a kernel in the run (2 + 1 steps).')
print('%-54s %6s %9s %7s %7s %7s %7s %7s %8s' % ('kernel', 'n', 'cycles/n', 'mfma%', 'wait%', 'istall%', 'active%', 'ldsw%', 'ldsconf%'))
for k, v in rows[:36]:
    n = max(cnt[k], 1)
    ga = v.get('GRBM_GUI_ACTIVE1', 0.0)
    wc = max(v.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    print('%-54s %6d %9.0f %7.1f %7.1f %7.1f %7.1f %7.1f %8.1f' % (
        k, n, ga / 8.0 / n, 100.0 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(ga / 8.0 * 1024, 1),
        100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        100 * v.get('SQ_WAIT_INST_LDS', 0) / wc, 100.0 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 0), 1)))
