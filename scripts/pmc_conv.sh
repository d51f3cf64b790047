#!/bin/bash
# memory-path counters of ONE conv geometry (args as scripts/dev/dev_one_conv.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python scripts/dev/dev_one_conv.py "$@" 2>&1 | grep conv
i=0
# (TA_* / TCP_* counter passes hung rocprofv3 on this pool: not collected; every pass runs under `timeout`)
for pass in "TCC_HIT TCC_MISS TCC_REQ TCC_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ TCC_TAG_STALL TCC_READ TCC_CYCLE GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pc$i
  timeout 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pc$i -o p -- python scripts/dev/dev_one_conv.py "$@" > gpurun_out/pc$i.log 2>&1
  python - $i <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
f = glob.glob('gpurun_out/pc%s/**/*counter_collection.csv' % i, recursive=True)
if not f:
    print('pass', i, 'no output'); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'conv_igemm' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print('pass %s:' % i, '  '.join('%s=%.4g' % (k, v / max(n[k], 1)) for k, v in agg.items()))
PY
  rm -rf gpurun_out/pc$i
done
