#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 ms/step %.3f host %.2f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
  $B --no-h2d 2>/dev/null | show resident
  $B 2>/dev/null | show h2d
  HSA_ENABLE_SDMA=0 $B 2>/dev/null | show h2d-nosdma
done
