#!/bin/bash
# quick check on the GPU box: selected tests + a few short bench lines (ms/step, host enqueue)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -5; fi
for rep in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ms/step %.3f host %.2f loss %.4f %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['loss_source'], d['loss_target']))"
done
