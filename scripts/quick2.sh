#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -15; fi
show() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 ms/step %.3f host %.2f loop %.2f plan %s loss %.4f %.4f | %s' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_loop_ms_per_step'], d.get('plan_replay'), d['loss_source'], d['loss_target'], d['data']))"; }
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>gpurun_out/err_plan.txt | show plan+h2d
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | show plan
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager --no-h2d 2>/dev/null | show eager
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager 2>/dev/null | show eager+h2d
done
tail -3 gpurun_out/err_plan.txt
