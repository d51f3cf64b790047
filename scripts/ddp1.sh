#!/bin/bash
# forced RCCL path on one GPU (world = 1): overhead of issuing the bucketed all-reduces next to backward
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 ms/step %.3f host %.2f plan %s' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('plan_replay')))"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d"
for rep in 1 2; do
  $B 2>/dev/null | show plain
  RGDA_FORCE_DDP=1 $B 2>gpurun_out/ddp_err.txt | show ddp
  RGDA_FORCE_DDP=1 $B --no-comm-overlap 2>/dev/null | show ddp-after-backward
  RGDA_FORCE_DDP=1 NCCL_MAX_NCHANNELS=4 $B 2>/dev/null | show ddp-4ch
  RGDA_FORCE_DDP=1 $B --eager 2>/dev/null | show ddp-eager
done
tail -3 gpurun_out/ddp_err.txt
