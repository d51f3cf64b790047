#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
  echo base; $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  echo T82=256; RGDA_T82=256 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"
  echo NO_PIPE; RGDA_NO_PIPE=1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
