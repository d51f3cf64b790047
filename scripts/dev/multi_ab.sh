#!/bin/bash
# interleaved A/B/C... of several library builds on ONE box: scripts/dev/multi_ab.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0; for l in "$@"; do cp $l /tmp/lib_$i.so; i=$((i+1)); done
for rep in 1 2 3; do i=0; for l in "$@"; do
  cp /tmp/lib_$i.so regda_amd/csrc/librgda_hip.so
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$l ms/step %.3f' % d['ms_per_step'])"
  i=$((i+1)); done; done
cp /tmp/lib_0.so regda_amd/csrc/librgda_hip.so
