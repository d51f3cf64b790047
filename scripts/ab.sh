#!/bin/bash
# A/B on ONE box: regda_amd/csrc/base/librgda_hip.so (A) against the current build (B), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/librgda_hip.so /tmp/lib_B.so; cp regda_amd/csrc/base/librgda_hip.so /tmp/lib_A.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms/step %.3f host %.2f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for rep in 1 2 3; do
  cp /tmp/lib_A.so regda_amd/csrc/librgda_hip.so; run A
  cp /tmp/lib_B.so regda_amd/csrc/librgda_hip.so; run B
done
