#!/bin/bash
# A/B on ONE box, interleaved.  LIBA / LIBB: library files (default: csrc/base/librgda_hip.so vs the current build);
# ENVA / ENVB: environment for each arm (tuning builds read RGDA_* hooks); TESTS: pytest files run first with LIBB.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp ${LIBB:-regda_amd/csrc/librgda_hip.so} /tmp/lib_B.so; cp ${LIBA:-regda_amd/csrc/base/librgda_hip.so} /tmp/lib_A.so
cp /tmp/lib_B.so regda_amd/csrc/librgda_hip.so
if [ -n "$TESTS" ]; then env $ENVB timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -4; fi
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 ms/step %.3f host %.2f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for rep in 1 2 3; do
  cp /tmp/lib_A.so regda_amd/csrc/librgda_hip.so; env $ENVA bash -c "$(declare -f run); run A"
  cp /tmp/lib_B.so regda_amd/csrc/librgda_hip.so; env $ENVB bash -c "$(declare -f run); run B"
done
