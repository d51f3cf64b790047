#!/bin/bash
# tuning lib: correctness of the conv tests + step timing under several env settings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 ms/step %.3f loss %.4f %.4f' % (d['ms_per_step'], d['loss_source'], d['loss_target']))"; }
for e in "$@"; do
  echo "=== $e"
  env $e timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -x -q -k "big_tile or conv_fwd_dgrad" 2>&1 | tail -2
done
for rep in 1 2 3; do
  for e in "X=1" "$@"; do env $e bash -c "$(declare -f run); run '$e'"; done
done
