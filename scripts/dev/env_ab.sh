#!/bin/bash
# interleaved A/B/C... of environment settings with the tuning build on ONE box: scripts/dev/env_ab.sh "X=1" "RGDA_FOO=2" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/librgda_hip.so /tmp/lib_product.so; cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
for rep in 1 2 3; do for e in "$@"; do
  env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('[$e] ms/step %.3f' % d['ms_per_step'])"
done; done
cp /tmp/lib_product.so regda_amd/csrc/librgda_hip.so
