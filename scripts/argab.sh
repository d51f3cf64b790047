#!/bin/bash
# A/B of bench.py argument sets on ONE box, interleaved: scripts/argab.sh "<args A>" "<args B>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  for a in "$@"; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d $a 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('[%s] ms/step %.3f' % ('$a', d['ms_per_step']))"
  done
done
