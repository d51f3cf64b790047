#!/bin/bash
# usage (via gpurun): [ARMS="dir[:ENV=..[,ENV2=..]] ..."] [TUNING=1] [BARGS="--no-teacher"] [REPS=3] bash scripts/ab.sh
# Interleaved A/B/... of the whole step on ONE box (box-to-box spread on this pool is +-2 %).  An arm is a tree holding
# bench.py (default arms: the tree exported by scripts/ab_snapshot.sh in .ab_base/, and the working tree) plus an
# optional comma-separated environment (the RGDA_* hooks of a tuning build: TUNING=1 copies the working tree's
# csrc/tuning/librgda_hip.so over the product library of the box's scratch copy first).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -n "$TUNING" ] && cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
ARMS=${ARMS:-".ab_base ."}
for rep in $(seq 1 ${REPS:-3}); do
  for arm in $ARMS; do
    dir=${arm%%:*}; envs=""; [ "$arm" != "$dir" ] && envs=$(echo ${arm#*:} | tr ',' ' ')
    ( cd $dir; env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-40s ms/step %.3f host %.2f' % ('$arm', d['ms_per_step'], d['host_enqueue_ms_per_step']))" )
  done
done
