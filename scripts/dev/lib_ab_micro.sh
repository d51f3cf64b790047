#!/bin/bash
# A/B of one dev micro-benchmark between csrc/base/librgda_hip.so (A) and the current build (B) on one box.  $1 = script
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/librgda_hip.so /tmp/lib_B.so; cp regda_amd/csrc/base/librgda_hip.so /tmp/lib_A.so
for arm in A B A B; do
  cp /tmp/lib_$arm.so regda_amd/csrc/librgda_hip.so; echo "== $arm"; timeout 300 python $1 2>&1 | grep -v amdgpu.ids | tail -${2:-12}
done
cp /tmp/lib_B.so regda_amd/csrc/librgda_hip.so
