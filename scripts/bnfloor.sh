#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
for e in "X=1" "RGDA_BN_ROWS=4" "RGDA_BN_ROWS=2" "RGDA_BN_VPB=32 RGDA_BN_ROWS=4" "RGDA_BN_VPB=32 RGDA_BN_ROWS=2"; do echo "--- $e"; env $e python scripts/dev/dev_copy_floor.py 2>&1 | grep "M=" | sed 's/.*| bn_train/bn_train/'; done
