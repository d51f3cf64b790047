#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp regda_amd/csrc/librgda_hip.so /tmp/lib_B.so
echo "--- base"; cp regda_amd/csrc/base/librgda_hip.so regda_amd/csrc/librgda_hip.so; python scripts/dev/dev_copy_floor.py 2>&1 | grep "M=" | sed 's/.*| bn_train/bn_train/' 
echo "--- new"; cp /tmp/lib_B.so regda_amd/csrc/librgda_hip.so; python scripts/dev/dev_copy_floor.py 2>&1 | grep "M=" | sed 's/.*| bn_train/bn_train/'
cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
for e in "RGDA_BN_ROWS=4" "RGDA_BN_ROWS=16" "RGDA_BN_VPB=8" "RGDA_BN_VPB=32 RGDA_BN_ROWS=4"; do echo "--- new tuning $e"; env $e python scripts/dev/dev_copy_floor.py 2>&1 | grep "M=" | sed 's/.*| bn_train/bn_train/'; done
