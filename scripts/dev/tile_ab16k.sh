cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
for shape in "16 32 32 256 256 3 1 1" "16 32 32 1024 256 1 0 1"; do
  for e in X=1 RGDA_TILE=128,64,3 RGDA_TILE=64,128,3 RGDA_TILE=64,64,3 RGDA_TILE=128,128,82 RGDA_TILE=128,64,4; do
    echo -n "$e: "; env $e python scripts/dev/dev_one_conv.py $shape 40 2>&1 | grep conv
  done
done
