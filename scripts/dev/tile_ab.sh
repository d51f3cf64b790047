cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
for shape in "16 32 32 256 1024 1 0 1" "16 32 32 512 2048 1 0 1" "16 64 64 128 512 1 0 1" "16 128 128 64 256 1 0 1" "16 32 32 2048 512 1 0 1" "16 32 32 1024 2048 1 0 1"; do
  for e in X=1 RGDA_T82=100000 RGDA_TILE=128,256,83 RGDA_TILE=128,64,3; do
    echo -n "$e: "; env $e python scripts/dev/dev_one_conv.py $shape 40 2>&1 | grep conv
  done
done
