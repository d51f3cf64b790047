#!/bin/bash
# tuning build of the library next to the product one: regda_amd/csrc/tuning/librgda_hip.so (RGDA_* env hooks live);
# copy it over librgda_hip.so on the GPU box (scratch copy) for an experiment
set -e
cd "$(dirname "$0")/../regda_amd/csrc"
mkdir -p tuning
python3 gen_thunks.py ../../include/rgda_hip.h plan_thunks.inc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -DRGDA_TUNING -c $f -o tuning/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tuning/librgda_hip.so tuning/*.o
echo built tuning/librgda_hip.so
