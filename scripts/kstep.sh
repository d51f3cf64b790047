#!/bin/bash
# usage (via gpurun): bash scripts/kstep.sh <kernel substring> [<kernel substring> ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial > gpurun_out/ks.log 2>&1
for pat in "$@"; do python scripts/kstep.py gpurun_out/ks/r_results.db "$pat"; done | tee gpurun_out/ks.txt
rm -rf gpurun_out/ks
