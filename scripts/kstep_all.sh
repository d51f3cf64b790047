#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial > gpurun_out/ks.log 2>&1
python scripts/kstep_all.py gpurun_out/ks/r_results.db 45 | tee gpurun_out/ks_all.txt
rm -rf gpurun_out/ks
