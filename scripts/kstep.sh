#!/bin/bash
# kernel time of one serial step by kernel name (+ per-launch list) for the current build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-h2d --eager > gpurun_out/ks.log 2>&1
python scripts/kstep_all.py gpurun_out/ks/r_results.db 60 > gpurun_out/ks_all.txt
python scripts/klist.py gpurun_out/ks/r_results.db > gpurun_out/klist.txt
rm -rf gpurun_out/ks
head -${1:-24} gpurun_out/ks_all.txt
