#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_prof.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --serial "$@" > gpurun_out/prof_$tag.log 2>&1
python scripts/prof_summary.py gpurun_out/prof_$tag/r_results.db 5 > gpurun_out/prof_$tag.txt
grep '"metric"' gpurun_out/prof_$tag.log >> gpurun_out/prof_$tag.txt
rm -rf gpurun_out/prof_$tag
cat gpurun_out/prof_$tag.txt
