#!/bin/bash
# usage (via gpurun): bash scripts/timeline.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl_$tag
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl_$tag -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline "$@" > gpurun_out/tl_$tag.log 2>&1
python scripts/timeline.py gpurun_out/tl_$tag/r_results.db 3 | tee gpurun_out/tl_$tag.txt
grep -o '"ms_per_step[^,]*' gpurun_out/tl_$tag.log
rm -rf gpurun_out/tl_$tag
