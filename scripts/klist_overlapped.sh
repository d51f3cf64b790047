#!/bin/bash
# per-launch list of one OVERLAPPED step (all streams), eager launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ko
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/ko -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/ko.log 2>&1
python scripts/klist.py gpurun_out/ko/r_results.db > gpurun_out/klist_overlapped.txt
rm -rf gpurun_out/ko
wc -l gpurun_out/klist_overlapped.txt
