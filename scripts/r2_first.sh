#!/bin/bash
# round-2 first GPU call: the whole GPU suite, the bench line, and a per-launch kernel list of one serial step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --phases > gpurun_out/bench0.json 2> gpurun_out/bench0.err; tail -c 600 gpurun_out/bench0.json
rm -rf gpurun_out/ks
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --serial > gpurun_out/ks.log 2>&1
python scripts/kstep_all.py gpurun_out/ks/r_results.db 60 > gpurun_out/ks_all.txt
python scripts/klist.py gpurun_out/ks/r_results.db > gpurun_out/klist.txt
rm -rf gpurun_out/ks
head -30 gpurun_out/ks_all.txt
