#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tk
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tk -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --serial > gpurun_out/tk.log 2>&1
python scripts/torch_kernels.py gpurun_out/tk/r_results.db 2 | tee gpurun_out/tk.txt
rm -rf gpurun_out/tk
