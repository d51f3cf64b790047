#!/bin/bash
# the whole GPU suite, not stopping at the first failure; log under gpurun_out/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
grep -E "^\[|passed|failed|FAILED|ERROR|rc " gpurun_out/pytest_gpu.log | tail -40
