#!/bin/bash
# usage (via gpurun): bash scripts/ktrace.sh <tag> <kernel substring> <group> -- <command...>
tag=$1; pat=$2; grp=$3; shift 4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_$tag
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/kt_$tag -o r -- "$@" > gpurun_out/kt_$tag.log 2>&1
python scripts/ktrace.py gpurun_out/kt_$tag/r_results.db "$pat" $grp | tee gpurun_out/kt_$tag.txt
rm -rf gpurun_out/kt_$tag
