#!/bin/bash
# usage (via gpurun): bash scripts/round_profiles.sh <round tag, e.g. r02>
# Regenerates everything profiles/ holds for a round into gpurun_out/ (copy the files over afterwards):
#   <tag>_bench_1gpu.json              the default bench line (roofline + cpu_baseline) + <tag>_step_phases.txt
#   <tag>_kernel_stats.txt             rocprofv3 --kernel-trace --stats of bench.py --serial (one stream: the
#                                      per-kernel averages the roofline probe's HIP events must agree with)
#   <tag>_kernel_stats_overlapped.txt  the same for the default multi-stream step
#   <tag>_kernel_one_step.txt          kernel time of ONE serial step by kernel name (no model initialisation)
#   <tag>_pmc_hbm_traffic.txt + pmc_traffic.json   FETCH_SIZE / WRITE_SIZE passes
#   <tag>_pmc_mfma.txt                 MFMA busy / wave states / LDS conflicts per kernel (separate --pmc passes)
# every rocprofv3 run sits under `timeout` (a counter pass that hangs must not take the box down)
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out profiles
timeout 600 bash scripts/pmc_traffic.sh > /dev/null 2>&1
cp gpurun_out/pmc_traffic.txt gpurun_out/${tag}_pmc_hbm_traffic.txt
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # the bench line below reads it
timeout 600 bash scripts/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma.txt gpurun_out/${tag}_pmc_mfma.txt
timeout 400 bash scripts/gpu_prof.sh ${tag}s --no-h2d > /dev/null 2>&1; cp gpurun_out/prof_${tag}s.txt gpurun_out/${tag}_kernel_stats.txt
timeout 400 bash scripts/kstep.sh > /dev/null 2>&1; cp gpurun_out/ks_all.txt gpurun_out/${tag}_kernel_one_step.txt
rm -rf gpurun_out/prof_o
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_o -o r -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_o.log 2>&1
python scripts/prof_summary.py gpurun_out/prof_o/r_results.db 5 > gpurun_out/${tag}_kernel_stats_overlapped.txt
grep '"metric"' gpurun_out/prof_o.log >> gpurun_out/${tag}_kernel_stats_overlapped.txt
rm -rf gpurun_out/prof_o
timeout 900 python bench.py --phases > gpurun_out/${tag}_bench_1gpu.json 2> gpurun_out/${tag}_bench_1gpu.err
grep ' ms  ' gpurun_out/${tag}_bench_1gpu.err > gpurun_out/${tag}_step_phases.txt
python bench.py --no-h2d --no-cpu-baseline --no-roofline 2>/dev/null | grep '"metric"' > gpurun_out/${tag}_bench_resident_inputs.json
cat gpurun_out/${tag}_bench_1gpu.json | cut -c1-900
cat gpurun_out/${tag}_step_phases.txt
