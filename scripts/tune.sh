#!/bin/bash
# The ONE development / measurement entry point of this repo (everything under scripts/ is reached through it).
#
#   in the build container (no GPU):
#     bash scripts/tune.sh build-tuning        library with the RGDA_* environment hooks -> regda_amd/csrc/tuning/
#     bash scripts/tune.sh snapshot [ref]      export a git ref (default HEAD) to .ab_base/ and build it: arm A of `ab`
#   on a GPU box (prefix with: /usr/local/graft/bin/gpurun --timeout N -- ...):
#     bash scripts/tune.sh tests [pytest args] the whole -m gpu suite, log in gpurun_out/pytest_gpu.log
#     [ARMS=..] [TUNING=1] [BARGS=..] [REPS=3] bash scripts/tune.sh ab
#                                              interleaved A/B/.. of the whole step on ONE box (box-to-box spread is
#                                              +-2 %): an arm is "dir[:ENV=v[,ENV2=v]]" (LIB=path = another build of
#                                              the library for that arm); default arms ".ab_base ."
#     bash scripts/tune.sh kstep [bench args]  kernel time of ONE serial step by kernel name (+ per-launch list)
#     bash scripts/tune.sh klist [bench args]  every launch of one overlapped step in start order (gaps, streams)
#     bash scripts/tune.sh profiles rNN        regenerate everything profiles/ holds for a round into gpurun_out/
#     bash scripts/tune.sh dev <script> [args] a scripts/dev/ probe with the tuning library copied over the product one
#                                              (conv_phases.py, kbench.py, one_conv.py, conv_ablate.sh, mfma_ceiling.py,
#                                               gemm_calib.py, wgrad_calib.py, coresidency.py, grid_barrier.py)
# Every rocprofv3 run sits under `timeout`; counter passes use --kernel-trace only (no other trace domains).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cmd=${1:-help}; shift || true
on_box() { cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$ROOT}"; mkdir -p gpurun_out; }
last_json() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-44s ms/step %.3f host %.2f' % ('$1', d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }

case $cmd in
build-tuning)
  cd "$ROOT/regda_amd/csrc" && mkdir -p tuning && python3 gen_thunks.py ../../include/rgda_hip.h plan_thunks.inc
  for f in *.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -DRGDA_TUNING -c $f -o tuning/${f%.hip}.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tuning/librgda_hip.so tuning/*.o && echo built regda_amd/csrc/tuning/librgda_hip.so ;;
snapshot)
  cd "$ROOT"; ref=${1:-HEAD}; rm -rf .ab_base && mkdir .ab_base
  git archive "$ref" bench.py regda_amd oracle include configs profiles/pmc_traffic.json | tar -x -C .ab_base
  make -C .ab_base/regda_amd/csrc -j8 > /dev/null && echo "snapshot of $ref ($(git rev-parse --short "$ref")) built in .ab_base/" ;;
tests)
  on_box
  timeout 2400 python -m pytest tests -m gpu -q "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log
  grep -E "^\[|passed|failed|FAILED|ERROR|rc " gpurun_out/pytest_gpu.log | tail -40 ;;
ab)
  on_box
  [ -n "${TUNING:-}" ] && cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
  for rep in $(seq 1 ${REPS:-3}); do for arm in ${ARMS:-.ab_base .}; do
    dir=${arm%%:*}; envs=""; [ "$arm" != "$dir" ] && envs=$(echo ${arm#*:} | tr ',' ' ')
    # LIB=<path to another build of librgda_hip.so> in an arm's environment: that library is put in place for the run
    [ -f $dir/regda_amd/csrc/librgda_hip.so.arm0 ] || cp $dir/regda_amd/csrc/librgda_hip.so $dir/regda_amd/csrc/librgda_hip.so.arm0
    lib=$dir/regda_amd/csrc/librgda_hip.so.arm0; for e in $envs; do case $e in LIB=*) lib=${e#LIB=} ;; esac; done
    cp $lib $dir/regda_amd/csrc/librgda_hip.so
    ( cd $dir; env $envs python bench.py --steps 10 --repeats 0 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d ${BARGS:-} 2>/dev/null | last_json "$arm" )
  done; done ;;
kstep)
  on_box; rm -rf gpurun_out/ks
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 3 --repeats 0 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-h2d --eager "$@" > gpurun_out/ks.log 2>&1
  python scripts/lib/kstep_all.py gpurun_out/ks/r_results.db 60 > gpurun_out/ks_all.txt
  python scripts/lib/klist.py gpurun_out/ks/r_results.db > gpurun_out/klist_serial.txt
  rm -rf gpurun_out/ks; head -30 gpurun_out/ks_all.txt ;;
klist)
  on_box; rm -rf gpurun_out/ks
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/ks -o r -- python bench.py --steps 4 --repeats 0 --warmup 2 --no-cpu-baseline --no-roofline --no-h2d "$@" > gpurun_out/ks.log 2>&1
  python scripts/lib/klist.py gpurun_out/ks/r_results.db > gpurun_out/klist_overlapped.txt
  python scripts/lib/timeline.py gpurun_out/ks/r_results.db 2 > gpurun_out/timeline.txt
  rm -rf gpurun_out/ks; head -12 gpurun_out/timeline.txt | cut -c1-200 ;;
dev)
  on_box; cp regda_amd/csrc/tuning/librgda_hip.so regda_amd/csrc/librgda_hip.so
  s=$1; shift
  case $s in *.sh) bash scripts/dev/$s "$@" ;; *) python scripts/dev/$s "$@" ;; esac ;;
profiles)
  on_box; tag=$1
  # ---- HBM traffic per kernel: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), kernel trace only
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$ctr
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -o p -- python bench.py --steps 2 --repeats 0 --warmup 1 --no-cpu-baseline --no-roofline --serial --eager --no-h2d > gpurun_out/pmc_$ctr.log 2>&1
  done
  python scripts/lib/pmc_traffic.py > gpurun_out/${tag}_pmc_hbm_traffic.txt
  mkdir -p profiles; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # the bench line below reads it
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  # ---- MFMA busy / wave states / LDS counters per kernel: two more passes
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
  P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  i=0; for pass in "$P1" "$P2"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_m$i
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmc_m$i -o p -- python bench.py --steps 2 --repeats 0 --warmup 1 --no-cpu-baseline --no-roofline --serial --eager --no-h2d > gpurun_out/pmc_m$i.log 2>&1
  done
  python scripts/lib/pmc_mfma.py > gpurun_out/${tag}_pmc_mfma.txt; rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2
  # ---- kernel statistics: serial (the averages the roofline probe's HIP events must agree with), one serial step, overlapped
  rm -rf gpurun_out/prof_s
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o r -- python bench.py --steps 4 --repeats 0 --warmup 1 --no-cpu-baseline --no-roofline --serial --no-h2d > gpurun_out/prof_s.log 2>&1
  python scripts/lib/prof_summary.py gpurun_out/prof_s/r_results.db 5 > gpurun_out/${tag}_kernel_stats.txt
  grep '"metric"' gpurun_out/prof_s.log >> gpurun_out/${tag}_kernel_stats.txt; rm -rf gpurun_out/prof_s
  bash scripts/tune.sh kstep > /dev/null 2>&1; cp gpurun_out/ks_all.txt gpurun_out/${tag}_kernel_one_step.txt
  rm -rf gpurun_out/prof_o
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_o -o r -- python bench.py --steps 4 --repeats 0 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/prof_o.log 2>&1
  python scripts/lib/prof_summary.py gpurun_out/prof_o/r_results.db 5 > gpurun_out/${tag}_kernel_stats_overlapped.txt
  grep '"metric"' gpurun_out/prof_o.log >> gpurun_out/${tag}_kernel_stats_overlapped.txt; rm -rf gpurun_out/prof_o
  # ---- the bench lines
  timeout 900 python bench.py --phases --full-json gpurun_out/${tag}_bench_1gpu_full.json > gpurun_out/${tag}_bench_1gpu.json 2> gpurun_out/${tag}_bench_1gpu.err
  grep ' ms  ' gpurun_out/${tag}_bench_1gpu.err > gpurun_out/${tag}_step_phases.txt
  { echo "box: $(hostname)  date: $(date -u +%FT%TZ)  commit: ${GRAFT_COMMIT:-see profiles/README.md}"; rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1; } > gpurun_out/${tag}_provenance.txt
  cut -c1-900 gpurun_out/${tag}_bench_1gpu.json; cat gpurun_out/${tag}_step_phases.txt ;;
*)
  sed -n 2,22p "$0" ;;
esac
