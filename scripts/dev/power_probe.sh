#!/bin/bash
# dev: is the step power-limited?  Samples rocm-smi (average socket power, sclk / mclk, temperature, perf level) every
# 0.5 s while `bench.py` runs 400 steps, and once on the idle chip before.  usage: bash scripts/dev/power_probe.sh
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
out=gpurun_out/power_probe.txt
{ echo "## idle"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|Performance|Max Graphics" ; } > $out
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> $out
python bench.py --steps 1500 --repeats 0 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d > gpurun_out/power_bench.log 2>&1 &
pid=$!
# (the first `import torch` on a fresh box takes a minute or two: sample until the run ends, keep the busy samples)
i=0
while kill -0 $pid 2>/dev/null; do
  i=$((i+1))
  { echo "## sample $i"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|Temperature \(Sensor junction" ; } > gpurun_out/.pp
  grep -q "sclk clock level: S: (9[0-9]Mhz)\|sclk clock level: S: (1[0-9][0-9]Mhz)" gpurun_out/.pp || cat gpurun_out/.pp >> $out
  sleep 1
done
wait $pid
grep '"metric"' gpurun_out/power_bench.log | cut -c1-200 >> $out
cat $out
