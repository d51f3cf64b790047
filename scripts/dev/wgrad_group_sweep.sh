cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for g in 500 250 120 60 30; do
  python bench.py --steps 10 --repeats 0 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --wgrad-group-gflop $g 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('group_gflop $g  ms/step %.3f' % d['ms_per_step'])"
done; done
