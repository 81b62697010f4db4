# Default bench with and without the per-kernel HIP events (what the instrumentation costs).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "--stage-timing=all" "--timing-every=1" "--stage-timing=off" "--timing-every=4" "--timing-every=1" "--timing-every=4"; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline $a 2>gpurun_out/bt.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bt.err; exit 1; }
done
