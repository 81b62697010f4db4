# Default bench with and without the per-kernel HIP events (what the instrumentation costs).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "--stage-timing=all" "--stage-timing=dominant" "--stage-timing=off" "--stage-timing=dominant"; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline $a 2>gpurun_out/bt.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bt.err; exit 1; }
done
