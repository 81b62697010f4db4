# Does the measured rate depend on the warm-up length / number of timed steps?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in "--steps 40 --warmup 4" "--steps 40 --warmup 40" "--steps 200 --warmup 40" "--steps 400 --warmup 4" "--steps 40 --warmup 4"; do
  timeout -k 10 200 python bench.py $a --no-cpu-baseline 2>gpurun_out/kw.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])" || { tail -5 gpurun_out/kw.err; exit 1; }
done
