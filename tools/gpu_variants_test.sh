# Parity tests, then the bench with each library variant under cut-detection_amd/lib/variants.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -2 gpurun_out/gputests.log
for lib in cut-detection_amd/lib/libcutdet.so cut-detection_amd/lib/libcutdet.so $(ls cut-detection_amd/lib/variants/*.so 2>/dev/null); do
  echo "== $lib"
  CUTDET_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --stage-timing all 2>gpurun_out/var.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms_total'] for k,v in d['stages'].items()})" || { tail -5 gpurun_out/var.err; exit 1; }
done
