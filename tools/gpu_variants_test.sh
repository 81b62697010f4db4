# Runs the GPU parity tests and the bench against every library variant under lib/variants.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in cut-detection_amd/lib/libcutdet.so cut-detection_amd/lib/variants/*.so; do
  echo "== $lib"
  CUTDET_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
  CUTDET_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>gpurun_out/var.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms_total'] for k,v in d['stages'].items()})" || { tail -5 gpurun_out/var.err; exit 1; }
done
