# Bench at other source geometries (1080p: 3000-frame clip to bound the HBM footprint).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for g in "1080 1920 3000" "180 320 10000"; do
  set -- $g
  echo "== $2 x $1"
  timeout -k 10 300 python bench.py --steps 24 --warmup 2 --no-cpu-baseline --height $1 --width $2 --clip-frames $3 2>gpurun_out/geom.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v['ms_total'] for k,v in d['stages'].items()})" || tail -5 gpurun_out/geom.err
done
