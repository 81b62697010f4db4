set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 40 --warmup 4 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.json
