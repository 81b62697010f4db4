# GPU parity tests followed by the default bench (run through gpurun from the repo root).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 300 python bench.py --steps 40 --warmup 4 ${BENCH_ARGS:-} > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.json
