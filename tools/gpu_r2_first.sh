#!/bin/bash
# round-2 first GPU pass: no-alloc test alone, the GPU suite, bench N=1, 2-rank rehearsal on one GPU, strong 200k
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread -m gpu > gpurun_out/noalloc.log 2>&1 || { tail -30 gpurun_out/noalloc.log; exit 1; }
tail -3 gpurun_out/noalloc.log
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 300 python bench.py > gpurun_out/bench1.json 2> gpurun_out/bench1.err || { tail -20 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.json
CUTDET_BENCH_BACKEND=gloo CUTDET_BENCH_ONE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench2.json 2> gpurun_out/bench2.err || { tail -20 gpurun_out/bench2.err; exit 1; }
cat gpurun_out/bench2.json
timeout -k 10 300 python bench.py --total-frames 200000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_strong.json 2> gpurun_out/bench_strong.err || { tail -20 gpurun_out/bench_strong.err; exit 1; }
cat gpurun_out/bench_strong.json
