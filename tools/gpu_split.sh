#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_bf16_split.py -x -q --timeout 120 --timeout-method thread -m gpu -s > gpurun_out/one.log 2>&1 || { tail -40 gpurun_out/one.log; exit 1; }
grep -E "split|passed" gpurun_out/one.log
dt=bf16_split
timeout -k 10 300 python bench.py --dtype $dt --no-cpu-baseline --stage-timing all > gpurun_out/bench_${dt}_stages.json 2> gpurun_out/bench_$dt.err || { tail -20 gpurun_out/bench_$dt.err; exit 1; }
python -c "import json,sys; d=json.load(open('gpurun_out/bench_${dt}_stages.json')); print('$dt', d['value'], {k:round(v['ms_total']/v['launches']*1e3,1) for k,v in d['stages'].items()})"
