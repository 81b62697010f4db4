#!/bin/bash
# parity tests that exercise the general resize, then 1080p / 180p bench lines for each arithmetic
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_bf16_split.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/one.log 2>&1 || { tail -40 gpurun_out/one.log; exit 1; }
tail -2 gpurun_out/one.log
for cfg in "f32 1080 1920 512" "bf16 1080 1920 512" "bf16_split 1080 1920 512" "f32 180 320 256"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --dtype $1 --height $2 --width $3 --batch $4 --clip-frames 8192 --no-cpu-baseline --stage-timing all > gpurun_out/bench_geom.json 2> gpurun_out/bench_geom.err || { tail -20 gpurun_out/bench_geom.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_geom.json')); print('$cfg', d['value'], {k:round(v['ms_total']/v['launches']*1e3,1) for k,v in d['stages'].items()})" | tee -a gpurun_out/geom_rates.txt
done
