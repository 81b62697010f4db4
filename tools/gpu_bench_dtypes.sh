#!/bin/bash
# bench lines for the non-headline arithmetic modes (and per-stage timing of each)
set -o pipefail
mkdir -p gpurun_out
for dt in bf16_split bf16; do
  timeout -k 10 300 python bench.py --dtype $dt --no-cpu-baseline > gpurun_out/bench_$dt.json 2> gpurun_out/bench_$dt.err || { tail -20 gpurun_out/bench_$dt.err; exit 1; }
  cat gpurun_out/bench_$dt.json
  timeout -k 10 300 python bench.py --dtype $dt --no-cpu-baseline --stage-timing all > gpurun_out/bench_${dt}_stages.json 2> gpurun_out/bench_$dt.err || { tail -20 gpurun_out/bench_$dt.err; exit 1; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_${dt}_stages.json')); print('$dt', d['value'], {k:round(v['ms_total']/v['launches']*1e3,1) for k,v in d['stages'].items()})"
done
