#!/bin/bash
# 1080p / 180p bench lines: overlapped resize pipeline vs one fused call per batch
set -o pipefail
mkdir -p gpurun_out
for cfg in "f32 1080 1920 512" "bf16 1080 1920 512" "bf16_split 1080 1920 512" "f32 180 320 256" "bf16_split 180 320 256"; do
  set -- $cfg
  for ov in "" "--no-overlap"; do
    timeout -k 10 300 python bench.py --dtype $1 --height $2 --width $3 --batch $4 --clip-frames 8192 --no-cpu-baseline $ov > gpurun_out/bench_geom.json 2> gpurun_out/bench_geom.err || { tail -20 gpurun_out/bench_geom.err; exit 1; }
    python -c "import json,sys; d=json.load(open('gpurun_out/bench_geom.json')); print('$cfg $ov', d['value'], d['roofline']['launch_ms'])" | tee -a gpurun_out/geom_rates2.txt
  done
done
