#!/bin/bash
# every bench line the documentation quotes, one JSON line per file under gpurun_out/bench_all/
set -o pipefail
O=gpurun_out/bench_all; rm -rf $O; mkdir -p $O
run() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }; python - "$O/$name.json" "$name" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%-22s %10.0f f/s  ms/step %8.3f  conv2 %.4f ms frac %.4f  whole-net %.3f  cpu %s" % (sys.argv[2], d["value"], d["ms_per_step"], r["launch_ms"], r["frac"], r["whole_net_frac_of_mfma_peak"], (d.get("cpu_baseline") or {}).get("value")))
PY
}
if [ "$1" != "multi" ]; then
run f32_720p
run f32_720p_stages --no-cpu-baseline --stage-timing all
run split_720p --dtype bf16_split --no-cpu-baseline
run split_720p_stages --dtype bf16_split --no-cpu-baseline --stage-timing all
run bf16_720p --dtype bf16 --no-cpu-baseline
run bf16_720p_stages --dtype bf16 --no-cpu-baseline --stage-timing all
G="--height 1080 --width 1920 --batch 512 --clip-frames 8192 --no-cpu-baseline"
run f32_1080p_b512 $G
run split_1080p_b512 --dtype bf16_split $G
run bf16_1080p_b512 --dtype bf16 $G
run f32_180p --height 180 --width 320 --no-cpu-baseline
run strong_200k --total-frames 200000 --steps 3 --warmup 1 --no-cpu-baseline
fi
CUTDET_BENCH_BACKEND=gloo CUTDET_BENCH_ONE_GPU=1 run two_ranks_one_gpu --gpus 2 --steps 5 --warmup 2
CUTDET_BENCH_BACKEND=gloo CUTDET_BENCH_ONE_GPU=1 run eight_ranges_one_gpu --gpus 8 --total-frames 200000 --steps 1 --warmup 1
CUTDET_BENCH_BACKEND=gloo CUTDET_BENCH_ONE_GPU=1 run eight_clips_one_gpu --gpus 8 --independent-clips --clip-frames 4096 --steps 3 --warmup 1
