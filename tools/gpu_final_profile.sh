# Full evidence run: micro-benchmarks, per-stage bench breakdown, bf16 lines, then the rocprof passes.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
{ echo "== tools/micro/mfma_rate"; timeout -k 10 120 tools/micro/mfma_rate; echo "== tools/micro/coexec_bench"; timeout -k 10 120 tools/micro/coexec_bench; echo "== tools/micro/gather_bench"; timeout -k 10 120 tools/micro/gather_bench; } > gpurun_out/final/microbench.txt 2>&1
echo micro done
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --stage-timing all > gpurun_out/final/bench_stage_all.json 2> gpurun_out/final/err.txt
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --dtype bf16 --stage-timing all > gpurun_out/final/bench_bf16_720p.json 2>> gpurun_out/final/err.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --dtype bf16 --height 1080 --width 1920 --batch 512 --clip-frames 4096 > gpurun_out/final/bench_bf16_1080p_b512.json 2>> gpurun_out/final/err.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --height 1080 --width 1920 --batch 512 --clip-frames 4096 > gpurun_out/final/bench_f32_1080p_b512.json 2>> gpurun_out/final/err.txt
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --height 180 --width 320 > gpurun_out/final/bench_f32_180p.json 2>> gpurun_out/final/err.txt
echo benches done
bash tools/gpu_profile.sh
