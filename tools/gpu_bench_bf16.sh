# bf16-mode bench lines (non-headline): 720p batch 256 and the 1080p batch-512 configuration.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --dtype bf16 > gpurun_out/bench_bf16_720p.json 2> gpurun_out/bf.err || { tail -5 gpurun_out/bf.err; exit 1; }
cat gpurun_out/bench_bf16_720p.json
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --dtype bf16 --height 1080 --width 1920 --batch 512 --clip-frames 4096 > gpurun_out/bench_bf16_1080p.json 2> gpurun_out/bf.err || { tail -5 gpurun_out/bf.err; exit 1; }
cat gpurun_out/bench_bf16_1080p.json
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --height 1080 --width 1920 --batch 512 --clip-frames 4096 > gpurun_out/bench_f32_1080p.json 2> gpurun_out/bf.err || { tail -5 gpurun_out/bf.err; exit 1; }
cat gpurun_out/bench_f32_1080p.json
