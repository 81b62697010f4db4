#!/bin/bash
# Round-2 evidence: kernel trace + PMC passes (one counter group per run) of
#   a) the default bench (fp32, 720p, batch 256)
#   b) bf16 and split-bf16 at 1920x1080, batch 512
# Run through gpurun from the repo root; summaries are made afterwards with tools/summarize_prof.py.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run_set() {   # name, bench args for the traced run, bench args for the PMC runs
  local OUT=gpurun_out/prof_$1; shift
  local TR="$1" PM="$2"
  rm -rf $OUT && mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline $TR > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; return 1; }
  echo "$OUT trace done"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline $PM > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -5 $OUT/fetch.err; return 1; }
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline $PM > $OUT/bench_write.json 2> $OUT/write.err || { tail -5 $OUT/write.err; return 1; }
  timeout -s KILL 240 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --no-cpu-baseline $PM > $OUT/bench_sq.json 2> $OUT/sq.err || { tail -5 $OUT/sq.err; return 1; }
  timeout -s KILL 240 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --no-cpu-baseline $PM > $OUT/bench_sq2.json 2> $OUT/sq2.err || { tail -5 $OUT/sq2.err; return 1; }
  timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- python3 bench.py --no-cpu-baseline $PM > $OUT/bench_grbm.json 2> $OUT/grbm.err || { tail -5 $OUT/grbm.err; return 1; }
  echo "$OUT pmc done"
  python3 tools/summarize_prof.py $OUT > $OUT/summary.md
  # keep what is merged back small: the per-dispatch CSVs are large
  find $OUT -name "*_kernel_trace.csv" -size +2M -delete
  find $OUT -name "*_counter_collection.csv" -size +2M -delete
}
run_set f32_720p "" "--steps 2 --warmup 1" || exit 1
G="--height 1080 --width 1920 --batch 512 --clip-frames 4096"
run_set bf16_1080p "--dtype bf16 $G" "--dtype bf16 $G --steps 2 --warmup 1" || exit 1
run_set split_1080p "--dtype bf16_split $G" "--dtype bf16_split $G --steps 2 --warmup 1" || exit 1
run_set split_720p "--dtype bf16_split" "--dtype bf16_split --steps 2 --warmup 1" || exit 1
du -sh gpurun_out/prof_*
