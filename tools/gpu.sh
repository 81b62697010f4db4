#!/bin/bash
# Everything this repository runs on the GPU box goes through this one script (from the repo root, via gpurun):
#
#   tools/gpu.sh tests [pytest args]        the GPU suite in one process (default: all of tests/), then the
#                                           allocation-free test alone in a fresh process
#   tools/gpu.sh tests-options              the whole GPU suite once per alternative kernel / schedule, selected through the
#                                           handles' environment defaults (CUTDET_RESIZE_ROWS, CUTDET_S2_TWO_WAVES, ...)
#   tools/gpu.sh one <pytest args>          one pytest target, output shown (-s)
#   tools/gpu.sh bench NAME [bench args]    python bench.py [args] -> gpurun_out/bench/NAME.json (+ one summary line)
#   tools/gpu.sh profile NAME "TRACE ARGS" "PMC ARGS"
#                                           rocprofv3 kernel trace of `bench.py TRACE ARGS`, then one PMC pass per
#                                           counter group of `bench.py PMC ARGS`, summarised by tools/summarize_prof.py
#                                           into gpurun_out/prof_NAME/summary.md (copy what is to be kept to profiles/)
#   tools/gpu.sh stats NAME [bench args]   rocprofv3 --kernel-trace --stats of bench.py (per-kernel durations)
#   tools/gpu.sh trace NAME [bench args]   kernel trace of bench.py + tools/trace_overlap.py (timeline, overlap)
#   tools/gpu.sh py <script> [args]         a python tool under a time limit (tools/*.py)
#   tools/gpu.sh ablate [range_stages args] omission ablation of the range call: tools/range_stages.py with each stage
#                                           (and pairs of stages) left out - needs the timing-experiment build
#                                           `make -C cut-detection_amd/csrc variant NAME=skip FLAGS=-DCD_EXP_SKIP`
#
# Every GPU step runs under its own time limit and the script stops at the first failure.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
MODE=$1; shift

summary_line() {   # file, name
python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
line = "%-22s %10.0f f/s  ms/step %8.3f  conv2 %.4f ms frac %.4f  whole-net %.3f" % (
    sys.argv[2], d["value"], d["ms_per_step"], r["launch_ms"], r["frac"], r["whole_net_frac_of_mfma_peak"])
for k in ("fp32_mfma_block1", "config4", "strong_200k", "independent_clips"):
    if k in d:
        line += "  %s %.0f f/s" % (k, d[k]["value"])
if d.get("cpu_baseline"):
    line += "  cpu %.0f f/s" % d["cpu_baseline"]["value"]
print(line)
PY
}

case "$MODE" in
tests-options)
  for e in CUTDET_RESIZE_ROWS=1 CUTDET_S2_TWO_WAVES=1 CUTDET_F32_BLOCK1_FP32_MFMA=1 CUTDET_FUSED12=1 CUTDET_RANGE_MODE=1; do
    echo "== $e"
    env $e timeout -k 10 600 python3 -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu -p no:cacheprovider \
      > gpurun_out/suite_$e.log 2>&1 || { tail -40 gpurun_out/suite_$e.log; exit 1; }
    tail -1 gpurun_out/suite_$e.log
  done
  ;;
ablate)
  LIBV=$PWD/cut-detection_amd/lib/variants/libcutdet_skip.so
  [ -f "$LIBV" ] || { echo "build the skip variant first (see the header of this script)"; exit 1; }
  for m in 0 1 2 4 8 3 6 5; do     # 1 resize pass, 2 block 1, 4 block 2, 8 block 3 + head
    echo "== stages left out: mask $m"
    CUTDET_SKIP=$m CUTDET_LIB=$LIBV timeout -k 10 300 python3 -u tools/range_stages.py "$@" | grep "range entry" || exit 1
  done
  ;;
tests)
  T="${@:-tests}"
  timeout -k 10 1100 python3 -u -m pytest $T -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/gputests.log 2>&1 \
    || { tail -60 gpurun_out/gputests.log; exit 1; }
  tail -3 gpurun_out/gputests.log
  timeout -k 10 300 python3 -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread -m gpu \
    > gpurun_out/noalloc.log 2>&1 || { tail -30 gpurun_out/noalloc.log; exit 1; }
  tail -2 gpurun_out/noalloc.log
  ;;
one)
  timeout -k 10 900 python3 -u -m pytest "$@" -x -q --timeout 600 --timeout-method thread -m gpu -s > gpurun_out/one.log 2>&1
  rc=$?; tail -60 gpurun_out/one.log; exit $rc
  ;;
bench)
  NAME=$1; shift
  mkdir -p gpurun_out/bench
  timeout -k 10 500 python3 bench.py "$@" > gpurun_out/bench/$NAME.json 2> gpurun_out/bench/$NAME.err \
    || { echo "$NAME failed"; tail -20 gpurun_out/bench/$NAME.err; exit 1; }
  summary_line gpurun_out/bench/$NAME.json $NAME
  ;;
profile)
  NAME=$1; TR="$2"; PM="$3"
  OUT=gpurun_out/prof_$NAME
  rm -rf $OUT && mkdir -p $OUT
  SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"
  SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  echo "trace: bench.py --no-cpu-baseline $TR ; pmc: bench.py --no-cpu-baseline $PM" > $OUT/commands.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline $TR \
    > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
  pass() {  # dir, counters...
    local d=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -- python3 bench.py --no-cpu-baseline $PM \
      > $OUT/bench_$d.json 2> $OUT/$d.err || { tail -5 $OUT/$d.err; exit 1; }
  }
  pass pmc_fetch FETCH_SIZE
  pass pmc_write WRITE_SIZE
  pass pmc_sq $SQ1
  pass pmc_sq2 $SQ2
  python3 tools/summarize_prof.py $OUT > $OUT/summary.md
  cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
  # keep what is merged back small: the per-dispatch CSVs are large
  find $OUT -name "*_kernel_trace.csv" -size +2M -delete
  find $OUT -name "*_counter_collection.csv" -size +2M -delete
  cat $OUT/summary.md
  ;;
stats)   # NAME [bench args]: rocprofv3 --kernel-trace --stats of bench.py, the per-kernel stats table kept
  NAME=$1; shift
  OUT=gpurun_out/stats_$NAME
  rm -rf $OUT && mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --no-cpu-baseline "$@" \
    > $OUT/bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
  cp $OUT/t/*/*_kernel_stats.csv $OUT/kernel_stats.csv
  find $OUT -name "*_kernel_trace.csv" -size +2M -delete
  head -6 $OUT/kernel_stats.csv | cut -c1-60,150-400; cut -c1-400 $OUT/bench.json
  ;;
trace)   # NAME [bench args]: kernel trace only, timeline + overlap analysis kept under gpurun_out/trace_NAME/
  NAME=$1; shift
  OUT=gpurun_out/trace_$NAME
  rm -rf $OUT && mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --no-cpu-baseline "$@" \
    > $OUT/bench.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
  python3 tools/trace_overlap.py $OUT/t > $OUT/overlap.txt
  find $OUT -name "*_kernel_trace.csv" -size +8M -delete
  tail -12 $OUT/overlap.txt; cat $OUT/bench.json | cut -c1-300
  ;;
py)
  timeout -k 10 600 python3 -u "$@"
  ;;
*)
  echo "usage: tools/gpu.sh tests|one|bench|profile|py ..."; exit 2
  ;;
esac
