#!/bin/bash
# the whole GPU suite in one process, then the allocation-free test alone in a fresh one
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/gputests.log 2>&1 || { tail -40 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/noalloc.log 2>&1 || { tail -30 gpurun_out/noalloc.log; exit 1; }
tail -2 gpurun_out/noalloc.log
