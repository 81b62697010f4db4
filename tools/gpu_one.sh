#!/bin/bash
# one pytest target on the GPU: tools/gpu_one.sh <pytest args...>
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest "$@" -x -q --timeout 300 --timeout-method thread -m gpu -s > gpurun_out/one.log 2>&1
rc=$?
tail -40 gpurun_out/one.log
exit $rc
