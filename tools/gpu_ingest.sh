#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -m gpu -k "ingest or decoded or driver_loop or cli" > gpurun_out/one.log 2>&1 || { tail -40 gpurun_out/one.log; exit 1; }
tail -3 gpurun_out/one.log
timeout -k 10 600 python -u tools/ingest_rate.py 2>&1 | tee gpurun_out/ingest_rate.txt
