# PMC passes for the memory pipe (TA / TCP / TD) of the convolution kernels.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mem
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 tools/run_conv_loop.py 10 > $OUT/a.log 2>&1
echo pass a done
timeout -s KILL 120 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum --kernel-trace --output-format csv -d $OUT/b -- python3 tools/run_conv_loop.py 10 > $OUT/b.log 2>&1
echo pass b done
timeout -s KILL 120 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_sum TCP_LFIFO_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/c -- python3 tools/run_conv_loop.py 10 > $OUT/c.log 2>&1
echo pass c done
find $OUT -name "*counter_collection.csv"
