# PMC passes for LDS / instruction-issue behaviour of the convolution kernels.
# CUTDET_LIB selects the library build (default: the in-tree one).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 tools/run_conv_loop.py 10 > $OUT/a.log 2>&1
echo pass a done
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/b -- python3 tools/run_conv_loop.py 10 > $OUT/b.log 2>&1
echo pass b done
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/c -- python3 tools/run_conv_loop.py 10 > $OUT/c.log 2>&1
echo pass c done
