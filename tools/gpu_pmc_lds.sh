# One PMC pass: LDS counters for every kernel of the default bench run.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_lds
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
echo done
