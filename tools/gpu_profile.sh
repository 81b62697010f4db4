# Kernel-trace + PMC profiles of the default bench run (run through gpurun from the repo root).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
echo trace done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
echo write done
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/sq.err
echo sq done
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_grbm.json 2> $OUT/grbm.err
echo grbm done
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/bench_sq2.json 2> $OUT/sq2.err
echo sq2 done
find $OUT -name "*.csv" | head -50
du -sh $OUT
