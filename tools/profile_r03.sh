#!/bin/bash
# rocprofv3 passes over the bench workload (run on the GPU box from the repo root):
#   1. kernel trace + stats of the timed loop exactly as the driver runs it, WITHOUT the untimed pre-warm (BENCH_PREWARM_S=0: the
#      AverageNs of the stats CSV is then the number DESIGN.md quotes for the timed loop's launches, warm-up steps included)
#   2.-6. PMC passes, each in its own run, synchronous calls (one batch at a time, 4 batches):
#      FETCH_SIZE | WRITE_SIZE | SQ issue | SQ mix | lanes per VALU instruction + LDS bank conflicts
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
BENCH_PREWARM_S=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/cu_share.py $OUT/trace/t_kernel_trace.csv > $OUT/cu_share.json || true
echo "trace done" >> $OUT/progress.log
export BENCH_DEPTH=1 BENCH_PREWARM_S=0     # PMC passes: synchronous calls, exactly steps + warmup = 4 batches
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $B --steps 3 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o p -- $B --steps 3 --warmup 1 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
echo "sq done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o p -- $B --steps 3 --warmup 1 > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err || echo "sq2 failed" >> $OUT/progress.log
echo "sq2 done" >> $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc_lanes -o p -- $B --steps 3 --warmup 1 > $OUT/pmc_lanes.json 2> $OUT/pmc_lanes.err || echo "lanes failed" >> $OUT/progress.log
echo "lanes done" >> $OUT/progress.log
python3 $R/tools/pmc_summary.py $OUT $OUT r03 > $OUT/summary.log 2>&1 || true
tail -20 $OUT/summary.log
