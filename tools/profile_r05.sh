#!/bin/bash
# ONE profiling script, run last, at the build that is committed.  rocprofv3 passes on the GPU box, from the repo root:
#   A. BASELINE configs[2] (the headline workload)
#      1. kernel trace + stats of the timed loop exactly as the driver runs it, WITHOUT the untimed pre-warm
#      2.-6. PMC passes, each in its own run, synchronous calls (4 batches): FETCH_SIZE | WRITE_SIZE | SQ issue | SQ mix | lanes + LDS conflicts
#   B. BASELINE configs[3], one GPU's LPT shard (2048 sequences, L 100..3000, ms=200; tools/ab_cfg4.py, 3 calls)
#      7. kernel trace + stats   8.-9. FETCH_SIZE | WRITE_SIZE   10. SQ issue  11. lanes
#   C. BASELINE configs[3] whole on one GPU (BENCH_CFG4=1 python bench.py): its bench line
#   D. the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5)
#   (round 5: + 1b. a SERIAL kernel trace of the headline workload beside the pipelined one - tools/dominant.py)
# The digest of rafft_amd/csrc of the tree that RAN goes beside the raw files; tools/pmc_summary.py (run in the build container, where
# git knows the commit) turns gpurun_out/r05_prof into profiles/r05_*.  (Under rocprofv3 the program itself follows `--`.)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_prof
rm -rf $OUT; mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_digest())" > $OUT/csrc_digest.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
BENCH_PREWARM_S=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/cu_share.py $OUT/trace/t_kernel_trace.csv > $OUT/cu_share.json || true
echo "trace done" >> $OUT/progress.log
# 1b. (round 5) the same kernels SERIAL: every kernel of a step on one stream, one wave at a time, no long-tail split - what each kernel
#     costs when it has the chip to itself (the pipelined trace above charges a kernel for the time it queues for CUs).  60 batches
#     (tools/ab_bench.py 20: 20 sequential + 2 x 20 at depth 1); tools/dominant.py sets the two side by side
(RAFFT_SERIAL=1 RAFFT_SPLIT=0 AB_DEPTH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o t -- python3 $R/tools/ab_bench.py 20 > $OUT/serial_run.log 2> $OUT/serial.err) || echo "serial trace failed" >> $OUT/progress.log
rm -f $OUT/serial/t_kernel_trace.csv
echo "serial trace done" >> $OUT/progress.log
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
unset BENCH_DEPTH BENCH_PREWARM_S
# ---- B. the configs[3] shard
C4=$OUT/cfg4; mkdir -p $C4
A="python3 $R/tools/ab_cfg4.py"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $C4/trace -o t -- $A > $C4/trace_run.log 2> $C4/trace.err
python3 $R/tools/cu_share.py $C4/trace/t_kernel_trace.csv > $C4/cu_share.json || true
echo "cfg4 trace done" >> $OUT/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $C4/pmc_fetch -o p -- $A > $C4/pmc_fetch.log 2> $C4/pmc_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $C4/pmc_write -o p -- $A > $C4/pmc_write.log 2> $C4/pmc_write.err
echo "cfg4 traffic done" >> $OUT/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $C4/pmc_sq -o p -- $A > $C4/pmc_sq.log 2> $C4/pmc_sq.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $C4/pmc_lanes -o p -- $A > $C4/pmc_lanes.log 2> $C4/pmc_lanes.err || echo "cfg4 lanes failed" >> $OUT/progress.log
echo "cfg4 sq done" >> $OUT/progress.log
# the kernel traces of the PMC passes are not needed beside the counter files, and the raw trace of the cfg3 loop is big
rm -f $OUT/pmc_*/p_kernel_trace.csv $C4/pmc_*/p_kernel_trace.csv
# ---- C. configs[3] whole on one GPU
cd $R && BENCH_CFG4=1 timeout -k 10 500 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_cfg4_n1.json 2> $OUT/bench_cfg4_n1.err || echo "cfg4 whole failed" >> $OUT/progress.log
# ---- D. the driver's command itself, last (its line is the round's bench record: profiles/r05_bench_n1.json)
cd $R && timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r05_bench_n1.json 2> $R/gpurun_out/r05_bench_n1.err || echo "bench failed" >> $OUT/progress.log
echo "all done" >> $OUT/progress.log
du -sh $OUT
