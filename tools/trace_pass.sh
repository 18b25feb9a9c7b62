#!/bin/bash
# pass A.1 of tools/profile_r04.sh alone (kernel trace + stats of the driver's command without the untimed pre-warm), then the driver's command itself
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_prof
mkdir -p $OUT; rm -rf $OUT/trace
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_digest())" > $OUT/csrc_digest.txt
cd /tmp && export TMPDIR=/tmp
BENCH_PREWARM_S=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err
python3 $R/tools/cu_share.py $OUT/trace/t_kernel_trace.csv > $OUT/cu_share.json || true
cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r04_bench_n1.json 2> $R/gpurun_out/r04_bench_n1.err
