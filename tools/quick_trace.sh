#!/bin/bash
# one rocprofv3 kernel trace of the bench loop (20 timed + 5 warm-up steps, no pre-warm, no extras): per-kernel totals
# usage (GPU box, repo root): bash tools/quick_trace.sh TAG      -> gpurun_out/TAG_kernel_stats.csv, gpurun_out/TAG_trace_bench.json
set -e
TAG=${1:-qt}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${TAG}_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_PREWARM_S=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_trace_bench.json 2> $OUT/trace.err
cp $OUT/t_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
python3 $R/tools/cu_share.py $OUT/t_kernel_trace.csv > $R/gpurun_out/${TAG}_cu_share.json || true
python3 $R/tools/concurrency.py $OUT/t_kernel_trace.csv > $R/gpurun_out/${TAG}_concurrency.txt || true; rm -f $OUT/*.db
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6/25:8.3f} ms/batch avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
