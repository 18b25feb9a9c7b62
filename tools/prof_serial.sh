#!/bin/bash
# per-kernel durations with every kernel of a step on ONE stream (RAFFT_SERIAL=1: no overlap between the expand classes),
# synchronous calls: what each kernel costs when it has the chip to itself.  usage: tools/prof_serial.sh <tag> [env...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/ps_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RAFFT_SERIAL=1 AB_DEPTH=1 "$@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/ab_bench.py 12 > $OUT/run.log 2>&1
f=$(find $OUT -name "t_kernel_stats.csv" | head -1)
echo "== $TAG $@"; tail -1 $OUT/run.log; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
