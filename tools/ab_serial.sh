#!/bin/bash
# serial per-kernel times (every kernel of a step on ONE stream, one wave at a time, no long-tail split) of two builds on one box:
#   tools/ab_serial.sh <libA.so> <libB.so>     -> per-kernel ms per batch for each, interleaved twice
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for L in $1 $2; do
  OUT=$R/gpurun_out/abs_$(basename $L .so)_$rep; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp RAFFT_LIB=$R/$L AB_LIB=$R/$L RAFFT_SERIAL=1 RAFFT_SPLIT=0 AB_DEPTH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/ab_bench.py 20 > $OUT/run.log 2>&1)
  python3 - "$OUT/t_kernel_stats.csv" "$L" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nb = 20 + 2 * 20      # ab_bench.py 20: 20 sequential + 2 x 20 "pipelined" (depth 1) batches
out = []
for r in rows[:9]:
    out.append(f"{r['Name'].split('(')[0].replace('void ','')[:34]} {float(r['TotalDurationNs'])/1e6/nb:.3f}")
print(sys.argv[2], "| ms per batch:", "; ".join(out))
PY
  rm -rf $OUT
done
done
