#!/bin/bash
# serial per-kernel times (one stream, one wave at a time) of ONE build under environment settings: tools/ab_serial_env.sh "A=1" "B=2" ...  ("-": none)
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  OUT=$R/gpurun_out/abse_$$; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && env $envs TMPDIR=/tmp RAFFT_SERIAL=1 RAFFT_SPLIT=0 AB_DEPTH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/ab_bench.py 20 > $OUT/run.log 2>&1)
  python3 - "$OUT/t_kernel_stats.csv" "$cfg" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nb = 20 + 2 * 20
print(f"[{sys.argv[2]}] ms per batch:", "; ".join(f"{r['Name'].split('(')[0].replace('void ','')[:34]} {float(r['TotalDurationNs'])/1e6/nb:.3f}" for r in rows[:9]))
PY
  rm -rf $OUT
done
