#!/bin/bash
# steady-state throughput of the bench workload for scheduler settings (80 batches per setting, interleaved twice):
#   tools/ab_waves.sh     -> lines "waves W depth D merge M: ... seq/s"
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for cfg in "2 10 11500" "3 15 11500" "3 12 9200" "4 16 9200" "3 18 13800" "2 8 9200"; do
  set -- $cfg
  RAFFT_MAX_WAVES=$1 AB_DEPTH=$2 RAFFT_MERGE_SEQS=$3 python3 $R/tools/ab_bench.py 80 2>/dev/null | sed "s/^/waves $1 depth $2 merge $3: /" | cut -c1-150
done
done
