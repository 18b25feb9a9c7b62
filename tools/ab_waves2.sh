#!/bin/bash
# steady state (160 batches per setting, interleaved twice) for bigger merged waves: "waves depth merge"
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for cfg in "3 15 11500" "3 21 16100" "2 20 23000" "3 30 23000" "2 14 16100" "3 27 20700" "2 28 32200"; do
  set -- $cfg
  RAFFT_MAX_WAVES=$1 AB_DEPTH=$2 RAFFT_MERGE_SEQS=$3 python3 $R/tools/ab_bench.py 160 2>/dev/null | sed "s/^/waves $1 depth $2 merge $3: /" | cut -c1-150
done
done
