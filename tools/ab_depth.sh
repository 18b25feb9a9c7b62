#!/bin/bash
# the driver's command (bench.py --steps 20 --warmup 5) for scheduler settings "waves depth merge", interleaved twice; value and steady-state value
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for cfg in "3 15 11500" "3 18 13800" "3 20 16100" "3 21 16100" "2 20 23000" "3 24 18400" "4 20 11500"; do
  set -- $cfg
  RAFFT_MAX_WAVES=$1 BENCH_DEPTH=$2 RAFFT_MERGE_SEQS=$3 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('waves $1 depth $2 merge $3:', j['value'], 'steady', j.get('steady_state_value'), 'ms/step', j['ms_per_step'])"
done
done
