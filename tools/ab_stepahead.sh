#!/bin/bash
# round 5: the host issuing a step ahead of its read-backs (default) against lock-step (RAFFT_STEP_AHEAD=0), interleaved, on the driver's
# command with its extras: headline, steady state, one synchronous call, the Python API's fold_batch
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for sa in 0 1; do
  RAFFT_STEP_AHEAD=$sa python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[STEP_AHEAD=$sa]', j['value'], 'steady', j.get('steady_state_value'), 'ms/call sequential', j.get('ms_per_call_sequential'), 'submit+wait', j.get('ms_per_call_submit_wait'), 'py fold_batch seq/s', (j.get('python_api') or {}).get('fold_batch_sequences_per_s'), 'x4 call', (j.get('larger_batch_info') or {}).get('sequences_per_s'))"
done
done
