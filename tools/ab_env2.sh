#!/bin/bash
# like tools/ab_env.sh with the steady-state figure: tools/ab_env2.sh "A=1" "B=2 C=3" ...   ("-" = no setting)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[$cfg]', j['value'], 'ms/step', j['ms_per_step'], 'steady', j.get('steady_state_value'), 'alloc', j['allocations_in_timed_region']['device_buffers'])"
done
done
