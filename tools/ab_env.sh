#!/bin/bash
# the driver's command under environment settings, interleaved: tools/ab_env.sh "A=1" "B=2 C=3" ...   ("-" = no setting)
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[$cfg]', j['value'], 'ms/step', j['ms_per_step'], 'expand64 launch ms', j['roofline']['mean_launch_ms'])"
done
done
