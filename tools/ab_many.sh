#!/bin/bash
# interleaved A/B of several builds on the bench loop (no extras): tools/ab_many.sh rounds lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$PWD}
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    RAFFT_LIB=$R/$L python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], 'expand64 mean launch ms', d['roofline']['mean_launch_ms'], 'frac', d['roofline']['frac'], 'allocs', d.get('allocations_in_timed_region'), 'regrows', d.get('regrows_in_timed_region'))"
  done
done
