#!/bin/bash
# same-box A/B of two builds of the library on the bench loop: tools/ab_libs.sh <libA.so> <libB.so> [rounds]   (interleaved runs)
R=${GRAFT_REPO_ROOT:-$PWD}
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    RAFFT_LIB=$R/$L python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], 'expand64 mean launch ms', d['roofline']['mean_launch_ms'], 'frac', d['roofline']['frac'])"
  done
done
