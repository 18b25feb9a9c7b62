#!/bin/bash
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python tools/ab_bench.py 40 2>/dev/null | sed 's/sequential.*; pipelined/pipelined/' || exit 1; }
for rep in 1 2 3; do
run AB_DEPTH=1
run AB_DEPTH=4
run AB_DEPTH=8
run AB_DEPTH=12
done
