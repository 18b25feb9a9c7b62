#!/bin/bash
run() { echo -n "$* : "; env "$@" AB_DEPTH=8 timeout -k 10 120 python tools/ab_bench.py 40 2>/dev/null | sed 's/sequential median/seq/; s/(min [0-9.]*)//; s/; regrows.*//' || exit 1; }
run X=base
run RAFFT_MAX_WAVES=3
run RAFFT_WPB=10
run RAFFT_WPB=10 RAFFT_MAX_WAVES=3
run RAFFT_WPB=10 RAFFT_MAX_WAVES=4
run RAFFT_WPB=8 RAFFT_MAX_WAVES=3
run RAFFT_WPB=8 RAFFT_MAX_WAVES=4
