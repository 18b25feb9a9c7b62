#!/bin/bash
run() { echo -n "$* : "; env "$@" AB_DEPTH=8 timeout -k 10 120 python tools/ab_bench.py 40 2>/dev/null | sed 's/sequential median/seq/; s/(min [0-9.]*)//; s/; regrows.*//' || exit 1; }
run X=base
run RAFFT_DEDUPE_PER_CU=8
run AB_LIB=rafft_amd/libraffthip_m5.so
run AB_LIB=rafft_amd/libraffthip_m6.so
run AB_LIB=rafft_amd/libraffthip_m8.so
run X=base
