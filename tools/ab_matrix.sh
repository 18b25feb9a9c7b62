#!/bin/bash
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/ab_cfg4.py 2>/dev/null || exit 1; }
run RAFFT_TAB=0
run RAFFT_TAB=4
run RAFFT_TAB=8
run RAFFT_TAB=12
run RAFFT_TAB=12 RAFFT_NT2=512
run RAFFT_TAB=0 RAFFT_NT2=512
