#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R RAFFT_SERIAL=1
OUT=$R/gpurun_out/pmc_lanes; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU --output-format csv -d $OUT -o p -- python3 $R/tools/trace_batch.py 0 > $OUT/out.log 2> $OUT/err.log
python3 - $OUT/p_counter_collection.csv <<'PY'
import csv, collections, sys
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0][:40]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items():
    if v.get("SQ_INSTS_VALU"):
        print(k, {n:int(x) for n,x in v.items()}, "lanes/VALU-instr", round(v.get("SQ_THREAD_CYCLES_VALU",0)/v["SQ_INSTS_VALU"]/4*1,2) if v.get("SQ_THREAD_CYCLES_VALU") else None)
PY
tail -3 $OUT/err.log | cut -c1-160
