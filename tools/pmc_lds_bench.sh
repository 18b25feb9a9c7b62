#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_lds_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RAFFT_SERIAL=1 PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p -- python3 $R/tools/trace_batch.py 0 > $OUT/out.log 2> $OUT/err.log
python3 - <<'PY'
import csv, collections, os
f=os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/pmc_lds_bench/p_counter_collection.csv')
tot=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:40]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k,v in tot.items():
    if 'expand' in k: print(k, round(dur[k]/1e6,1), 'ms', {n:int(x) for n,x in v.items()})
PY
