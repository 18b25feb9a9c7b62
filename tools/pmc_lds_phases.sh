#!/bin/bash
# LDS bank-conflict cycles of the expand kernels on one benchmark batch, one phase of expand_kernel doubled at a time
# (RAFFT_REP bit, see tools/phase_probe.py): the difference to the first line is that phase's share of the conflicts.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R RAFFT_SERIAL=1
for rep in 0 1 2 4 8 16 32 64 128; do
  OUT=$R/gpurun_out/pmc_lds_ph/$rep; rm -rf $OUT; mkdir -p $OUT
  RAFFT_REP=$rep timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT -o p -- python3 $R/tools/trace_batch.py 0 > $OUT/out.log 2> $OUT/err.log || exit 1
  python3 - $OUT/p_counter_collection.csv $rep <<'PY'
import csv, collections, sys
tot=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].split("(")[0].replace("void expand_kernel","exp")[:22]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
print("rep", sys.argv[2], " | ".join(f"{k} {dur[k]/1e6:.1f} ms conf {v['SQ_LDS_BANK_CONFLICT']/1e6:.0f}M act {v['SQ_LDS_IDX_ACTIVE']/1e6:.0f}M n {v['SQ_INSTS_LDS']/1e6:.0f}M" for k,v in tot.items() if k.startswith("exp")), flush=True)
PY
done
