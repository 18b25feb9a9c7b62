#!/bin/bash
# SQ counters per kernel on the benchmark batch, synchronous calls, every kernel of a step on one stream; usage: tools/pmc_sq.sh <tag> [env...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R RAFFT_SERIAL=1 RAFFT_SPLIT=0 "$@"
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/a -o p -- python3 $R/tools/trace_batch.py 0 > $OUT/out.log 2> $OUT/err.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o p -- python3 $R/tools/trace_batch.py 0 >> $OUT/out.log 2>> $OUT/err.log
python3 - $OUT <<'PY'
import csv, collections, sys, os
tot = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter()
for sub in "ab":
    seen = set()
    f = os.path.join(sys.argv[1], sub, "p_counter_collection.csv")
    if not os.path.exists(f): continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if sub == "a" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); n[k] += 1
for k in sorted(dur, key=lambda k: -dur[k])[:10]:
    v = tot[k]
    if not v.get("SQ_WAVE_CYCLES"): continue
    wi = v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"] + v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0) + v.get("SQ_INSTS_SMEM", 0)
    print(f"{k:36s} {dur[k]/1e6:7.2f} ms {n[k]:4d} launches | Minst: valu {v['SQ_INSTS_VALU']/1e6:7.1f} salu {v['SQ_INSTS_SALU']/1e6:7.1f} lds {v['SQ_INSTS_LDS']/1e6:6.1f} vmem {(v.get('SQ_INSTS_VMEM_RD',0)+v.get('SQ_INSTS_VMEM_WR',0))/1e6:6.1f} smem {v.get('SQ_INSTS_SMEM',0)/1e6:6.1f} branch {v.get('SQ_INSTS_BRANCH',0)/1e6:6.1f}"
          f" | waves {v['SQ_WAVES']/1e3:7.1f}k | wave cycles: active {v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']:.2f} parked {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.2f} stalled {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.2f}"
          f" | lanes/VALU {v.get('SQ_THREAD_CYCLES_VALU',0)/max(v['SQ_INSTS_VALU'],1):.1f} | issue {wi/(1024*2.4e9*dur[k]*1e-9):.3f} | resident waves/SIMD {4*v['SQ_WAVE_CYCLES']/(1024*2.4e9*dur[k]*1e-9):.2f} | LDS conflict cyc {v.get('SQ_LDS_BANK_CONFLICT',0)/1e6:.1f}M")
PY
tail -2 $OUT/err.log | cut -c1-200
