#!/bin/bash
# round 5: wave-instructions of the one-wavefront expand kernel BY PHASE.  RAFFT_TWICE=k launches the kernel a second time on the same
# work list with every region stopping after a phase (3: window_slide, 4: ranking, 5: lag values, 6: correlation, 7: LDS fill,
# 8: header, 9: claim; 1: the whole region again); the SQ instruction counters of that run minus those of a run without the second
# launch are what the phases up to there issue.  General build (the skip levels are compiled out of the production build),
# twelve wavefronts per workgroup (the only packing that carries the skip bits), small-region classes off.
#   tools/pmc_phases.sh [levels...]      -> gpurun_out/pmc_phases.txt
R=${GRAFT_REPO_ROOT:-$PWD}
LEVELS=${@:-0 1 3 4 5 6 7 8}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R RAFFT_SERIAL=1 RAFFT_SPLIT=0 RAFFT_WPB=12 RAFFT_SMALL=0,0
: > $R/gpurun_out/pmc_phases.txt
for tw in $LEVELS; do
  OUT=$R/gpurun_out/pmc_ph_$tw; rm -rf $OUT; mkdir -p $OUT
  RAFFT_TWICE=$tw timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT -o p -- python3 $R/tools/trace_batch.py 0 > $OUT/out.log 2> $OUT/err.log
  python3 - $OUT/p_counter_collection.csv $tw >> $R/gpurun_out/pmc_phases.txt <<'PY'
import csv, collections, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("expand_kernel<64"): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in tot.items():
    print("TWICE", sys.argv[2], k, "launches", len(n[k]), " ".join(f"{c}={int(x)}" for c, x in sorted(v.items())))
PY
  rm -rf $OUT
done
cat $R/gpurun_out/pmc_phases.txt
