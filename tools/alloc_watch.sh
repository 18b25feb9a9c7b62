#!/bin/bash
# which device allocations fall into bench.py's timed region: tools/alloc_watch.sh <runs> [lib.so]   (RAFFT_TRACE_ALLOC=1)
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-3}; L=${2:-rafft_amd/libraffthip.so}
for i in $(seq $N); do
  RAFFT_LIB=$R/$L RAFFT_TRACE_ALLOC=1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/aw_$i.json 2> $R/gpurun_out/aw_$i.err
  python3 - "$R/gpurun_out/aw_$i.json" "$R/gpurun_out/aw_$i.err" <<'PY'
import json, re, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t0 = None; inside = []; before = 0
for l in open(sys.argv[2]):
    m = re.search(r"timed region starts t=([0-9.]+)", l)
    if m: t0 = float(m.group(1)); continue
    m = re.search(r"device buffer -> ([0-9.]+) MB in ([0-9.]+) ms", l)
    if m:
        if t0 is None: before += 1
        else: inside.append((float(m.group(1)), float(m.group(2))))
print("run", d["value"], "allocations before the timed region", before, "inside", len(inside), "MB", round(sum(x for x, _ in inside), 1), "ms", round(sum(y for _, y in inside), 1),
      "biggest", sorted(inside)[-3:], "| regrows", d.get("regrows_in_timed_region"), "latency", d["kernel_ms_per_step"], "launches/step", d["roofline"]["launches_per_step"], "expand64 ms", d["roofline"]["mean_launch_ms"])
PY
done
