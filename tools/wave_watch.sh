#!/bin/bash
# the waves bench.py's timed region is folded as (RAFFT_TRACE=1): tools/wave_watch.sh <runs> [lib.so]
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-3}; L=${2:-rafft_amd/libraffthip.so}
for i in $(seq $N); do
  RAFFT_LIB=$R/$L RAFFT_TRACE=1 RAFFT_TRACE_ALLOC=1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/ww_$i.json 2> $R/gpurun_out/ww_$i.err
  python3 - "$R/gpurun_out/ww_$i.json" "$R/gpurun_out/ww_$i.err" <<'PY'
import json, re, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
inside = False; waves = []
for l in open(sys.argv[2], errors="replace"):
    if "timed region starts" in l: inside = True; continue
    if "timed region ends" in l: inside = False; continue
    m = re.search(r"wave S=(\d+) setup ([0-9.]+) ms, loop ([0-9.]+) ms \((\d+) steps\), tail ([0-9.]+)", l)
    if m and inside: waves.append((int(m.group(1)), float(m.group(3)), int(m.group(4))))
print("run", d["value"], "ms/step", d["ms_per_step"], "waves (S, loop ms, steps):", waves)
PY
done
