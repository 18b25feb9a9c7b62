"""Repeat the bench with allocation tracing: which device / pinned allocations fall inside the timed region?"""
import subprocess, sys, os, re, json
env = dict(os.environ, RAFFT_TRACE_ALLOC="1")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    t0 = t1 = None
    inside = []
    for l in r.stderr.splitlines():
        m = re.search(r"t=([0-9.]+)", l)
        if not m:
            continue
        t = float(m.group(1))
        if "timed region starts" in l: t0 = t
        elif "timed region ends" in l: t1 = t
        elif t0 is not None and t1 is None: inside.append(l[8:90])
    print(round(d["value"]), d["ms_per_step"], "allocations inside the timed region:", len(inside), inside[:6], flush=True)
