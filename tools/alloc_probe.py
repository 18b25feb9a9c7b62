import subprocess, sys, os, re
env=dict(os.environ, RAFFT_TRACE_ALLOC="1")
for rep in range(3):
    r=subprocess.run([sys.executable,"bench.py","--steps","20","--warmup","5","--no-cpu-baseline","--no-extras"],env=env,capture_output=True,text=True)
    lines=[l for l in r.stderr.splitlines() if l.startswith("[rafft]")]
    import json
    d=json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    big=[l for l in lines if float(re.search(r"in ([0-9.]+) ms",l).group(1))>1.0] if lines else []
    print(d["value"], d["ms_per_step"], "allocs", len(lines), "slow(>1ms):", [l[8:60] for l in big][:8])
