"""Profiling helper: doubles one phase of expand_kernel at a time (RAFFT_REP bit) and
prints the per-kernel time deltas.  Results stay correct (phases are idempotent)."""
import gzip, os, subprocess, sys, json
code = r'''
import gzip, sys, json
sys.path.insert(0, ".")
import rafft_amd
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
rafft_amd.fold_batch(seqs, 100, 50, 1000)
rafft_amd.fold_batch(seqs, 100, 50, 1000)
print(json.dumps(rafft_amd.last_stats()))
'''
base = None
for rep in (0, 1, 2, 4, 8):
    env = dict(os.environ, RAFFT_REP=str(rep))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    st = json.loads(out)
    if rep == 0:
        base = st
    print(f"rep={rep:2d} expand {st['ms_expand']:8.2f} ms  (+{st['ms_expand'] - base['ms_expand']:7.2f})  total {st['ms_total']:8.2f}", flush=True)
