"""Profiling helper: doubles one phase of expand_kernel at a time (RAFFT_REP bit) and prints the
per-class kernel time deltas, classes serialized on one stream (RAFFT_SERIAL=1) so that the numbers
are standalone.  Results stay correct (phases are idempotent)."""
import os, subprocess, sys, json
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
names = {0: "baseline", 1: "FFT/corr x2", 2: "rank sort x2", 4: "window_slide x2", 8: "dE x2", 16: "LDS fill x2", 32: "direct corr+values x2", 64: "emit x2", 128: "mask setup x2"}
for rep in (0, 1, 2, 4, 8, 16, 32, 64, 128):
    env = dict(os.environ, RAFFT_REP=str(rep), RAFFT_SERIAL="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    st = json.loads(out)
    if rep == 0:
        base = st
    print(f"{names[rep]:16s} tiny {st['ms_expand_c1']:7.2f} (+{st['ms_expand_c1'] - base['ms_expand_c1']:6.2f})  small {st['ms_expand']:7.2f} (+{st['ms_expand'] - base['ms_expand']:6.2f})  medium {st['ms_expand_c2']:7.2f} (+{st['ms_expand_c2'] - base['ms_expand_c2']:6.2f})"
          f"  large {st['ms_expand_c3']:7.2f} (+{st['ms_expand_c3'] - base['ms_expand_c3']:6.2f})  total {st['ms_total']:7.2f}", flush=True)
