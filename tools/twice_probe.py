import os, subprocess, sys, json
code = r'''
import gzip, sys, json
sys.path.insert(0, ".")
import rafft_amd
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
for i in range(3): rafft_amd.fold_batch(seqs, 100, 50, 1000)
print(json.dumps(rafft_amd.last_stats()))
'''
for tw in ("0", "1", "2"):
    env = dict(os.environ, RAFFT_TWICE=tw, RAFFT_SERIAL="1", RAFFT_SPANS="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    st = json.loads(out)
    print(tw, {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}, st["n_regrows"], flush=True)
