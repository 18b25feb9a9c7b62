import os, subprocess, sys, json
code = r'''
import gzip, sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import rafft_amd
seqs = [l.split("\t")[1] for l in gzip.open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests/golden/bench_inputs.tsv.gz"), "rt")]
for i in range(3): rafft_amd.fold_batch(seqs, 100, 50, 1000)
print(json.dumps(rafft_amd.last_stats()))
'''
for tw in ("0", "1", "3", "4", "5", "6", "7", "8", "9"):
    env = dict(os.environ, RAFFT_TWICE=tw, RAFFT_SERIAL="1", RAFFT_SPANS="2", RAFFT_SMALL="0,0", RAFFT_SPLIT="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
    st = json.loads(out)
    print(tw, {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}, st["n_regrows"], flush=True)
