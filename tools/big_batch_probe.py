"""Stage times of one call holding REP copies of the benchmark set (one wave of REP x 2296 sequences, nothing else in
flight): where a well-filled wave spends its time.  RAFFT_SPANS=2 records every stage."""
import gzip, os, sys, json
sys.path.insert(0, ".")
os.environ.setdefault("RAFFT_SPANS", "2")
import rafft_amd
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")] * rep
for i in range(3):
    rafft_amd.fold_batch(seqs, 100, 50, 1000)
if len(sys.argv) > 2:
    os.environ["RAFFT_TRACE"] = sys.argv[2]
    rafft_amd.fold_batch(seqs, 100, 50, 1000)
st = rafft_amd.last_stats()
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k.startswith("n_")}))
