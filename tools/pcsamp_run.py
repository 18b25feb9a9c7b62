"""One warm batch + two measured batches of the benchmark workload (for profilers)."""
import gzip, sys
sys.path.insert(0, ".")
import rafft_amd
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
for _ in range(3):
    rafft_amd.fold_batch(seqs, 100, 50, 1000)
print("done", rafft_amd.last_stats()["ms_total"])
