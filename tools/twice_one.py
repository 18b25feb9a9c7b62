import gzip, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import rafft_amd
seqs = [l.split("\t")[1] for l in gzip.open(os.path.join(ROOT, "tests/golden/bench_inputs.tsv.gz"), "rt")]
for i in range(3):
    rafft_amd.fold_batch(seqs, 100, 50, 1000)
