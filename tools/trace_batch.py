"""Per-step timeline of one benchmark batch (RAFFT_TRACE=1 on the second call; the first warms the workspace)."""
import gzip, os, sys
sys.path.insert(0, '.')
import rafft_amd
seqs = [l.split('\t')[1] for l in gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests/golden/bench_inputs.tsv.gz'), 'rt')]
rafft_amd.fold_batch(seqs, 100, 50, 1000)
rafft_amd.fold_batch(seqs, 100, 50, 1000)
os.environ["RAFFT_TRACE"] = sys.argv[1] if len(sys.argv) > 1 else "1"
rafft_amd.fold_batch(seqs, 100, 50, 1000)
print(rafft_amd.last_stats())
