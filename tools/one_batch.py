import gzip, sys
sys.path.insert(0,'.')
import rafft_amd
seqs=[l.split('\t')[1] for l in gzip.open('tests/golden/bench_inputs.tsv.gz','rt')]
rafft_amd.fold_batch(seqs,100,50,1000)
