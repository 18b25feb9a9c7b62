import gzip, sys
sys.path.insert(0,'.')
import rafft_amd
seqs=[l.split('\t')[1] for l in gzip.open('tests/golden/bench_inputs.tsv.gz','rt')]
for mb in (1000, 250, 50):
    rafft_amd.fold_batch(seqs,100,50,mb)
    rafft_amd.fold_batch(seqs,100,50,mb)
    st=rafft_amd.last_stats()
    print(mb, {k:round(v,2) for k,v in st.items() if k in ('ms_total','ms_expand','ms_expand_wall','ms_beam','ms_materialize','n_steps','n_children','n_structs','n_node_expansions')}, flush=True)
