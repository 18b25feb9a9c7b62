import gzip, time, sys
sys.path.insert(0,'.')
import rafft_amd
rows=[l.rstrip('\n').split('\t') for l in gzip.open('tests/golden/bench_inputs.tsv.gz','rt')]
seqs=[r[1] for r in rows]
for it in range(3):
    t=time.time()
    res=rafft_amd.fold_batch(seqs,100,50,1000,traj=False)
    el=time.time()-t
    print(it, len(seqs)/el,'seq/s',el, rafft_amd.last_stats(), flush=True)
nb=sum(1 for r,f in zip(rows,res) if min(f,key=lambda s:s.dcal).str_struct==r[2])
print('best_nrj match',nb,len(rows))
