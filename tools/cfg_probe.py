import sys, time, numpy as np
sys.path.insert(0,'.')
import rafft_amd, oracle
# cfg2: 1000 random L=200, n=100 ms=50
rng=np.random.default_rng(200)
seqs=["".join(rng.choice(list("ACGU"),200)) for _ in range(1000)]
for it in range(2):
    t=time.time(); res=rafft_amd.fold_batch(seqs,100,50,1000); el=time.time()-t
    st=rafft_amd.last_stats(); print('cfg2',it,len(seqs)/ (st['ms_total']/1e3),'seq/s lib', el, {k:round(v,2) for k,v in st.items() if k.startswith('ms_')}, flush=True)
bad=0
for s,r in list(zip(seqs,res))[:60]:
    o=oracle.fold(s,100,50,1000)
    bad+= [(x.str_struct,x.dcal) for x in o]!=[(x.str_struct,x.dcal) for x in r]
print('cfg2 mismatches vs oracle (60 checked):',bad, flush=True)
# cfg5: one 400-nt, ms=1000, traj
rng=np.random.default_rng(400)
s5="".join(rng.choice(list("ACGU"),400))
t=time.time(); fin,traj=rafft_amd.fold(s5,100,1000,1000,traj=True); el=time.time()-t
st=rafft_amd.last_stats(); print('cfg5 steps',len(traj),'final',len(fin),'lib ms',st['ms_total'],'wall',el, flush=True)
t=time.time(); ofin,otraj=oracle.fold(s5,100,1000,1000,traj=True); print('oracle s',time.time()-t, flush=True)
print('cfg5 traj equal:', [[(x.str_struct,x.dcal) for x in stp] for stp in traj]==[[(x.str_struct,x.dcal) for x in stp] for stp in otraj], flush=True)
# cfg4-like: 64 seqs L~U[100,3000], ms=200
rng=np.random.default_rng(3000)
lens=rng.integers(100,3001,size=64)
seqs4=["".join(rng.choice(list("ACGU"),int(n))) for n in lens]
t=time.time(); res4=rafft_amd.fold_batch(seqs4,100,200,1000); el=time.time()-t
st=rafft_amd.last_stats(); print('cfg4-like 64 seqs lib ms',st['ms_total'],'wall',el, 'steps',st['n_steps'], flush=True)
# check 6 shortest vs oracle
idx=np.argsort(lens)[:6]
bad=0
for i in idx:
    o=oracle.fold(seqs4[i],100,200,1000)
    bad+= [(x.str_struct,x.dcal) for x in o]!=[(x.str_struct,x.dcal) for x in res4[i]]
print('cfg4-like mismatches (6 shortest):',bad, [int(lens[i]) for i in idx], flush=True)
