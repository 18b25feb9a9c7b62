import sys, time, numpy as np
import os; sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import rafft_amd, oracle
rng=np.random.default_rng(5)
def rnd(n): return "".join(rng.choice(list("ACGU"),int(n)))
def key(b): return [(x.str_struct,x.dcal) for x in b]
# 1. very large batch of short sequences
seqs=[rnd(n) for n in rng.integers(20,90,size=40000)]
t=time.time(); res=rafft_amd.fold_batch(seqs,100,5,100); print("40k short:",time.time()-t, rafft_amd.last_stats()['n_regrows'], flush=True)
for k in range(0,40000,1999): assert key(res[k])==key(oracle.fold(seqs[k],100,5,100)), k
# 2. large beam on a medium batch
seqs=[rnd(n) for n in rng.integers(100,400,size=200)]
t=time.time(); res=rafft_amd.fold_batch(seqs,100,1000,1000); print("200 x ms=1000:",time.time()-t, rafft_amd.last_stats()['n_regrows'], flush=True)
for k in (0,57,199): assert key(res[k])==key(oracle.fold(seqs[k],100,1000,1000)), k
# 3. max_branch huge, beam small
seqs=[rnd(n) for n in rng.integers(100,300,size=300)]
t=time.time(); res=rafft_amd.fold_batch(seqs,100,3,10000); print("mb=10000:",time.time()-t, flush=True)
for k in (0,150,299): assert key(res[k])==key(oracle.fold(seqs[k],100,3,10000)), k
# 4. many identical sequences (same loops in different sequences must not be confused)
s=rnd(150); seqs=[s]*500
res=rafft_amd.fold_batch(seqs,100,20,1000); o=key(oracle.fold(s,100,20,1000))
assert all(key(r)==o for r in res); print("500 identical ok", flush=True)
# 5. concurrent heavy + traj batches of different params in flight
pend=[rafft_amd.submit_batch([rnd(n) for n in rng.integers(50,500,size=600)],100,ms,1000,traj=tr) for ms,tr in ((50,False),(10,True),(50,False),(200,False),(1,True),(50,False))]
for p in pend: r=p.result(); assert len(r)==600
print("mixed in-flight ok", flush=True)
print("ALL OK")
