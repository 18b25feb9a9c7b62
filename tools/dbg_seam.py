import sys, numpy as np
sys.path.insert(0,'.')
import oracle
from rafft_amd import rafft as R
rng=np.random.default_rng(3000)
lens = [1, 2, 5, 17, 33, 64, 129, 257, 300, 511, 700, 1025, 1500]
seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
s=seqs[-1]; L=len(s)
fin,traj=oracle.fold(s,100,8,1000,traj=True)
def regions(db):
    # loops: exterior + one per pair; region = unpaired positions directly inside
    st=[]; pt={}
    for i,c in enumerate(db):
        if c=='(': st.append(i)
        elif c==')': j=st.pop(); pt[j]=i; pt[i]=j
    out=[]
    def loop(lo,hi):
        pos=[];p=lo
        while p<hi:
            if p in pt and pt[p]>p: p=pt[p]+1
            else:
                if p not in pt: pos.append(p)
                p+=1
        return pos
    out.append(loop(0,L))
    for i in sorted(pt):
        if pt[i]>i: out.append(loop(i+1,pt[i]))
    return [r for r in out if len(r)>=1]
nbad=0; ntest=0
for step in (9,10,11):
    for st in traj[step][:8]:
        db=st.str_struct
        for pos in regions(db):
            if len(pos)<40: continue
            ntest+=1
            g=R.expand_node(s,db,pos); o=oracle.expand_node(s,db,pos)
            for k in ("lag","nb","mi","mj","score","ddcal","kept"):
                if g[k]!=o[k]:
                    bad=[i for i,(a,b) in enumerate(zip(g[k],o[k])) if a!=b]
                    nbad+=1
                    if nbad<4: print(step,len(pos),k,'MISMATCH at',bad[:5],[(g[k][i],o[k][i]) for i in bad[:3]],'lag',[g['lag'][i] for i in bad[:3]],'o',[(o['nb'][i],o['mi'][i],o['mj'][i],o['score'][i]) for i in bad[:3]],'g',[(g['nb'][i],g['mi'][i],g['mj'][i],g['score'][i]) for i in bad[:3]], 'pos head',pos[:6])
                    break
print('tested',ntest,'bad',nbad)
