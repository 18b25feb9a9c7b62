import sys, gzip, json
sys.path.insert(0,'.')
import rafft_amd
cases=json.load(gzip.open('tests/golden/fold_traj.json.gz','rt'))
groups={}
for c in cases: groups.setdefault(tuple(sorted(c["params"].items())),[]).append(c)
for key,cs in groups.items():
    print(key,len(cs),[len(c["seq"]) for c in cs],flush=True)
    got=rafft_amd.fold_batch([c["seq"] for c in cs],traj=True,**dict(key))
    ok=all([[ [s.str_struct,s.dcal] for s in st] for st in traj]==c["traj"] for c,(fin,traj) in zip(cs,got))
    print('  ok',ok,flush=True)
