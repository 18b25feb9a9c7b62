import sys, gzip, json
sys.path.insert(0,'.')
import oracle
from rafft_amd import rafft as R
recs=json.load(gzip.open('tests/golden/node_expand.json.gz','rt'))
nbad=0
for i,r in enumerate(recs):
    g = R.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
    o = oracle.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
    if g["kept"]!=o["kept"] or g["ddcal"]!=o["ddcal"]:
        nbad+=1
        if nbad<4:
            print(i,len(r["pos"]),r["nb_mode"],'kept',g["kept"],o["kept"],'dd',[g["ddcal"][k] for k in o["kept"]],[o["ddcal"][k] for k in o["kept"]], 'nb', [g['nb'][k] for k in o['kept']])
print('bad',nbad,len(recs))
