"""Golden folds beyond 16 384 nt by the CPU oracle (minutes each: generated once, committed): tests/golden/fold_verylong_<L>.json.gz.
   python tools/make_golden_verylong.py L nb_mode max_stack max_branch seed"""
import gzip, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import oracle
L, nb, ms, mb, seed = (int(x) for x in sys.argv[1:6])
rng = np.random.default_rng(seed)
s = "".join(rng.choice(list("ACGU"), L))
t = time.time()
fin, traj = oracle.fold(s, nb, ms, mb, traj=True)
out = {"generator": "tools/make_golden_verylong.py (oracle/rafft_oracle.c through oracle.fold)", "L": L, "seed": seed, "nb_mode": nb, "max_stack": ms, "max_branch": mb,
       "oracle_seconds": round(time.time() - t, 1), "sequence": s,
       "final": [[x.str_struct, x.dcal] for x in fin],
       "traj_dcal": [[x.dcal for x in st] for st in traj], "traj_pairs": [[x.str_struct.count("(") for x in st] for st in traj]}
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', f'fold_verylong_{L}.json.gz')
with gzip.open(p, "wt") as f:
    json.dump(out, f)
print(L, out["oracle_seconds"], "s", len(traj), "steps", os.path.getsize(p), "bytes")
