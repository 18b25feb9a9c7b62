"""A/B on one GPU's LPT shard of BASELINE configs[3] (2048 sequences, L 100..3000, ms=200): library ms per call."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
from rafft_amd import _native as N
if os.environ.get('AB_LIB'):
    N.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
import rafft_amd
from rafft_amd import sharding
rng = np.random.default_rng(3000)
lens = rng.integers(100, 3001, size=16384)
seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
mine = [seqs[i] for i in sharding.lpt_shards([len(s) for s in seqs], 8)[0]]
os.environ.setdefault("RAFFT_SPANS", "2")
out = []
for call in range(3):
    rafft_amd.fold_batch(mine, 100, 200, 1000)
    st = rafft_amd.last_stats()
    out.append(st)
st = out[-1]
print(f"lib {st['ms_total']:.1f} ms  c1 {st['ms_expand']:.1f}  c2 {st['ms_expand_c2']:.1f}  c3 {st['ms_expand_c3']:.1f}  expand wall {st['ms_expand_wall']:.1f}  "
      f"beam {st['ms_beam']:.1f}  materialize {st['ms_materialize']:.1f}  regrows {st['n_regrows']}", flush=True)
