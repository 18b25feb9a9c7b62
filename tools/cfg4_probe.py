"""BASELINE configs[3]: 16k random seqs L~U[100,3000], n=100, ms=200, sharded over 8 GPUs.
This probe folds one GPU's LPT shard (1/8 of the batch) and checks invariants."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import rafft_amd
from rafft_amd import sharding, rafft as R
rng = np.random.default_rng(3000)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lens = rng.integers(100, 3001, size=N)
seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
shards = sharding.lpt_shards([len(s) for s in seqs], 8)
mine = [seqs[i] for i in shards[0]]
print("shard 0:", len(mine), "sequences, sum L", sum(map(len, mine)), flush=True)
for call in range(2):      # the first call also allocates the HBM workspace
    t = time.time()
    res = rafft_amd.fold_batch(mine, 100, 200, 1000)
    el = time.time() - t
    st = rafft_amd.last_stats()
    print(f"call {call}: wall {el:.2f} s, lib {st['ms_total']/1e3:.3f} s, regrows {st['n_regrows']}", flush=True)
print(f"wall {el:.2f} s, lib {st['ms_total']/1e3:.2f} s, {len(mine)/(st['ms_total']/1e3):.1f} seq/s, steps {st['n_steps']}",
      {k: round(v, 1) for k, v in st.items() if k.startswith('ms_')}, flush=True)
flat = [(s, x.str_struct, x.dcal) for s, beam in zip(mine, res) for x in beam]
got, stt = R.eval_structures([f[0] for f in flat[::7]], [f[1] for f in flat[::7]])
print("energy re-evaluation mismatches:", sum(1 for g, f in zip(got, flat[::7]) if g != f[2]), "of", len(got))
