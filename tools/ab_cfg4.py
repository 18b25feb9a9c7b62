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
# (round 5) algorithmic bytes and HIP-event time per kernel family of the last call: a roofline fraction for this workload too
fam = [("expand_kernel<64>", "alg_bytes_expand", "ms_expand"), ("expand_small_kernel", "alg_bytes_expand_small", "ms_expand_c1"),
       ("expand_kernel<256> (up to 1024 positions)", "alg_bytes_expand_c2", "ms_expand_c2"),
       ("expand_kernel<256,false,1,2,3> + <512> (1025..4096 positions)", "alg_bytes_expand_c3", "ms_expand_c3"),
       ("beam_step + materialize + dedupe", "alg_bytes_beam", None)]
for name, ab, ms in fam:
    t = st[ms] if ms else st["ms_beam"] + st["ms_materialize"]
    print(f"  {name}: {st[ab] / 1e6:.1f} MB algorithmic in {t:.1f} ms = {st[ab] / max(t, 1e-9) / 1e6:.1f} GB/s = {st[ab] / max(t, 1e-9) / 1e6 / 8000 * 100:.3f} % of 8 TB/s", flush=True)
