"""BASELINE configs[3] whole on one GPU, two calls, RAFFT_TRACE=1: the per-wave lines (setup / loop / tail) of the second call."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import rafft_amd
rng = np.random.default_rng(3000)
lens = rng.integers(100, 3001, size=16384)
seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
rafft_amd.fold_batch(seqs, 100, 200, 1000)
os.environ["RAFFT_TRACE"] = "1"
t = time.perf_counter()
r = rafft_amd.fold_batch(seqs, 100, 200, 1000)
print(f"second call {1e3 * (time.perf_counter() - t):.1f} ms wall, lib {rafft_amd.last_stats()['ms_total']:.1f} ms", file=sys.stderr, flush=True)
