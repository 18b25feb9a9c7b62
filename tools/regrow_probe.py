"""Prints wall time and arena regrowths (n_regrows) of typical workloads: the planner should never need one."""
import gzip, sys, time
import numpy as np
sys.path.insert(0, ".")
import rafft_amd
rng = np.random.default_rng(3)
rnd = lambda L: "".join(rng.choice(list("ACGU"), int(L)))
bench = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
cases = {
    "bench set ms=50": (bench, 100, 50, 1000, False),
    "bench set x2 ms=50": (bench * 2, 100, 50, 1000, False),
    "cfg2 1000 x L=200 ms=50": ([rnd(200) for _ in range(1000)], 100, 50, 1000, False),
    "cfg4-like 512 x L 100..3000 ms=200": ([rnd(L) for L in rng.integers(100, 3001, size=512)], 100, 200, 1000, False),
    "cfg5 L=400 ms=1000 traj": ([rnd(400)], 100, 1000, 1000, True),
    "10 x L=300 ms=50": ([rnd(300) for _ in range(10)], 100, 50, 1000, False),
    "100 x L=100 ms=1": ([rnd(100) for _ in range(100)], 100, 1, 100, False),
    "3 x L=3000 ms=50": ([rnd(3000) for _ in range(3)], 100, 50, 1000, False),
    "bench set ms=50 traj": (bench, 100, 50, 1000, True),
    "64 x L=800 ms=100 mb=5000": ([rnd(800) for _ in range(64)], 100, 100, 5000, False),
}
for name, (seqs, n, ms, mb, traj) in cases.items():
    for it in range(2):
        t = time.time()
        rafft_amd.fold_batch(seqs, n, ms, mb, traj=traj)
        el = time.time() - t
        st = rafft_amd.last_stats()
    print(f"{name:40s} C-ABI {st['ms_total']:9.2f} ms  regrows {st['n_regrows']}  steps {st['n_steps']}", flush=True)
