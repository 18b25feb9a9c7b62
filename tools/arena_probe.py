"""Arena usage per workload (RAFFT_TRACE=1 prints used/cap of every arena at the end of a wave): what plan_caps' factors are fitted to.
   python tools/arena_probe.py 2> gpurun_out/arena_probe.log"""
import os, sys, gzip, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
os.environ["RAFFT_TRACE"] = "1"
import rafft_amd
from rafft_amd import sharding


def show(tag, seqs, ms, mb=1000, **kw):
    print(f"==== {tag}: {len(seqs)} sequences, mean L {np.mean([len(s) for s in seqs]):.0f}, ms {ms}", file=sys.stderr, flush=True)
    rafft_amd.fold_batch(seqs, 100, ms, mb, **kw)
    st = rafft_amd.last_stats()
    print(f"stats {tag}: structs {st['n_structs']} instances {st['n_node_instances']} created {st['n_nodes_created']} regrows {st['n_regrows']}", file=sys.stderr, flush=True)


root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
bench = [l.split("\t")[1].strip() for l in gzip.open(os.path.join(root, "tests/golden/bench_inputs.tsv.gz"), "rt") if l.strip()]
show("benchmark set", bench, 50)
rng = np.random.default_rng(200)
show("configs[1] 1000 x 200 nt", ["".join(rng.choice(list("ACGU"), 200)) for _ in range(1000)], 50)
rng = np.random.default_rng(7)
show("4000 x 40 nt", ["".join(rng.choice(list("ACGU"), 40)) for _ in range(4000)], 50)
show("benchmark set ms=1", bench, 1)
show("benchmark set ms=400 mb=100", bench[::4], 400, 100)
rng = np.random.default_rng(3000)
lens = rng.integers(100, 3001, size=16384)
seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
show("configs[3] shard", [seqs[i] for i in sharding.lpt_shards([len(s) for s in seqs], 8)[0]], 200)
rng = np.random.default_rng(11)
show("8 x 8000 nt ms=20", ["".join(rng.choice(list("ACGU"), 8000)) for _ in range(8)], 20)
show("GC-rich 500 x 600 nt", ["".join(rng.choice(list("GC"), 600)) for _ in range(500)], 50)
