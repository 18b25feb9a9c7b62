"""A burst of small batches queued back to back (what one rank of an N-GPU bench run sees: 20 steps of 1/N of the set, all
submitted at once): wall time of the burst and how many waves it was folded in.  usage: burst_probe.py [parts] [steps]"""
import ctypes as C, gzip, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from rafft_amd import _native as N, sharding
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open(os.path.join(ROOT, "tests/golden/bench_inputs.tsv.gz"), "rt")]
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mine = [seqs[i] for i in sharding.lpt_shards([len(s) for s in seqs], parts)[0]]
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
enc = [s.encode() for s in mine]; n = len(enc)
arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])

def burst(k):
    jobs = []
    t = time.perf_counter()
    for _ in range(k):
        job = C.c_void_p(); N.check(lib.rafft_fold_submit(C.byref(p), n, arr, lens, 0, C.byref(job))); jobs.append(job)
    t_sub = time.perf_counter() - t
    launches = 0
    for job in jobs:
        res = C.POINTER(N.Result)(); N.check(lib.rafft_fold_wait(job, C.byref(res))); lib.rafft_free_result(res)
        st = N.Stats(); lib.rafft_get_stats(C.byref(st)); launches += 1 if st.n_steps else 0
    return (time.perf_counter() - t) * 1e3, t_sub * 1e3, launches

for _ in range(3): burst(steps)
r = [burst(steps) for _ in range(5)]
print(f"{steps} batches of {n} sequences (shard 0 of {parts}) queued at once: {min(x[0] for x in r):.1f} ms best, {sorted(x[0] for x in r)[2]:.1f} ms median "
      f"(submitting takes {r[0][1]:.2f} ms); batches that carried a wave's statistics: {r[-1][2]} of {steps}; "
      f"= {len(seqs) * steps / (sorted(x[0] for x in r)[2] / 1e3) / 1e3:.0f} k sequences/s for {parts} such ranks")
