"""Median/min wall time of rafft_fold_batch on the benchmark set over N calls (A/B comparisons of builds)."""
import ctypes as C, gzip, statistics, sys, time
sys.path.insert(0, ".")
from rafft_amd import _native as N
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
import os
if os.environ.get('AB_LIB'):
    N.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
enc = [s.encode() for s in seqs]; n = len(enc)
arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])
ts = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    res = C.POINTER(N.Result)()
    t = time.perf_counter()
    N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, 0, C.byref(res)))
    el = time.perf_counter() - t
    lib.rafft_free_result(res)
    if it >= 2: ts.append(el * 1e3)
print(f"median {statistics.median(ts):.3f} ms  min {min(ts):.3f}  max {max(ts):.3f}  ({n / statistics.median(ts) * 1e3:.0f} seq/s)", flush=True)
