"""Throughput of rafft_fold_batch against the batch size (benchmark set replicated R times in ONE call)."""
import ctypes as C, gzip, sys, time
sys.path.insert(0, ".")
from rafft_amd import _native as N
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
for R in (1, 2, 4, 8):
    enc = [s.encode() for s in seqs] * R
    n = len(enc)
    arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])
    best = 1e9
    for it in range(4):
        res = C.POINTER(N.Result)()
        t = time.perf_counter()
        N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, 0, C.byref(res)))
        el = time.perf_counter() - t
        lib.rafft_free_result(res)
        if it: best = min(best, el)
    print(f"R={R}: {n} sequences in {best*1e3:.2f} ms -> {n/best:.0f} seq/s", flush=True)
