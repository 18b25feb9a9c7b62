import ctypes as C, gzip, sys, time, os
sys.path.insert(0, ".")
from rafft_amd import _native as N
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open("tests/golden/bench_inputs.tsv.gz", "rt")]
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
R = 4
enc = [s.encode() for s in seqs] * R
n = len(enc)
arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])
os.environ["RAFFT_SPANS"] = "2"
for it in range(3):
    res = C.POINTER(N.Result)()
    N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, 0, C.byref(res)))
    lib.rafft_free_result(res)
st = N.Stats(); lib.rafft_get_stats(C.byref(st))
print({k: round(v, 2) for k, v in st.as_dict().items() if k.startswith("ms_")})
