"""repeat the bench's timed loop in ONE process and print per-repeat throughput, regrowths and the slowest batch
(looking for the occasional slow run of bench.py)"""
import ctypes as C, gzip, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from rafft_amd import _native as N
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open(os.path.join(ROOT, "tests/golden/bench_inputs.tsv.gz"), "rt")]
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
enc = [s.encode() for s in seqs]; n = len(enc)
arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])
depth = int(os.environ.get("BENCH_DEPTH", "10"))
def run(steps):
    q = []; worst = 0.0; regrows = 0
    def wait(job):
        nonlocal worst, regrows
        res = C.POINTER(N.Result)(); N.check(lib.rafft_fold_wait(job, C.byref(res))); lib.rafft_free_result(res)
        st = N.Stats(); lib.rafft_get_stats(C.byref(st)); worst = max(worst, st.ms_total); regrows += st.n_regrows
    for _ in range(steps):
        job = C.c_void_p(); N.check(lib.rafft_fold_submit(C.byref(p), n, arr, lens, 0, C.byref(job))); q.append(job)
        if len(q) >= depth: wait(q.pop(0))
    while q: wait(q.pop(0))
    return worst, regrows
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: run(depth)
run(5)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t = time.perf_counter(); worst, regrows = run(20); el = time.perf_counter() - t
    print(f"rep {rep}: {n * 20 / el:9.0f} seq/s  {el / 20 * 1e3:6.3f} ms/step  slowest batch {worst:6.1f} ms  regrows {regrows}", flush=True)
    time.sleep(0.05)
