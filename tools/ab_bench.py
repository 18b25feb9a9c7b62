"""A/B comparisons of builds on the benchmark set: latency of one synchronous call, throughput with two batches in
flight (the bench loop), and the summed duration of the dominant kernel per batch.  AB_LIB=<path> picks the build."""
import ctypes as C, gzip, os, statistics, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from rafft_amd import _native as N
from rafft_amd.rafft import _params
seqs = [l.split("\t")[1] for l in gzip.open(os.path.join(ROOT, "tests/golden/bench_inputs.tsv.gz"), "rt")]
if os.environ.get('AB_LIB'):
    N.LIB_PATH = os.path.abspath(os.environ['AB_LIB'])
lib = N.lib(); N.check(lib.rafft_init(0))
p = _params(100, 50, 1000, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
enc = [s.encode() for s in seqs]; n = len(enc)
arr = (C.c_char_p * n)(*enc); lens = (C.c_int * n)(*[len(e) for e in enc])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 14

def submit():
    job = C.c_void_p()
    N.check(lib.rafft_fold_submit(C.byref(p), n, arr, lens, 0, C.byref(job)))
    return job

def wait(job):
    res = C.POINTER(N.Result)()
    N.check(lib.rafft_fold_wait(job, C.byref(res)))
    lib.rafft_free_result(res)
    st = N.Stats(); lib.rafft_get_stats(C.byref(st))
    global regrows, worst
    regrows += st.n_regrows; worst = max(worst, st.ms_total)
    return st.ms_expand

ts, ex = [], []
regrows, worst = 0, 0.0
for it in range(iters):
    t = time.perf_counter()
    e = wait(submit())
    if it >= 2: ts.append((time.perf_counter() - t) * 1e3); ex.append(e)
depth = int(os.environ.get("AB_DEPTH", "2"))
for rnd in range(2):
    q = []
    t = time.perf_counter()
    for it in range(iters):
        q.append(submit())
        if len(q) >= depth: wait(q.pop(0))
    while q: wait(q.pop(0))
    tp = (time.perf_counter() - t) / iters * 1e3
print(f"sequential median {statistics.median(ts):.3f} ms (min {min(ts):.3f}); pipelined x{depth} {tp:.3f} ms/batch = {n / tp * 1e3:.0f} seq/s; "
      f"expand<64> {statistics.median(ex):.3f} ms/batch; regrows {regrows}, slowest batch {worst:.1f} ms", flush=True)
