"""latency of rafft.fold on ONE sequence (the reference's own use): wall per call and the library's phases (RAFFT_TRACE=1 on the last call)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import rafft_amd
rng = np.random.default_rng(1)
for L, ms in ((76, 50), (200, 50), (400, 50), (1500, 50), (16384, 4), (32768, 4), (32768, 1), (76, 1)):
    s = "".join(rng.choice(list("ACGU"), L))
    rafft_amd.fold(s, 100, ms, 100)
    ts = []
    for _ in range(20 if L < 4000 else 3):
        t = time.perf_counter(); rafft_amd.fold(s, 100, ms, 100); ts.append(time.perf_counter() - t)
    st = rafft_amd.last_stats()
    print(f"L={L} ms={ms}: wall median {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms (min {1e3 * min(ts):.3f}), lib {st['ms_total']:.3f} ms, steps {st['n_steps']}", file=sys.stderr, flush=True)
os.environ["RAFFT_TRACE"] = "1"
rafft_amd.fold(s, 100, 1, 100)
