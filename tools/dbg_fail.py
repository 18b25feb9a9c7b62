import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import rafft_amd
from rafft_amd import _native
rng = np.random.default_rng(29)
rnd = lambda lens: ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
A, B = rnd(rng.integers(30, 120, size=300)), rnd(rng.integers(30, 120, size=340))
kw = dict(nb_mode=100, max_stack=10, max_branch=200)
os.environ["RAFFT_MAX_WAVES"] = "1"
os.environ["RAFFT_TEST_HARD_FAIL"] = "2"
os.environ["RAFFT_TRACE"] = "1"
A2 = A + rnd([1500, 1700])
blocker = rafft_amd.submit_batch(rnd(rng.integers(200, 400, size=400)), nb_mode=100, max_stack=50, max_branch=1000)
pa, pb = rafft_amd.submit_batch(A2, **kw), rafft_amd.submit_batch(B, **kw)
blocker.result()
try:
    r = pa.result()
    print("A2 returned", len(r), "first", r[0][0].str_struct if len(r[0]) else None, "last len", len(r[-1]))
except Exception as e:
    print("A2 raised", e)
print("B ok", len(pb.result()))
