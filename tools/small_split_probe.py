import sys, time, os
import numpy as np
sys.path.insert(0, ".")
import rafft_amd
rng = np.random.default_rng(5)
for S in (64, 200, 400):
    lens = [int(x) for x in rng.integers(60, 300, size=S)]
    lens[3] = 2000; lens[S // 2] = 2400
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    for mode in ("0", "-1"):
        os.environ["RAFFT_SPLIT"] = mode
        ts = []
        for it in range(5):
            rafft_amd.fold_batch(seqs, 100, 50, 1000)
            ts.append(rafft_amd.last_stats()["ms_total"])
        print(f"S={S} RAFFT_SPLIT={mode}: {min(ts[1:]):.2f} ms", flush=True)
