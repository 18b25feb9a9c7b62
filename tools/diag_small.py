import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rafft_amd, numpy as np
rng = np.random.default_rng(1)
seqs = ["".join(rng.choice(list("ACGU"), 60)) for _ in range(300)]
for _ in range(3):
    rafft_amd.fold_batch(seqs, 100, 5, 1000)
