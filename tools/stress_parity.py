"""One-off stress run: random sequences (incl. low-complexity ones with many ties) x random parameters, GPU
trajectories against the CPU oracle.  Both expand routings (RAFFT_MERGE_*)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import oracle, rafft_amd

def as_lists(traj):
    return [[[s.str_struct, s.dcal] for s in st] for st in traj]

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2024)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for case in range(n_cases):
    nb_mode = int(rng.choice([3, 20, 50, 100, 100, 200, 500]))
    ms = int(rng.choice([1, 2, 5, 20, 50, 120]))
    mb = int(rng.choice([1, 7, 50, 100, 1000, 1000]))
    hp = int(rng.choice([3, 3, 3, 1, 5]))
    w = [(3.0, 2.0, 1.0), (3.0, 2.0, 1.0), (1.0, 1.0, 1.0), (2.5, 1.5, 0.25), (3.0, 2.0, 0.0)][int(rng.integers(0, 5))]
    alpha = [list("ACGU"), list("GC"), list("AU"), list("GGGCCCAU"), list("ACGUN")][int(rng.integers(0, 5))]
    lens = [int(x) for x in rng.integers(5, 260, size=6)] + [int(rng.integers(300, 700))]
    if ms <= 20:      # a long sequence too: its small loops take the one-wavefront class by their span, its big ones the wide classes
        lens.append(int(rng.integers(1300, 2600)))
    seqs = ["".join(rng.choice(alpha, n)) for n in lens]
    for mode in (0, 1):
        if mode == 0:
            os.environ["RAFFT_MERGE_BELOW"] = "0"; os.environ["RAFFT_MERGE2_BELOW"] = "0"
        else:
            os.environ.pop("RAFFT_MERGE_BELOW", None); os.environ.pop("RAFFT_MERGE2_BELOW", None)
        got = rafft_amd.fold_batch(seqs, nb_mode, ms, mb, hp, 0.0, True, 37.0, *w)
        for s, (fin, traj) in zip(seqs, got):
            _, o = oracle.fold(s, nb_mode, ms, mb, hp, 0.0, True, 37.0, *w)
            if as_lists(traj) != as_lists(o):
                bad += 1
                print("MISMATCH", case, mode, len(s), nb_mode, ms, mb, hp, w, s, flush=True)
    if case % 5 == 4:
        print(f"case {case + 1}/{n_cases}  mismatches {bad}  {time.time() - t0:.0f} s", flush=True)
print("done, mismatches:", bad)
sys.exit(1 if bad else 0)
