"""Golden vectors ABOVE n = 2381, where the reference's correlation goes through scipy's fp64 FFT
(rafft/utils.py:115-122: scipy.signal.convolve(method="auto") picks the FFT for inputs that long; below it
convolves directly and the values are exact integers).  Container only: imports the reference's own Python
with the stand-in `RNA` module of tools/make_golden.py (energies from the KAT-pinned oracle evaluator,
everything else REFERENCE code, scipy included).

Outputs (tests/golden/):
  fold_traj_long.json.gz    full trajectories of the two 23S benchmark sequences (2915, 2968 nt) and of two
                            random sequences (2500, 3000 nt) at max_stack 1 and 5, nb_mode 100, max_branch 1000
                            [+ the headline configuration max_stack 50 for the two 23S sequences when run with --ms50]
  node_expand_long.json.gz  per-region records for every region with n >= 2381 met in those runs: the
                            reference's ranked top-nb_mode lags with their values, window_slide tuples and
                            the kept candidates.  (The full correlation profile - 2n-1 doubles - is not stored:
                            the exact profile is recomputed by the test; what the FFT noise can change is the
                            ORDER of exactly tied lags and, when a tie straddles the cut, the SET.)

Usage: python tools/make_golden_long.py [--ms50] [--jobs N]
"""
import gzip
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FFT_FROM = 2381      # scipy 1.15.3 switches to the FFT at this input length


def sequences():
    import csv
    bench = list(csv.reader(open("/root/reference/benchmark_results/benchmark_cleaned_all_length.csv")))
    longs = sorted((r[0] for r in bench if len(r[0]) >= FFT_FROM), key=len)
    assert [len(s) for s in longs] == [2915, 2968], [len(s) for s in longs]
    rng = np.random.default_rng(2381)
    rnd = ["".join(rng.choice(list("ACGU"), L)) for L in (2500, 3000)]
    return longs + rnd


def ranges(pos):
    """ascending positions as [start, length] runs (a 3000-entry list per record otherwise)"""
    out = []
    for p in pos:
        if out and out[-1][0] + out[-1][1] == p:
            out[-1][1] += 1
        else:
            out.append([int(p), 1])
    return out


def one_case(args):
    si, seq, ms = args
    import make_golden as MG       # installs the stand-in RNA module, imports the reference
    R, U = MG.R, MG.U
    recs = []
    orig = MG._orig_create

    def rec_create_childs(upair, cur_str, gp):
        n = len(upair.pos_list)
        if n >= FFT_FROM:
            cor_l = U.auto_cor(upair.forward, upair.backward)
            cs = sorted(cor_l, key=lambda el: el[1])
            ranked = cs[::-1][:gp.nb_mode]
            ws = [R.window_slide(upair.forward, upair.backward, pos, upair.pos_list, gp.min_hp) for pos, _ in ranked]
            sol = R.find_best_consecutives(cs, upair, cur_str, gp)
            recs.append(dict(seq=si, db_pairs=[[int(a), int(b)] for a, b in cur_str.pair_list], pos=ranges(upair.pos_list),
                             nb_mode=gp.nb_mode, min_hp=gp.min_hp, max_stack=ms,
                             lags=[int(p) for p, _ in ranked], vals=[float(v).hex() for _, v in ranked],
                             ws=[[int(a), int(b), int(c), float(d)] for a, b, c, d in ws],
                             sol=[[int(s[0]), float(s[1]), int(s[2]), int(s[3]), int(round(s[4] * 100))] for s in sol]))
        return orig(upair, cur_str, gp)

    R.create_childs = rec_create_childs
    t0 = time.time()
    fin, traj = R.fold(seq, nb_mode=100, max_stack=ms, max_branch=1000, traj=True)
    el = time.time() - t0
    print(f"[long golden] seq {si} (L={len(seq)}) ms={ms}: {len(traj)} steps, {len(recs)} regions >= {FFT_FROM}, {el:.0f} s", flush=True)
    case = dict(seq=si, params=dict(nb_mode=100, max_stack=ms, max_branch=1000),
                traj=[[[s.str_struct, int(round(float(s.energy) * 100))] for s in st] for st in traj])
    return case, recs


def main():
    import multiprocessing as mp
    seqs = sequences()
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 4
    todo = [(si, s, ms) for ms in (1, 5) for si, s in enumerate(seqs)]
    if "--ms50" in sys.argv:
        todo = [(si, seqs[si], 50) for si in (0, 1)]
    with mp.get_context("fork").Pool(jobs) as pool:
        res = pool.map(one_case, todo, chunksize=1)
    cases = [c for c, _ in res]
    recs = [r for _, rr in res for r in rr]
    suffix = "_ms50" if "--ms50" in sys.argv else ""
    with gzip.GzipFile(os.path.join(GOLD, f"fold_traj_long{suffix}.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(dict(sequences=seqs, cases=cases), separators=(",", ":")).encode())
    with gzip.GzipFile(os.path.join(GOLD, f"node_expand_long{suffix}.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(dict(sequences=seqs, records=recs), separators=(",", ":")).encode())
    print(len(cases), "long fold cases;", len(recs), "long region records")


if __name__ == "__main__":
    main()
