"""Hold-out validation of the fitted Turner tables.

The built-in tables are a recalled prior corrected until all 11 505 reference-held (sequence, structure, energy)
rows are exact, so those rows cannot say how good the entries are that no row exercises.  This script measures
it: k-fold split BY SEQUENCE (all structures of a sequence go to the same fold), the two-pass fit of
make_tables.fit_tables() on k-1 folds, exact-dcal rate on the held-out fold - overall and split by whether the
held-out structure touches a table entry that no training row exercised.

    python -m tools.turner_fit.holdout [k]      -> profiles/r04_turner_holdout.json
"""
import json, os, sys, time
import numpy as np
from . import kats, model, make_tables

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "profiles", "r04_turner_holdout.json")


def run(k=5, seed=0, folds=None):
    ks = kats.load_fixture()
    seqs = sorted({s for s, _, _ in ks})
    rng = np.random.default_rng(seed)
    fold_of = {s: int(f) for s, f in zip(seqs, rng.integers(0, k, size=len(seqs)))}
    res = []
    for f in (range(k) if folds is None else folds):
        t0 = time.time()
        train = [x for x in ks if fold_of[x[0]] != f]
        test = [x for x in ks if fold_of[x[0]] == f]
        th, cnt = make_tables.fit_tables(train)
        seen_keys = set(th)
        n_seen = n_seen_ok = n_unseen = n_unseen_ok = 0
        abs_err = []
        for s, st, d in test:
            feats = model.features(s, st)
            unseen = any(c and key not in seen_keys for key, c in feats.c.items())
            e = model.energy(s, st, th)
            if unseen:
                n_unseen += 1; n_unseen_ok += int(e == d)
            else:
                n_seen += 1; n_seen_ok += int(e == d)
            if e != d:
                abs_err.append(abs(e - d))
        r = dict(fold=f, train=len(train), test=len(test), test_all_entries_seen=n_seen, exact_when_all_seen=n_seen_ok,
                 test_touching_unseen_entry=n_unseen, exact_when_touching_unseen=n_unseen_ok,
                 mean_abs_err_dcal_of_wrong=float(np.mean(abs_err)) if abs_err else 0.0,
                 max_abs_err_dcal=int(max(abs_err)) if abs_err else 0, seconds=round(time.time() - t0, 1))
        print(r, file=sys.stderr, flush=True)
        res.append(r)
    tot = {k_: sum(r[k_] for r in res) for k_ in ("test", "test_all_entries_seen", "exact_when_all_seen",
                                                     "test_touching_unseen_entry", "exact_when_touching_unseen")}
    tot["wrong_rate"] = 1.0 - (tot["exact_when_all_seen"] + tot["exact_when_touching_unseen"]) / max(1, tot["test"])
    tot["wrong_rate_when_touching_unseen"] = 1.0 - tot["exact_when_touching_unseen"] / max(1, tot["test_touching_unseen_entry"])
    tot["wrong_rate_when_all_seen"] = 1.0 - tot["exact_when_all_seen"] / max(1, tot["test_all_entries_seen"])
    return dict(k=k, seed=seed, split="by sequence", folds=res, total=tot)


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    out = run(k)
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps(out["total"], indent=1))
