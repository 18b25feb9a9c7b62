"""Which model predicts the UNSEEN 2x2 interior-loop entries best?  (round 5: the comparison behind make_tables.halfunit_int22,
which round 4 quoted without the script)

The entries of the 2x2 table that some reference-held energy row exercises are split in k folds BY ENTRY; every model is fitted
to the entries of k-1 folds and asked for the entries of the held-out fold - exact-value rate and mean absolute error:
  additive            round 2's prior: value = F[t1][a][d] + F[t2][c][b] + const, rounded to 10 dcal (make_tables.additive_int22)
  halfunit up/down/   round 4's prior: value = round10(H[t1][a][d] + H[t2][c][b] + D[{mismatch 1, mismatch 2}]), H in units of 5 dcal,
   away/even          halves rounded up / down / away from zero / to even (make_tables.halfunit_int22)
The entry values come from the first pass of the fit on ALL rows (what make_tables.fit_tables hands to the model).

    python -m tools.turner_fit.int22_models [k]      -> profiles/r05_int22_models.json
"""
import json, os, sys
import numpy as np
from . import kats, fit, prior as P, make_tables

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "profiles", "r05_int22_models.json")


def run(k=5, seed=0):
    ks = kats.load_fixture()
    P.int21_prior = make_tables.int21_rule
    th, cnt = fit.fitted_theta(ks)
    keys = sorted(key for key in th if key[0] == "int22")
    rng = np.random.default_rng(seed)
    fold = {key: int(f) for key, f in zip(keys, rng.integers(0, k, size=len(keys)))}
    models = {"additive": lambda t, c: make_tables.additive_int22(t, c)}
    for rule in ("up", "down", "away", "even"):
        models["halfunit " + rule] = (lambda r: (lambda t, c: make_tables.halfunit_int22(t, c, rule=r)))(rule)
    out = {"k": k, "seed": seed, "entries": len(keys), "split": "by entry", "models": {}}
    for name, make in models.items():
        n = ok = 0
        err = []
        for f in range(k):
            tr = {key: v for key, v in th.items() if key[0] != "int22" or fold[key] != f}
            pred = make(tr, cnt)
            for key in keys:
                if fold[key] == f:
                    p = pred(*key[1:])
                    n += 1; ok += int(p == th[key]); err.append(abs(p - th[key]))
        out["models"][name] = {"held_out_entries": n, "exact": ok, "exact_rate": round(ok / max(1, n), 4), "mean_abs_err_dcal": round(float(np.mean(err)), 2)}
        print(name, out["models"][name], file=sys.stderr, flush=True)
    return out


if __name__ == "__main__":
    res = run(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    json.dump(res, open(OUT, "w"), indent=1)
    print(json.dumps(res["models"], indent=1))
