"""Two-pass fit -> full ViennaRNA-layout tables -> params/turner2004_fitted.json

pass 1: L1 fit with rule priors.  From its int22 values an additive
pair-frame mismatch model is least-squares fitted and, rounded to 10 dcal,
becomes the prior of the *unseen* 2x2 entries; the GC-CG measured 1x2 block is
propagated to the GU-CG / GC-UG / GU-UG blocks (pattern visible in the fit).
pass 2: L1 fit again with the improved priors.  Every entry records whether a
KAT triple exercises it (`pinned`) or whether it is a rule/prior value.
"""
import json, sys, os, itertools
import numpy as np
from . import kats, model, fit, prior as P

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "params", "turner2004_fitted.json")


def additive_int22(th, cnt):
    idx = {}
    def fi(t, x, y):
        return idx.setdefault((t, x, y), len(idx))
    items = [(k, v) for k, v in th.items() if k[0] == "int22"]
    rows = [(fi(k[1], k[3], k[6]), fi(k[2], k[5], k[4])) for k, _ in items]
    A = np.zeros((len(rows), len(idx) + 1))
    for i, (p, q) in enumerate(rows):
        A[i, p] += 1; A[i, q] += 1; A[i, -1] = 1
    y = np.array([v for _, v in items], float)
    W = np.sqrt(np.array([min(cnt[k], 5) for k, _ in items], float))
    sol, *_ = np.linalg.lstsq(A * W[:, None], y * W, rcond=None)
    mean_f = {}
    for t in range(1, 7):
        vals = [sol[j] for (tt, x, yy), j in idx.items() if tt == t]
        mean_f[t] = float(np.mean(vals)) if vals else 0.0
    def pred(t1, t2, a, b, c, d):
        f1 = sol[idx[(t1, a, d)]] if (t1, a, d) in idx else mean_f[t1]
        f2 = sol[idx[(t2, c, b)]] if (t2, c, b) in idx else mean_f[t2]
        return int(round((f1 + f2 + sol[-1]) / 10.0)) * 10
    return pred


def halfunit_int22(th, cnt, rule="up", lam=0.3, sweeps=8):
    """Round 4: the prior of the UNSEEN 2x2 entries as the published table was built (Xia/Mathews/Turner: a 2x2 loop is the
    mean of the two symmetric tandem-mismatch loops its halves belong to, plus a term for the combination of the two
    mismatches): value = round10(H[t1][a][d] + H[t2][c][b] + D[{mismatch 1, mismatch 2}]) with H in units of 5 dcal (half of a
    value given to 0.1 kcal/mol) and halves rounded up.  Least squares on the pinned entries, snapped to the 5-dcal grid, then
    coordinate descent on the number of pinned entries reproduced exactly.  Against the additive model of round 2 (no D, no
    half units): 5-fold hold-out over the pinned ENTRIES (tools/turner_fit/int22_models.py -> profiles/r05_int22_models.json,
    round 5: the script round 4 quoted was not kept; this is the comparison again, by-entry folds, seed 0) additive 19.0 % exact
    (mean error 26 dcal), half units rounded up 65.9 % (9 dcal), down 69.2 %, away from zero 61.3 %, to even 56.1 % - round 4 had
    18.6 / 68.5 / 68.1 / 63.4 / 55.9 from its own split: up and down are within the spread of the split, the tables keep `up`."""
    import math

    def params_of(k):
        t1, t2, a, b, c, d = k
        return (("F", t1, a, d), ("F", t2, c, b), ("D",) + tuple(sorted(((a - 1) * 4 + d - 1, (c - 1) * 4 + b - 1))))

    def rnd(x):
        if x % 2 == 0:
            return x * 5
        if rule == "up":
            return (x + 1) * 5
        if rule == "down":
            return (x - 1) * 5
        if rule == "away":
            return (x + 1) * 5 if x > 0 else (x - 1) * 5
        lo = (x - 1) // 2                      # "even"
        return (lo if lo % 2 == 0 else lo + 1) * 10

    items = [(k[1:], v) for k, v in th.items() if k[0] == "int22"]
    w = {k[1:]: (2 if cnt[k] >= 2 else 1) for k in th if k[0] == "int22"}
    idx = {}
    for k, _ in items:
        for q in params_of(k):
            idx.setdefault(q, len(idx))
    n = len(idx)
    A = np.zeros((len(items) + n, n + 1)); y = np.zeros(len(items) + n)
    for i, (k, v) in enumerate(items):
        for q in params_of(k):
            A[i, idx[q]] += 1
        A[i, -1] = 1; y[i] = v
    for q, j in idx.items():
        A[len(items) + j, j] = math.sqrt(lam if q[0] == "D" else 0.01)
    sol, *_ = np.linalg.lstsq(A, y, rcond=None)
    h = {q: int(round((sol[j] + (sol[-1] / 2 if q[0] == "F" else 0.0)) / 5.0)) for q, j in idx.items()}
    byq = {}
    for k, v in items:
        for q in params_of(k):
            byq.setdefault(q, []).append((k, v))

    def score(lst):
        return sum(w[k] for k, v in lst if rnd(sum(h[q] for q in params_of(k))) == v)
    for _ in range(sweeps):
        changed = 0
        for q in list(h):
            lst = byq[q]; v0 = h[q]; best, bestv = score(lst), v0
            for dv in (-4, -3, -2, -1, 1, 2, 3, 4):
                if q[0] == "D" and dv % 2:
                    continue
                h[q] = v0 + dv
                sc = score(lst)
                if sc > best:
                    best, bestv = sc, v0 + dv
            h[q] = bestv
            changed += bestv != v0
        if not changed:
            break
    mean_f = {}
    for t in range(1, 7):
        vals = [v for q, v in h.items() if q[0] == "F" and q[1] == t]
        mean_f[t] = int(round(float(np.mean(vals)))) if vals else 0

    def pred(t1, t2, a, b, c, d):
        tot = 0
        for q in params_of((t1, t2, a, b, c, d)):
            if q in h:
                tot += h[q]
            elif q[0] == "F":
                tot += mean_f[q[1]]
        return rnd(tot)
    return pred


EXTRA_PASSES = 2
INT22_MODEL = "halfunit"          # "additive": round 2's prior of the unseen 2x2 entries (tools/turner_fit/holdout.py arbitrates)


def _pur(t, x, y):
    if (x, y) in ((3, 1), (3, 3)):
        return True
    return (x, y) == (1, 3) and t in (1, 4, 6)


def int21_rule(t1, t2, a, b, c):
    n = (t1 > 2) + (t2 > 2)
    base = 230 + 70 * n
    if _pur(t1, a, c) or _pur(t2, b, a):
        return base - (120 if n == 0 else 110)
    if a == 4 and (b == 4 or c == 4):
        return base - (80 if n < 2 else 70)
    return base


def fit_tables(ks):
    """the two-pass fit on the triples `ks` -> (theta, exercise count per key); leaves the priors of the unseen
    int21/int22 entries installed in `prior` (P.int21_prior / P.int22_prior), so model.energy(.., theta) evaluates
    any structure with exactly the tables main() would write"""
    P.int21_prior = int21_rule
    th1, cnt1 = fit.fitted_theta(ks)
    pred22 = halfunit_int22(th1, cnt1) if INT22_MODEL == "halfunit" else additive_int22(th1, cnt1)
    P.int22_prior = pred22

    # 2x1 loops closed by G.U pairs (blocks GU-CG, GC-UG, GU-UG): most entries follow the rule; a handful of loops (A/AA, A/GA, ...)
    # carry the value of the GC-CG block whatever the closing pairs.  Which ones is read off the exercised entries: those whose
    # value equals the GC-CG block's and differs from the rule.  (Round 2 propagated EVERY GC-CG entry: of the held-out structures
    # whose only unseen entry sat in these blocks 17 of 28 then came out 0.7 or 1.4 kcal/mol too low.)
    gu_blocks = ((3, 1), (2, 4), (3, 4))
    copied = set()
    for k, v in th1.items():
        if k[0] == "int21" and (k[1], k[2]) in gu_blocks and cnt1[k] >= 2:
            k21 = ("int21", 2, 1) + k[3:]
            if k21 in th1 and th1[k21] == v and v != int21_rule(k[1], k[2], *k[3:]):
                copied.add(k[3:])

    def int21_p2(t1, t2, a, b, c):
        if (t1, t2) in gu_blocks and (a, b, c) in copied:
            k = ("int21", 2, 1, a, b, c)
            if k in th1 and cnt1[k] >= 2:
                return th1[k]
        return int21_rule(t1, t2, a, b, c)
    P.int21_prior = int21_p2
    th, cnt = fit.fitted_theta(ks)
    for _ in range(EXTRA_PASSES):          # the 2x2 model again on the values of the better fit, and the fit again with its priors
        P.int22_prior = halfunit_int22(th, cnt) if INT22_MODEL == "halfunit" else additive_int22(th, cnt)
        th, cnt = fit.fitted_theta(ks)
    return th, cnt


def main():
    ks = kats.load_fixture()
    th, cnt = fit_tables(ks)
    # verify
    bad = sum(1 for s, st, d in ks if model.energy(s, st, th) != d)
    print("KAT mismatches after fit:", bad, "of", len(ks), file=sys.stderr)
    assert bad == 0

    def val(key):
        return th[key] if key in th else model.prior_value(key)

    out = {"pinned": sorted(["|".join(map(str, k)) for k in th]), "tables": {}}
    T = out["tables"]
    T["stack"] = [[val(model.canon_stack(a, b)) for b in range(1, 7)] for a in range(1, 7)]
    T["hairpin"] = [P.INF if P.HAIRPIN[i] >= P.INF else val(("hp", i)) for i in range(31)]
    T["bulge"] = [P.INF if P.BULGE[i] >= P.INF else val(("bulge", i)) for i in range(31)]
    T["interior"] = [P.INF if P.INTERIOR[i] >= P.INF else val(("int", i)) for i in range(31)]
    for name, key in (("mismatch_hairpin", "mmH"), ("mismatch_interior", "mmI"),
                      ("mismatch_interior_1n", "mm1n"), ("mismatch_interior_23", "mm23"),
                      ("mismatch_multi", "mmM"), ("mismatch_exterior", "mmE")):
        T[name] = [[[val((key, t, a, b)) for b in range(1, 5)] for a in range(1, 5)] for t in range(1, 7)]
    T["dangle5"] = [[val(("d5", t, a)) for a in range(1, 5)] for t in range(1, 7)]
    T["dangle3"] = [[val(("d3", t, a)) for a in range(1, 5)] for t in range(1, 7)]
    T["int11"] = [[[[val(model.canon_int11(t1, t2, a, b)) for b in range(1, 5)] for a in range(1, 5)]
                   for t2 in range(1, 7)] for t1 in range(1, 7)]
    T["int21"] = [[[[[val(("int21", t1, t2, a, b, c)) for c in range(1, 5)] for b in range(1, 5)]
                    for a in range(1, 5)] for t2 in range(1, 7)] for t1 in range(1, 7)]
    T["int22"] = [[[[[[val(model.canon_int22(t1, t2, a, b, c, d)) for d in range(1, 5)] for c in range(1, 5)]
                     for b in range(1, 5)] for a in range(1, 5)] for t2 in range(1, 7)] for t1 in range(1, 7)]
    T["ml_base"] = val(("MLbase",)); T["ml_closing"] = val(("MLclosing",)); T["ml_intern"] = val(("MLintern",))
    T["ninio"] = P.NINIO; T["max_ninio"] = P.MAX_NINIO; T["terminal_au"] = val(("termAU",)); T["lxc"] = P.LXC
    T["triloops"] = {k: val(("tri", k)) for k in P.TRILOOPS}
    T["tetraloops"] = {k: val(("tetra", k)) for k in P.TETRALOOPS}
    T["hexaloops"] = {k: val(("hexa", k)) for k in P.HEXALOOPS}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(out, open(OUT, "w"), indent=0, separators=(",", ":"))
    # stats
    import collections
    tot = {"int11": 6 * 6 * 16, "int21": 6 * 6 * 64, "int22": 6 * 6 * 256}
    seen = collections.Counter(k[0] for k in th)
    print({k: (seen[k], tot.get(k)) for k in seen}, file=sys.stderr)


if __name__ == "__main__":
    main()
