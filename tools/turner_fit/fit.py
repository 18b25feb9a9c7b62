"""L1-minimal correction of the prior tables so all KAT triples are exact."""
import sys, json, time, collections
import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog
from . import kats, model


def build(ks):
    keys = {}
    rows, cols, vals, r = [], [], [], []
    for n, (s, st, d) in enumerate(ks):
        f = model.features(s, st)
        e0 = f.const
        for k, c in f.c.items():
            if not c:
                continue
            e0 += c * model.prior_value(k)
            j = keys.setdefault(k, len(keys))
            rows.append(n); cols.append(j); vals.append(c)
        r.append(d - e0)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(len(ks), len(keys)), dtype=float)
    return A, np.array(r, float), list(keys)


def weight(key):
    # confidence in the prior: higher weight = more reluctant to change
    k = key[0]
    if k in ("stack", "hp", "bulge", "int", "termAU", "MLclosing", "MLintern", "MLbase"):
        return 20.0
    if k in ("tri", "tetra", "hexa"):
        return 5.0
    if k in ("int11", "int21", "int22"):
        return 1.0
    return 2.0


def fit(ks):
    A, r, keys = build(ks)
    m, n = A.shape
    w = np.array([weight(k) for k in keys])
    c = np.concatenate([w, w])
    Aeq = sp.hstack([A, -A]).tocsc()
    t = time.time()
    res = linprog(c, A_eq=Aeq, b_eq=r, bounds=(0, None), method="highs")
    print("lp", res.status, res.message, time.time() - t, file=sys.stderr)
    d = res.x[:n] - res.x[n:]
    return keys, d, A, r


if __name__ == "__main__":
    ks = kats.load_fixture()
    keys, d, A, r = fit(ks)
    nz = [(k, v) for k, v in zip(keys, d) if abs(v) > 1e-6]
    print(len(keys), "keys;", len(nz), "corrected")
    frac = [(k, v) for k, v in nz if abs(v - round(v)) > 1e-6]
    print("non-integer:", len(frac))
    by = collections.Counter(k[0] for k, _ in nz)
    print(by)
    tot = collections.Counter(k[0] for k in keys)
    print(tot)
    cnt = np.asarray((A != 0).sum(axis=0)).ravel()
    out = {"corr": [(list(k), float(v), int(cnt[keys.index(k)])) for k, v in nz]}
    json.dump(out, open("/tmp/fit_corr.json", "w"))
    for k, v in sorted(nz, key=lambda kv: (kv[0][0], kv[0][1:])):
        if k[0] not in ("int11", "int21", "int22"):
            print(k, model.prior_value(k), "->", model.prior_value(k) + v, "n=", cnt[keys.index(k)])


def fitted_theta(ks=None):
    ks = ks or kats.load_fixture()
    keys, d, A, r = fit(ks)
    cnt = np.asarray((A != 0).sum(axis=0)).ravel()
    th = {k: int(round(model.prior_value(k) + v)) for k, v in zip(keys, d)}
    return th, {k: int(c) for k, c in zip(keys, cnt)}
