"""ctypes front-end of oracle/liboracle.so (rafft_oracle.c) - test infrastructure.

Mirrors the reference's `rafft.fold` signature (rafft/rafft.py:219-221) so parity
tests read like calls into the reference.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rafft_oracle.c")
    tab = os.path.join(_HERE, "..", "params", "turner2004_tables.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(tab)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_fold.restype = C.c_void_p
        L.oracle_fold.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
        L.oracle_result_n_steps.argtypes = [C.c_void_p]
        L.oracle_result_step_size.argtypes = [C.c_void_p, C.c_int]
        L.oracle_result_struct.restype = C.c_char_p
        L.oracle_result_struct.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_result_dcal.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_result_energy.restype = C.c_double
        L.oracle_result_energy.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_result_counter.restype = C.c_long
        L.oracle_result_counter.argtypes = [C.c_void_p, C.c_int]
        L.oracle_result_free.argtypes = [C.c_void_p]
        L.oracle_eval_structure.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.oracle_result_unpinned.restype = C.c_long
        L.oracle_result_unpinned.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_set_table.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int]
        L.oracle_set_scalars.argtypes = [C.c_int] * 6 + [C.c_double]
        L.oracle_set_special.argtypes = [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
        L.oracle_set_pinned.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_set_pinned_scalars.argtypes = [C.c_int] * 4
        L.oracle_set_pinned_special.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.oracle_special_seq.restype = C.c_char_p
        L.oracle_special_seq.argtypes = [C.c_int, C.c_int]
        L.oracle_track_enable.argtypes = [C.c_int]
        L.oracle_eval_structure_tracked.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_long)]
        _LIB = L
    return _LIB


class Structure:
    """Same read surface as the reference's Structure (rafft/utils.py:32-39)."""
    __slots__ = ("str_struct", "energy", "dcal")

    def __init__(self, s, e, d):
        self.str_struct, self.energy, self.dcal = s, e, d

    def __repr__(self):
        return f"{self.str_struct} {self.energy:6.1f}"


def eval_structure(seq, db):
    out = C.c_int()
    rc = lib().oracle_eval_structure(seq.encode(), db.encode(), C.byref(out))
    if rc:
        raise ValueError(f"oracle_eval_structure rc={rc}")
    return out.value


TABLE_NAMES = ("stack", "hairpin", "bulge", "interior", "mismatch_hairpin", "mismatch_interior", "mismatch_interior_1n",
               "mismatch_interior_23", "mismatch_multi", "mismatch_exterior", "dangle5", "dangle3", "int11", "int21", "int22")


def set_tables(T):
    """Install energy tables (the dict tests/_par_reader.tables_at returns: arrays in this oracle's layout, `scalars`,
    `special`).  The temperature lives in the tables: the oracle itself knows no rescaling."""
    L = lib()
    for name in TABLE_NAMES:
        a = np.ascontiguousarray(T[name], dtype=np.int32).reshape(-1)
        rc = L.oracle_set_table(name.encode(), a.ctypes.data_as(C.POINTER(C.c_int)), a.size)
        assert rc == 0, name
    sc = T["scalars"]
    L.oracle_set_scalars(sc["ml_base"], sc["ml_closing"], sc["ml_intern"], sc["ninio"], sc["max_ninio"], sc["term_au"], sc["lxc"])
    for kind, ent in T["special"].items():
        e = (C.c_int * max(1, len(ent)))(*[x[1] for x in ent])
        assert L.oracle_set_special(kind, len(ent), "".join(x[0] for x in ent).encode(), e) == 0


def reset_tables():
    lib().oracle_reset_tables()


def set_pinned(fitted_json):
    """Mark which entries of the built-in tables a reference-held energy row exercises (the `pinned` list of
    params/turner2004_fitted.json, keys as tools/turner_fit/model.py canonicalises them); everything else is a
    rule/prior value.  Entries with an N base keep their default (pinned)."""
    import json
    keys = set(tuple(k.split("|")) for k in json.load(open(fitted_json))["pinned"])
    L = lib()
    has = lambda *k: tuple(str(x) for x in k) in keys

    def put(name, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
        assert L.oracle_set_pinned(name.encode(), a.tobytes(), a.size) == 0, name
    st = np.ones((7, 7), np.uint8)
    for a in range(1, 7):
        for b in range(1, 7):
            st[a, b] = has("stack", *min((a, b), (b, a)))
    put("stack", st)
    for name, key in (("hairpin", "hp"), ("bulge", "bulge"), ("interior", "int")):
        v = np.ones(31, np.uint8)
        for i in range(31):
            v[i] = has(key, i)
        put(name, v)
    for name, key in (("mismatch_hairpin", "mmH"), ("mismatch_interior", "mmI"), ("mismatch_interior_1n", "mm1n"),
                      ("mismatch_interior_23", "mm23"), ("mismatch_multi", "mmM"), ("mismatch_exterior", "mmE")):
        v = np.ones((7, 5, 5), np.uint8)
        for t in range(1, 7):
            for a in range(1, 5):
                for b in range(1, 5):
                    v[t, a, b] = has(key, t, a, b)
        put(name, v)
    for name, key in (("dangle5", "d5"), ("dangle3", "d3")):
        v = np.ones((7, 5), np.uint8)
        for t in range(1, 7):
            for a in range(1, 5):
                v[t, a] = has(key, t, a)
        put(name, v)
    v = np.ones((7, 7, 5, 5), np.uint8)
    for t in range(1, 7):
        for u in range(1, 7):
            for a in range(1, 5):
                for b in range(1, 5):
                    v[t, u, a, b] = has("int11", *min((t, u, a, b), (u, t, b, a)))
    put("int11", v)
    v = np.ones((7, 7, 5, 5, 5), np.uint8)
    for t in range(1, 7):
        for u in range(1, 7):
            for a in range(1, 5):
                for b in range(1, 5):
                    for c in range(1, 5):
                        v[t, u, a, b, c] = has("int21", t, u, a, b, c)
    put("int21", v)
    v = np.ones((7, 7, 5, 5, 5, 5), np.uint8)
    for t in range(1, 7):
        for u in range(1, 7):
            for a in range(1, 5):
                for b in range(1, 5):
                    for c in range(1, 5):
                        for d in range(1, 5):
                            v[t, u, a, b, c, d] = has("int22", *min((t, u, a, b, c, d), (u, t, c, d, a, b)))
    put("int22", v)
    L.oracle_set_pinned_scalars(has("MLbase"), has("MLclosing"), has("MLintern"), has("termAU"))
    for kind, key in enumerate(("tri", "tetra", "hexa")):
        flags = []
        k = 0
        while True:
            sq = L.oracle_special_seq(kind, k)
            if sq is None:
                break
            flags.append(1 if has(key, sq.decode()) else 0)
            k += 1
        assert L.oracle_set_pinned_special(kind, bytes(flags), len(flags)) == 0


def track(on=True):
    lib().oracle_track_enable(1 if on else 0)


def eval_structure_tracked(seq, db):
    """-> (dcal, number of look-ups of unpinned table entries behind it)"""
    out, n = C.c_int(), C.c_long()
    rc = lib().oracle_eval_structure_tracked(seq.encode(), db.encode(), C.byref(out), C.byref(n))
    if rc:
        raise ValueError(f"oracle_eval_structure rc={rc}")
    return out.value, n.value


def fold(sequence, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False,
         temp=37.0, gc_wei=3.0, au_wei=2.0, gu_wei=1.0, counters=None):
    """`temp` is accepted for signature parity only: the oracle evaluates with whatever tables are installed
    (set_tables; the built-in 37 C set by default) - rescaling to a temperature is the table provider's job."""
    if len(sequence) == 0:
        raise np.exceptions.AxisError("axis 1 is out of bounds for array of dimension 1")
    for ch in sequence:
        if ch not in "AGCUN":
            raise KeyError(ch)
    L = lib()
    err = C.c_int()
    r = L.oracle_fold(sequence.encode(), nb_mode, max_stack, max_branch, min_hp, min_nrj,
                      gc_wei, au_wei, gu_wei, C.byref(err))
    if not r:
        raise RuntimeError(f"oracle_fold err={err.value}")
    try:
        steps = []
        for s in range(L.oracle_result_n_steps(r)):
            steps.append([Structure(L.oracle_result_struct(r, s, k).decode(),
                                    L.oracle_result_energy(r, s, k), L.oracle_result_dcal(r, s, k))
                          for k in range(L.oracle_result_step_size(r, s))])
        if counters is not None:
            for i, nm in enumerate(("node_expansions", "lag_scans", "evals", "children", "dE_unpinned")):
                counters[nm] = counters.get(nm, 0) + L.oracle_result_counter(r, i)
            last = L.oracle_result_n_steps(r) - 1
            counters["final_unpinned"] = [L.oracle_result_unpinned(r, last, k) for k in range(L.oracle_result_step_size(r, last))]
    finally:
        L.oracle_result_free(r)
    return (steps[-1], steps) if traj else steps[-1]


def autocor(seq, pos, gc=3.0, au=2.0, gu=1.0):
    L = lib()
    n = len(pos)
    p = (C.c_int * n)(*pos)
    out = (C.c_double * (2 * n - 1))()
    rc = L.oracle_autocor(seq.encode(), p, n, C.c_double(gc), C.c_double(au), C.c_double(gu), out)
    assert rc == 0
    return np.array(out[:])


def expand_node(seq, db, pos, nb_mode=100, min_hp=3, min_nrj=0.0, gc=3.0, au=2.0, gu=1.0):
    L = lib()
    n = len(pos)
    K = max(1, min(nb_mode, 2 * n - 1))
    p = (C.c_int * n)(*pos)
    nr, nk = C.c_int(), C.c_int()
    I = lambda: (C.c_int * K)()
    D = lambda: (C.c_double * K)()
    lag, cv, nb, mi, mj, sc, dd, kept = I(), D(), I(), I(), I(), D(), I(), I()
    rc = L.oracle_expand_node(seq.encode(), db.encode(), p, n, nb_mode, min_hp, C.c_double(min_nrj),
                              C.c_double(gc), C.c_double(au), C.c_double(gu),
                              C.byref(nr), lag, cv, nb, mi, mj, sc, dd, C.byref(nk), kept)
    assert rc == 0
    r = nr.value
    return dict(lag=list(lag[:r]), cor=list(cv[:r]), nb=list(nb[:r]), mi=list(mi[:r]), mj=list(mj[:r]),
                score=list(sc[:r]), ddcal=list(dd[:r]), kept=list(kept[:nk.value]))
