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


def fold(sequence, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False,
         temp=37.0, gc_wei=3.0, au_wei=2.0, gu_wei=1.0, counters=None):
    if temp != 37.0:
        raise NotImplementedError("oracle: 37 C only")
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
            for i, nm in enumerate(("node_expansions", "lag_scans", "evals", "children")):
                counters[nm] = counters.get(nm, 0) + L.oracle_result_counter(r, i)
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
