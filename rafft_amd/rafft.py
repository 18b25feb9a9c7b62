"""rafft.fold() on the MI355X: same signature, return values and error behaviour as
the reference's rafft/rafft.py:219-239, computed by the HIP kernels of
libraffthip.so through the C-ABI of include/rafft_hip.h.

`fold_batch` is the batched entry the reference lacks (it spawns one CLI process
per sequence, benchmark_results/bench_fft.py:8-22)."""
import ctypes as C

import numpy as np

from . import _native as N
from . import params as _params_mod
from .utils import Structure, energies_from_dcal


def _params(nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei):
    p = N.Params()
    p.nb_mode, p.max_stack, p.max_branch, p.min_hp = int(nb_mode), int(max_stack), int(max_branch), int(min_hp)
    p.min_nrj, p.traj, p.temp = float(min_nrj), 1 if traj else 0, float(temp)
    p.gc_wei, p.au_wei, p.gu_wei = float(gc_wei), float(au_wei), float(gu_wei)
    return p


def _raise_like_reference(status, sequence):
    if status == N.ERR_BAD_CHAR:
        for ch in sequence:
            if ch not in "AGCUN":
                raise KeyError(ch)                       # prep_sequence, rafft/utils.py:73-80
    if status == N.ERR_EMPTY:
        raise np.exceptions.AxisError("axis 1 is out of bounds for array of dimension 1")  # utils.py:83
    if status == N.ERR_TOO_LONG:
        raise ValueError(f"sequence longer than {4096} nt is not supported by the LDS-resident kernels")
    raise N.RafftError(status, "per-sequence failure")


def fold_batch(sequences, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False,
               temp=37.0, gc_wei=3.0, au_wei=2.0, gu_wei=1.0, device=-1, raise_errors=True):
    """Fold many sequences in one GPU batch.  Returns one entry per input sequence:
    `structures` or `(structures, trajectory)` exactly as fold() does."""
    L = N.lib()
    _params_mod.ensure_default_params()
    p = _params(nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei)
    n = len(sequences)
    enc = [s.encode("ascii", "replace") for s in sequences]
    arr = (C.c_char_p * n)(*enc)
    lens = (C.c_int * n)(*[len(e) for e in enc])
    res = C.POINTER(N.Result)()
    N.check(L.rafft_fold_batch(C.byref(p), n, arr, lens, device, C.byref(res)))
    out = []
    try:
        for i in range(n):
            sr = res.contents.seq[i]
            if sr.status != N.OK:
                if raise_errors:
                    _raise_like_reference(sr.status, sequences[i])
                out.append(None)
                continue
            w, Ln = sr.length + 1, sr.length
            raw = C.string_at(sr.db, sr.n_structs * w).decode("ascii")
            dcal = np.ctypeslib.as_array(sr.dcal, shape=(sr.n_structs,)) if sr.n_structs else np.zeros(0, np.int32)
            en = energies_from_dcal(dcal).tolist()
            dl = dcal.tolist()
            rows = [Structure(raw[k * w:k * w + Ln], dl[k], en[k]) for k in range(sr.n_structs)]
            steps = [rows[sr.step_off[s]:sr.step_off[s] + sr.step_size[s]] for s in range(sr.n_steps)]
            out.append((steps[-1], steps) if traj else steps[-1])
    finally:
        L.rafft_free_result(res)
    return out


def fold(sequence, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False, temp=37.0,
         gc_wei=3.0, au_wei=2.0, gu_wei=1.0):
    "fold a given sequence (rafft/rafft.py:219-239)"
    return fold_batch([sequence], nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei)[0]


def eval_structures(seqs, dbs, temp=37.0):
    """GPU evaluation of eval_one_struct (rafft/utils.py:135-138) for many structures; dcal ints."""
    L = N.lib()
    _params_mod.ensure_default_params()
    n = len(seqs)
    a = (C.c_char_p * n)(*[s.encode() for s in seqs])
    b = (C.c_char_p * n)(*[s.encode() for s in dbs])
    out = (C.c_int * n)()
    st = (C.c_int * n)()
    N.check(L.rafft_eval_structures_at(float(temp), n, a, b, out, st))
    return list(out), list(st)


def expand_node(seq, db, pos, nb_mode=100, min_hp=3, min_nrj=0.0, gc=3.0, au=2.0, gu=1.0):
    """Kernel-level seam (tests): same dict as oracle.expand_node."""
    L = N.lib()
    _params_mod.ensure_default_params()
    n = len(pos)
    K = max(1, min(nb_mode, 2 * n - 1))
    p = _params(nb_mode, 1, 100, min_hp, min_nrj, False, 37.0, gc, au, gu)
    parr = (C.c_int * n)(*pos)
    nr, nk = C.c_int(), C.c_int()
    I = lambda: (C.c_int * K)()
    D = lambda: (C.c_double * K)()
    lag, cv, nb, mi, mj, sc, dd, kept = I(), D(), I(), I(), I(), D(), I(), I()
    N.check(L.rafft_expand_node(C.byref(p), seq.encode(), db.encode(), parr, n, C.byref(nr), lag, cv, nb, mi, mj, sc,
                                dd, C.byref(nk), kept))
    r = nr.value
    return dict(lag=list(lag[:r]), cor=list(cv[:r]), nb=list(nb[:r]), mi=list(mi[:r]), mj=list(mj[:r]),
                score=list(sc[:r]), ddcal=list(dd[:r]), kept=list(kept[:nk.value]))


def last_stats():
    s = N.Stats()
    N.lib().rafft_get_stats(C.byref(s))
    return s.as_dict()
