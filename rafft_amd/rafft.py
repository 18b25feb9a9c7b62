"""rafft.fold() on the MI355X: same signature, return values and error behaviour as
the reference's rafft/rafft.py:219-239, computed by the HIP kernels of
libraffthip.so through the C-ABI of include/rafft_hip.h.

`fold_batch` is the batched entry the reference lacks (it spawns one CLI process
per sequence, benchmark_results/bench_fft.py:8-22)."""
import ctypes as C
from collections.abc import Sequence

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
        raise ValueError("sequence longer than 32768 nt (RAFFT_MAX_LEN) is not supported")
    raise N.RafftError(status, "per-sequence failure")


class _Owner:
    """keeps a rafft_result alive while any view of it exists; frees it with the last one"""
    __slots__ = ("res", "lib")

    def __init__(self, lib, res):
        self.lib, self.res = lib, res

    def __del__(self):
        try:
            if self.res is not None:
                self.lib.rafft_free_result(self.res)
                self.res = None
        except Exception:
            pass


class Beam(Sequence):
    """The structures of one beam, rows [lo, hi) of one sequence's result, as the reference's list[Structure]
    (rafft/rafft.py:236-239) - materialised on first touch from one bytes buffer and one int32 array."""
    __slots__ = ("_owner", "_sr", "_lo", "_hi", "_rows")

    def __init__(self, owner, sr, lo, hi):
        self._owner, self._sr, self._lo, self._hi, self._rows = owner, sr, lo, hi, None

    def _build(self):
        if self._rows is None:
            sr, lo, n = self._sr, self._lo, self._hi - self._lo
            w, Ln = sr.length + 1, sr.length
            raw = C.string_at(C.addressof(sr.db.contents) + lo * w, n * w).decode("ascii") if n else ""
            dcal = np.ctypeslib.as_array(sr.dcal, shape=(sr.n_structs,))[lo:lo + n] if n else np.zeros(0, np.int32)
            en = energies_from_dcal(dcal).tolist()
            dl = dcal.tolist()
            self._rows = [Structure(raw[k * w:k * w + Ln], dl[k], en[k]) for k in range(n)]
        return self._rows

    def __len__(self):
        return self._hi - self._lo

    def __getitem__(self, k):
        return self._build()[k]

    def __iter__(self):
        return iter(self._build())

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(self._build())

    def __reduce__(self):          # travels between ranks (sharding.fold_sharded) as a plain list
        return (list, (self._build(),))

    def dcal(self):
        """exact integer energies (dcal/mol) of the beam as an int32 array - no Structure objects are built"""
        n = self._hi - self._lo
        return np.ctypeslib.as_array(self._sr.dcal, shape=(self._sr.n_structs,))[self._lo:self._hi].copy() if n else np.zeros(0, np.int32)

    def dot_brackets(self):
        """the beam's dot-bracket rows as one bytes buffer of (L+1)-byte NUL-terminated rows"""
        w = self._sr.length + 1
        return C.string_at(C.addressof(self._sr.db.contents) + self._lo * w, (self._hi - self._lo) * w) if self._hi > self._lo else b""


class BatchResult(Sequence):
    """What fold_batch returns: one entry per input sequence, `structures` or `(structures, trajectory)` exactly as
    fold() gives them, built lazily from the result rows the library left in pinned host memory (indexing a sequence
    costs one small object; Structure objects appear only for the beams that are actually looked at).
    Failed sequences (raise_errors=False) are None."""
    __slots__ = ("_owner", "_res", "_n", "_traj", "_cache")

    def __init__(self, owner, n, traj):
        self._owner, self._res, self._n, self._traj, self._cache = owner, owner.res.contents, n, traj, {}

    def __len__(self):
        return self._n

    def _one(self, i):
        r = self._cache.get(i)
        if r is None:
            sr = self._res.seq[i]
            if sr.status != N.OK:
                return None
            if self._traj:
                steps = [Beam(self._owner, sr, sr.step_off[s], sr.step_off[s] + sr.step_size[s]) for s in range(sr.n_steps)]
                r = (steps[-1], steps)
            else:
                last = sr.n_steps - 1
                r = Beam(self._owner, sr, sr.step_off[last], sr.step_off[last] + sr.step_size[last])
            self._cache[i] = r
        return r

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(k) for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._one(i)

    def status(self, i):
        return self._res.seq[i].status

    def raw(self, i):
        """Sequence i as flat buffers, without building any Structure: (L, step sizes, dot-bracket rows as an
        (n_structs, L) uint8 array, int32 dcal array) - views into the library's pinned result chunk, valid while this
        BatchResult lives.  Steps are laid out one after the other (one step without traj)."""
        sr = self._res.seq[i]
        if sr.status != N.OK:
            return None
        n, L = sr.n_structs, sr.length
        sizes = [sr.step_size[s] for s in range(sr.n_steps)]
        if n == 0:
            return L, sizes, np.zeros((0, L), np.uint8), np.zeros(0, np.int32)
        buf = (C.c_char * (n * (L + 1))).from_address(C.addressof(sr.db.contents))
        rows = np.frombuffer(buf, dtype=np.uint8).reshape(n, L + 1)[:, :L]
        return L, sizes, rows, np.ctypeslib.as_array(sr.dcal, shape=(n,))


class PendingBatch:
    """A batch in flight (submit_batch): `.result()` waits for it and returns what fold_batch returns."""
    __slots__ = ("_lib", "_job", "_seqs", "_traj", "_raise", "_res")

    def __init__(self, lib, job, seqs, traj, raise_errors):
        self._lib, self._job, self._seqs, self._traj, self._raise, self._res = lib, job, seqs, traj, raise_errors, None

    def result(self):
        if self._res is None:
            res = C.POINTER(N.Result)()
            job, self._job = self._job, None
            N.check(self._lib.rafft_fold_wait(job, C.byref(res)))
            owner = _Owner(self._lib, res)
            n = len(self._seqs)
            if self._raise and res.contents.n_failed:
                for i in range(n):
                    if res.contents.seq[i].status != N.OK:
                        _raise_like_reference(res.contents.seq[i].status, self._seqs[i])
            self._res = BatchResult(owner, n, self._traj)
        return self._res

    def __del__(self):
        try:
            if self._job is not None:          # never waited for: wait now, so that the library can release the batch
                res = C.POINTER(N.Result)()
                if self._lib.rafft_fold_wait(self._job, C.byref(res)) == 0:
                    self._lib.rafft_free_result(res)
        except Exception:
            pass


def submit_batch(sequences, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False,
                 temp=37.0, gc_wei=3.0, au_wei=2.0, gu_wei=1.0, device=-1, raise_errors=True):
    """Queue a batch and return at once (continuous batching: the nearly empty last folding steps of one batch run
    beside the busy first steps of the next).  `.result()` of the returned PendingBatch gives fold_batch's result."""
    L = N.lib()
    _params_mod.ensure_default_params()
    p = _params(nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei)
    n = len(sequences)
    enc = [s.encode("ascii", "replace") for s in sequences]
    arr = (C.c_char_p * n)(*enc)
    lens = (C.c_int * n)(*map(len, enc))
    job = C.c_void_p()
    N.check(L.rafft_fold_submit(C.byref(p), n, arr, lens, device, C.byref(job)))
    return PendingBatch(L, job, sequences, bool(traj), raise_errors)


def fold_batch(sequences, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False,
               temp=37.0, gc_wei=3.0, au_wei=2.0, gu_wei=1.0, device=-1, raise_errors=True):
    """Fold many sequences in one GPU batch.  Returns a BatchResult: one entry per input sequence, `structures` or
    `(structures, trajectory)` exactly as fold() does, materialised on access."""
    return submit_batch(sequences, nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei,
                        device, raise_errors).result()


def fold(sequence, nb_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False, temp=37.0,
         gc_wei=3.0, au_wei=2.0, gu_wei=1.0):
    "fold a given sequence (rafft/rafft.py:219-239)"
    r = fold_batch([sequence], nb_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc_wei, au_wei, gu_wei)[0]
    return (list(r[0]), [list(st) for st in r[1]]) if traj else list(r)        # plain lists, as the reference returns


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


def eval_structures_info(seqs, dbs):
    """eval_structures at 37 C plus, per structure, whether its energy reads an entry of the BUILT-IN interior-loop tables that no
    reference-held energy row exercises (a rule / model value - DESIGN.md 2.1): (dcal, status, guessed).  All zeros with a
    loaded ViennaRNA parameter file, whose every entry is ViennaRNA's."""
    L = N.lib()
    _params_mod.ensure_default_params()
    n = len(seqs)
    a = (C.c_char_p * n)(*[s.encode() for s in seqs])
    b = (C.c_char_p * n)(*[s.encode() for s in dbs])
    out, st, gs = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    N.check(L.rafft_eval_structures_info(n, a, b, out, st, gs))
    return list(out), list(st), list(gs)


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
