"""Sequences beyond 4096 nt (up to RAFFT_MAX_LEN = 32768): regions whose FFT would not fit the LDS are correlated by
the exact direct form on multi-word bit masks with their lag values in HBM, and classes 2 and 3 read the bases of a loop
from HBM instead of an LDS copy.  Full trajectories against the oracle."""
import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import rafft as R
from _oracle_pool import fold_many

pytestmark = pytest.mark.gpu


def traj_key(traj):
    return [[(x.str_struct, x.dcal) for x in st] for st in traj]


def test_gpu_sequences_beyond_4096_nt_vs_oracle():
    rng = np.random.default_rng(4097)
    lens = [4097, 5000, 6500, 300, 90, 2500, 9000]
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    seqs.append("".join(rng.choice(list("ACGUN"), 4500, p=[.23, .23, .23, .23, .08])))
    want = fold_many([(s, 100, 3, 1000, True) for s in seqs])
    got = rafft_amd.fold_batch(seqs, 100, 3, 1000, traj=True)
    for k, (fin, traj) in enumerate(got):
        assert traj_key(traj) == want[k], (k, len(seqs[k]))
    # alone, and with another beam width / lag count
    for s, kw in ((seqs[1], dict(nb_mode=40, max_stack=2, max_branch=7)), (seqs[0], dict(nb_mode=100, max_stack=1, max_branch=100))):
        fin, traj = rafft_amd.fold(s, traj=True, **kw)
        _, o = oracle.fold(s, kw["nb_mode"], kw["max_stack"], kw["max_branch"], traj=True)
        assert traj_key(traj) == traj_key(o)
    # whole-structure energies of long structures (eval kernel) = the sums the fold carried
    flat = [(s, x.str_struct, x.dcal) for s, (fin, traj) in zip(seqs, got) for x in fin]
    e, st = R.eval_structures([f[0] for f in flat], [f[1] for f in flat])
    assert not any(st) and e == [f[2] for f in flat]


def test_gpu_maximum_length_and_beyond():
    """RAFFT_MAX_LEN itself (32 768: the oracle needs a quarter of an hour there - its fold is a committed fixture, below; here
    size-independent properties and the whole-structure re-evaluation of every final structure by the other kernel), the 16 384 that
    was the limit until round 4 (the class for the biggest regions is planned for 16 384 positions unless a sequence is longer), and
    one position more than the maximum"""
    from test_gpu_scale import check_structures
    for L, min_pairs in ((16384, 2000), (32768, 4000)):
        rng = np.random.default_rng(L)
        s = "".join(rng.choice(list("ACGU"), L))
        fin = rafft_amd.fold(s, 100, 4, 1000)
        check_structures(s, fin, 4)
        assert fin[0].dcal < -100000 and fin[0].str_struct.count("(") > min_pairs
        e, st = R.eval_structures([s] * len(fin), [x.str_struct for x in fin])
        assert not any(st) and e == [x.dcal for x in fin]
        assert [(x.str_struct, x.dcal) for x in rafft_amd.fold(s, 100, 4, 1000)] == [(x.str_struct, x.dcal) for x in fin]
    with pytest.raises(ValueError):
        rafft_amd.fold(s + "A")


@pytest.mark.parametrize("L", [17000, 32768])
def test_gpu_sequences_beyond_16384_nt_vs_committed_oracle_folds(L):
    """the oracle's fold of a random sequence of 17 000 nt (ms=3; 109 s of CPU) and of 32 768 nt (ms=1; a quarter of an hour), generated
    once by tools/make_golden_verylong.py: final beam identical, and the energies and pair counts of every beam of the trajectory"""
    import gzip, json, os
    g = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"fold_verylong_{L}.json.gz"), "rt"))
    fin, traj = rafft_amd.fold(g["sequence"], g["nb_mode"], g["max_stack"], g["max_branch"], traj=True)
    assert [[x.str_struct, x.dcal] for x in fin] == g["final"]
    assert [[x.dcal for x in st] for st in traj] == g["traj_dcal"]
    assert [[x.str_struct.count("(") for x in st] for st in traj] == g["traj_pairs"]
    # in a batch with short sequences and another very long one (one wave, the plan for 32 768 positions)
    rng = np.random.default_rng(5)
    others = ["".join(rng.choice(list("ACGU"), n)) for n in (60, 900, 20000)]
    res = rafft_amd.fold_batch([g["sequence"]] + others, g["nb_mode"], g["max_stack"], g["max_branch"])
    assert [[x.str_struct, x.dcal] for x in res[0]] == g["final"]
    for s, r in zip(others[:2], res[1:3]):
        assert [(x.str_struct, x.dcal) for x in r] == [(x.str_struct, x.dcal) for x in oracle.fold(s, g["nb_mode"], g["max_stack"], g["max_branch"])]
