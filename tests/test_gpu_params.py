"""Loaded parameter files and temperatures on the GPU: the product path (C++ reader + ViennaRNA rescaling in
rafft_amd/csrc/rafft_params.h -> device tables -> fold kernels) against the oracle evaluating with tables that the
tests' own Python reader produced (tests/_par_reader.py) - two independent implementations of reading and rescaling.

The enthalpies of the test file are invented (no ViennaRNA parameter file exists on this box), so this pins the
MECHANISM - a loaded set replaces every table the kernels read, md.temperature rescales it the way ViennaRNA's
get_scaled_params does - not thermodynamic values; with a real rna_turner2004.par the energies are ViennaRNA's."""
import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import _native, params, rafft as R
import _par_reader as PR

pytestmark = pytest.mark.gpu


@pytest.fixture()
def synthetic(tmp_path):
    params.reset_params()
    p0 = tmp_path / "builtin.par"
    params.save_params(p0)
    par = PR.add_synthetic_enthalpies(PR.read_par(p0), seed=11)
    rng = np.random.default_rng(5)
    for k in ("stack", "int11", "int21", "int22", "mismatch_hairpin", "mismatch_interior", "mismatch_multi", "mismatch_exterior",
              "dangle5", "dangle3", "mismatch_interior_1n", "mismatch_interior_23"):
        par[k] = par[k] + 10 * rng.integers(-3, 4, size=par[k].shape)          # not the built-in values any more
    st = par["stack"][:6, :6]
    par["stack"][:6, :6] = np.minimum(st, st.T)                                  # (a stack table is symmetric in its two pairs)
    par["hairpin"] = par["hairpin"].copy(); par["hairpin"][3:] += 10 * rng.integers(-3, 4, size=28)
    par["bulge"] = par["bulge"].copy(); par["bulge"][1:] += 10 * rng.integers(-3, 4, size=30)
    par["interior"] = par["interior"].copy(); par["interior"][2:] += 10 * rng.integers(-3, 4, size=29)
    par["ml_closing"] += 30; par["ml_intern"] -= 10; par["terminal_au"] += 10; par["ninio"] += 10; par["lxc"] = 99.5
    par["Tetraloops"] = par["Tetraloops"] + [("GAAAAC", 120, 500), ("CGAAAG", 90, -300)]
    par["Triloops"] = par["Triloops"] + [("GAAAC", 300, 1000)]
    path = tmp_path / "synthetic.par"
    PR.write_par(par, path)
    yield path, par
    params.reset_params()
    oracle.reset_tables()


def traj_key(traj):
    return [[(x.str_struct, x.dcal) for x in st] for st in traj]


@pytest.mark.parametrize("temp", [37.0, 25.0, 60.0, 4.5])
def test_gpu_loaded_parameter_file_and_temperature_vs_oracle(synthetic, energy_kats, temp):
    path, par = synthetic
    params.load_params(path)
    oracle.set_tables(PR.tables_at(par, temp))
    # whole-structure energies (eval kernel) of reference-held structures, incl. long ones with big loops (lxc)
    kats = energy_kats[::23] + sorted(energy_kats, key=lambda k: -len(k[0]))[:20]
    got, st = R.eval_structures([k[0] for k in kats], [k[1] for k in kats], temp=temp)
    want = [oracle.eval_structure(k[0], k[1]) for k in kats]
    assert not any(st) and got == want
    assert temp == 37.0 or got != [k[2] for k in kats]               # really other tables, not the built-in ones
    # folds (local dE from branch prefix sums, hairpin hash table, beam order) - full trajectories
    rng = np.random.default_rng(int(temp * 10))
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in (40, 90, 150, 260, 420, 700)]
    seqs += ["GGGGAAAACCCCUUGGAAACAAGGCGAAAGCC", "GGGGAAUUAGCUCAAAUGGUAGAGCGCUCGCUUAGCAUGCGAGAGGUAGCGGGAUCGAUGCCCGCAUUCUCCACCA"]
    res = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=True, temp=temp)
    for s, (fin, traj) in zip(seqs, res):
        _, o = oracle.fold(s, 100, 20, 1000, traj=True)
        assert traj_key(traj) == traj_key(o), (len(s), temp)


def test_gpu_switching_parameter_sets_and_temperatures(synthetic):
    """the device tables follow the parameter set and the temperature of every call"""
    path, par = synthetic
    s = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"
    oracle.reset_tables()
    base = [(x.str_struct, x.dcal) for x in oracle.fold(s, 100, 10, 1000)]
    assert [(x.str_struct, x.dcal) for x in rafft_amd.fold(s, 100, 10, 1000)] == base
    params.load_params(path)
    seen = {}
    for temp in (25.0, 37.0, 25.0, 50.0):
        oracle.set_tables(PR.tables_at(par, temp))
        got = [(x.str_struct, x.dcal) for x in rafft_amd.fold(s, 100, 10, 1000, temp=temp)]
        assert got == [(x.str_struct, x.dcal) for x in oracle.fold(s, 100, 10, 1000)]
        assert seen.setdefault(temp, got) == got
    assert seen[25.0] != seen[50.0]
    params.reset_params()
    oracle.reset_tables()
    assert [(x.str_struct, x.dcal) for x in rafft_amd.fold(s, 100, 10, 1000)] == base
    with pytest.raises(_native.RafftError):
        rafft_amd.fold(s, temp=25.0)


def _oracle_tracks_the_interior_loop_tables_only():
    """the oracle marks every unpinned entry of every table; the device marks the tables an interior loop reads (1x1, 2x1, 2x2,
    the three interior mismatch tables, bulge / interior sizes): the others count as pinned for this comparison"""
    import os
    from conftest import ROOT
    oracle.set_pinned(os.path.join(ROOT, "params", "turner2004_fitted.json"))
    L = oracle.oracle.lib()
    for name, n in (("stack", 49), ("hairpin", 31), ("mismatch_hairpin", 175), ("mismatch_multi", 175), ("mismatch_exterior", 175),
                    ("dangle5", 35), ("dangle3", 35)):
        assert L.oracle_set_pinned(name.encode(), bytes([1] * n), n) == 0, name
    L.oracle_set_pinned_scalars(1, 1, 1, 1)
    for kind in range(3):
        k = 0
        while L.oracle_special_seq(kind, k) is not None:
            k += 1
        assert L.oracle_set_pinned_special(kind, bytes([1] * k), k) == 0


def test_gpu_says_which_energies_read_a_rule_or_model_value(bench_rows):
    """built-in tables: rafft_eval_structures_info flags exactly the structures whose energy reads an interior-loop table entry that
    no reference-held energy row exercises (the oracle tracks the same look-ups), and a fold counts the stem energies that did;
    a loaded parameter file - every entry ViennaRNA's - is never flagged"""
    params.reset_params()
    import json, os
    from conftest import ROOT
    J = json.load(open(os.path.join(ROOT, "params", "turner2004_fitted.json")))
    pinned = [k.split("|") for k in J["pinned"]]
    n11 = len({t for k in pinned if k[0] == "int11" for t in (tuple(k[1:]), (k[2], k[1], k[4], k[3]))})
    n21 = sum(1 for k in pinned if k[0] == "int21")
    n22 = len({t for k in pinned if k[0] == "int22" for t in (tuple(k[1:]), (k[2], k[1], k[5], k[6], k[3], k[4]))})
    assert params.unpinned_entries() == {"int11": 576 - n11, "int21": 2304 - n21, "int22": 9216 - n22}
    rng = np.random.default_rng(77)
    seqs = [r["seq"] for r in bench_rows[5::97] if len(r["seq"]) <= 500] + ["".join(rng.choice(list("ACGU"), 200)) for _ in range(12)]
    res = rafft_amd.fold_batch(seqs, 100, 50, 1000)
    st = rafft_amd.last_stats()
    assert st["n_dE_evals"] > 10000 and 0 < st["n_kept_guessed"] <= st["n_dE_guessed"] < 0.1 * st["n_dE_evals"]
    ss, dbs = [], []
    for s, beam in zip(seqs, res):
        for x in beam[:25]:
            ss.append(s); dbs.append(x.str_struct)
    dcal, status, guessed = R.eval_structures_info(ss, dbs)
    assert not any(status) and 0 < sum(guessed) < len(guessed)
    _oracle_tracks_the_interior_loop_tables_only()
    oracle.track(True)
    try:
        for s, db, d, g in zip(ss, dbs, dcal, guessed):
            od, n_unp = oracle.eval_structure_tracked(s, db)
            assert od == d and (n_unp > 0) == bool(g), (s, db, n_unp, g)
    finally:
        oracle.track(False)
        oracle.reset_tables()
    # ViennaRNA's own tables loaded (here: the built-in values written out and read back as a parameter file): nothing is a guess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "b.par")
        params.save_params(p)
        params.load_params(p)
        try:
            assert params.unpinned_entries() == {"int11": 0, "int21": 0, "int22": 0}
            res2 = rafft_amd.fold_batch(seqs, 100, 50, 1000)
            st2 = rafft_amd.last_stats()
            assert st2["n_dE_evals"] == 0 and st2["n_dE_guessed"] == 0 and st2["n_kept_guessed"] == 0
            assert [[(x.str_struct, x.dcal) for x in b] for b in res2] == [[(x.str_struct, x.dcal) for x in b] for b in res]
            assert R.eval_structures_info(ss[:50], dbs[:50])[2] == [0] * 50
        finally:
            params.reset_params()
