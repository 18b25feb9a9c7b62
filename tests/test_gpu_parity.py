"""Parity of the HIP path (through the C-ABI) against the oracle and the golden
fixtures.  Everything here needs a real MI355X: run with `-m gpu`."""
import os

import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import rafft as R
from conftest import GOLD

pytestmark = pytest.mark.gpu

EX = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"


def as_lists(traj):
    return [[[s.str_struct, s.dcal] for s in st] for st in traj]


@pytest.fixture(autouse=True, params=["classes_by_size", "classes_merged", "general_builds"])
def expand_class_routing(request, monkeypatch):
    """Steps with few new structures send all their regions to one wide expand kernel (the tail of a big batch;
    every step of the small batches in this file).  Each test runs both ways, so that the one-wavefront kernel
    (popcount correlation, bit-mask window_slide) and the wide kernels see the same cases - and a third time with the
    general builds of the expand kernels (diagnostics, seam and FFT paths compiled in) instead of the production builds."""
    if request.param in ("classes_by_size", "general_builds"):
        monkeypatch.setenv("RAFFT_MERGE_BELOW", "0")
        monkeypatch.setenv("RAFFT_MERGE2_BELOW", "0")
    if request.param == "general_builds":
        monkeypatch.setenv("RAFFT_PROD", "0")
    yield


def test_gpu_energy_kats_exact(energy_kats):
    """Device Turner-2004 evaluator == the reference's 11 505 published energies (exact dcal)."""
    seqs = [k[0] for k in energy_kats]
    dbs = [k[1] for k in energy_kats]
    got, st = R.eval_structures(seqs, dbs)
    assert not any(st)
    bad = [(i, g, k[2]) for i, (g, k) in enumerate(zip(got, energy_kats)) if g != k[2]]
    assert not bad, bad[:5]


@pytest.mark.parametrize("ms,fname", [(5, "example_rafft.out"), (20, "example_rafft_20.out")])
def test_gpu_reference_example_trajectories(ms, fname):
    fin, traj = rafft_amd.fold(EX, 100, ms, 1000, traj=True)
    assert rafft_amd.format_trajectory(EX, traj) == open(os.path.join(GOLD, fname)).read()
    assert [s.str_struct for s in fin] == [s.str_struct for s in traj[-1]]


def test_gpu_expand_node_vs_reference_python_and_oracle(node_records):
    """correlation profile, lag ranking, window_slide tuples, dE and kept order for single regions"""
    for r in node_records:
        g = R.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
        o = oracle.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
        assert g["lag"] == r["lags"] == o["lag"]
        cor_sorted = [r["cor"][k] for k in r["lags"]]
        assert g["cor"] == cor_sorted            # bit-exact fp64: integer counts, IEEE divide
        ws = [[a, b, c, d] for a, b, c, d in zip(g["nb"], g["mi"], g["mj"], g["score"])]
        assert ws == r["ws"]
        assert g["ddcal"] == o["ddcal"]
        assert g["kept"] == o["kept"]
        sol = [[g["nb"][k], g["score"][k], g["mi"][k], g["mj"][k], g["ddcal"][k]] for k in g["kept"]]
        assert sol == r["sol"]


def test_gpu_fold_matches_reference_python_golden(fold_cases):
    """full trajectories for every golden (sequence, params) case, batched per parameter set"""
    groups = {}
    for c in fold_cases:
        groups.setdefault(tuple(sorted(c["params"].items())), []).append(c)
    for key, cases in groups.items():
        got = rafft_amd.fold_batch([c["seq"] for c in cases], traj=True, **dict(key))
        for c, (fin, traj) in zip(cases, got):
            assert as_lists(traj) == c["traj"], (c["seq"], c["params"])


def test_gpu_cfg2_random_L200_vs_oracle():
    """BASELINE configs[1] shape (L=200 i.i.d., n=100, ms=50), reduced count"""
    rng = np.random.default_rng(200)
    seqs = ["".join(rng.choice(list("ACGU"), 200)) for _ in range(48)]
    got = rafft_amd.fold_batch(seqs, 100, 50, 1000, traj=True)
    for s, (fin, traj) in zip(seqs, got):
        _, o = oracle.fold(s, 100, 50, 1000, traj=True)
        assert as_lists(traj) == as_lists(o), s


def test_gpu_mixed_lengths_and_classes_vs_oracle():
    """ragged batch crossing all three expand size classes (n up to 1500), with N bases"""
    rng = np.random.default_rng(3000)
    lens = [1, 2, 5, 17, 33, 64, 129, 257, 300, 511, 700, 1025, 1500]
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    seqs.append("".join(rng.choice(list("ACGUN"), 150, p=[.22, .22, .22, .22, .12])))
    got = rafft_amd.fold_batch(seqs, 100, 8, 1000, traj=True)
    for s, (fin, traj) in zip(seqs, got):
        _, o = oracle.fold(s, 100, 8, 1000, traj=True)
        assert as_lists(traj) == as_lists(o), len(s)


def test_gpu_final_only_equals_last_step():
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGU"), 90)) for _ in range(16)]
    a = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=False)
    b = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=True)
    for fin, (fin2, traj) in zip(a, b):
        assert [(s.str_struct, s.dcal) for s in fin] == [(s.str_struct, s.dcal) for s in traj[-1]]


def test_gpu_error_behaviour():
    with pytest.raises(KeyError):
        rafft_amd.fold("acgu")
    with pytest.raises(KeyError):
        rafft_amd.fold("ACGT")
    with pytest.raises(np.exceptions.AxisError):
        rafft_amd.fold("")
    res = rafft_amd.fold_batch(["GGGAAACCC", "ACGT", ""], max_stack=3, raise_errors=False)
    assert res[1] is None and res[2] is None and res[0][0].str_struct == oracle.fold("GGGAAACCC", max_stack=3)[0].str_struct
    with pytest.raises(Exception):
        rafft_amd.fold("GGGAAACCC", temp=25.0)


def test_gpu_arena_overflow_regrows_and_stays_exact(monkeypatch):
    """start with starved HBM arenas (on memory still holding an earlier batch): the overflow flags
    must stop the step cleanly, the wave is re-run with doubled arenas and results are unchanged"""
    rng = np.random.default_rng(21)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(60, 400, size=40)]
    want = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=True)
    monkeypatch.setenv("RAFFT_EST", "0.05")
    got = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=True)
    monkeypatch.delenv("RAFFT_EST")
    for (f1, t1), (f2, t2) in zip(want, got):
        assert as_lists(t1) == as_lists(t2)


def test_gpu_single_sequences_need_no_regrowth():
    """the arena planner must hold for the smallest batch too (one sequence lands on few sub-arenas): a lone
    sequence of any length folds in one go - a regrowth would silently triple the latency of the CLI"""
    rng = np.random.default_rng(8)
    for L, ms in ((60, 50), (600, 50), (1500, 20), (2500, 50)):
        s = "".join(rng.choice(list("ACGU"), L))
        fin = rafft_amd.fold(s, 100, ms, 1000)
        assert 1 <= len(fin) <= ms and rafft_amd.last_stats()["n_regrows"] == 0, (L, ms, rafft_amd.last_stats())


def test_gpu_regrowth_after_early_harvest_stays_exact(monkeypatch):
    """without --traj the rows of finished sequences leave early through a copy stream; an arena overflow after
    that abandons the wave, and the re-run must replace every result"""
    rng = np.random.default_rng(23)
    lens = [int(n) for n in rng.integers(40, 120, size=300)] + [600, 700]
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    want = rafft_amd.fold_batch(seqs, 100, 20, 1000)
    assert rafft_amd.last_stats()["n_regrows"] == 0
    monkeypatch.setenv("RAFFT_TEST_OVF_AT", "9")       # the early harvest of this batch happens after step 6
    monkeypatch.setenv("RAFFT_SPLIT", "0")
    got = rafft_amd.fold_batch(seqs, 100, 20, 1000)
    assert rafft_amd.last_stats()["n_regrows"] == 1
    for f1, f2 in zip(want, got):
        assert [(s.str_struct, s.dcal) for s in f1] == [(s.str_struct, s.dcal) for s in f2]


def test_gpu_fft_and_direct_correlation_agree(monkeypatch, node_records):
    """short regions use the popcount form, long ones the LDS FFT; forcing the FFT everywhere must not
    change a single lag value, rank or trajectory"""
    rng = np.random.default_rng(31)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(20, 140, size=24)]
    a = rafft_amd.fold_batch(seqs, 100, 10, 1000, traj=True)
    monkeypatch.setenv("RAFFT_FORCE_FFT", "1")
    b = rafft_amd.fold_batch(seqs, 100, 10, 1000, traj=True)
    for r in node_records[:60]:
        g = R.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
        assert g["lag"] == r["lags"] and g["cor"] == [r["cor"][k] for k in r["lags"]]
    monkeypatch.delenv("RAFFT_FORCE_FFT")
    for (f1, t1), (f2, t2) in zip(a, b):
        assert as_lists(t1) == as_lists(t2)
    # the wide classes: FFT everywhere (limit 0), the default split at 1024 positions, the direct form everywhere (regions of
    # up to 1500 positions here); the one-wavefront class keeps its own popcount / FFT split
    longer = ["".join(rng.choice(list("ACGU"), int(n))) for n in (150, 400, 700, 1100, 1500)]
    runs = {}
    for lim in ("0", "1024", "4096"):
        monkeypatch.setenv("RAFFT_DIRECT_N", lim)
        runs[lim] = [as_lists(t) for _, t in rafft_amd.fold_batch(longer, 100, 6, 1000, traj=True)]
    monkeypatch.delenv("RAFFT_DIRECT_N")
    assert runs["0"] == runs["1024"] == runs["4096"]
    # regions of 1025-4096 positions: by default the FFT-free kernel (direct correlation, lag values in HBM, three workgroups per CU),
    # with RAFFT_C3_DIRECT=0 the LDS FFT plan (one workgroup per CU)
    big = longer[3:] + ["".join(rng.choice(list("ACGU"), 2600))]
    d3 = [as_lists(t) for _, t in rafft_amd.fold_batch(big, 100, 4, 1000, traj=True)]
    monkeypatch.setenv("RAFFT_C3_DIRECT", "0")
    f3 = [as_lists(t) for _, t in rafft_amd.fold_batch(big, 100, 4, 1000, traj=True)]
    monkeypatch.delenv("RAFFT_C3_DIRECT")
    assert d3 == f3
    # (by default both kernels are launched on the class's work list and its length decides on the device which one works:
    #  RAFFT_C3_SWITCH = 0 -> always the FFT-free kernel, huge -> always the FFT plan, 2 -> they alternate from step to step here)
    for sw in ("0", "1000000", "2"):
        monkeypatch.setenv("RAFFT_C3_SWITCH", sw)
        assert [as_lists(t) for _, t in rafft_amd.fold_batch(big, 100, 4, 1000, traj=True)] == d3, sw
    monkeypatch.delenv("RAFFT_C3_SWITCH")
    _, o = oracle.fold(longer[2], 100, 6, 1000, traj=True)
    assert runs["1024"][2] == as_lists(o)


def test_gpu_small_region_kernel_is_interchangeable(monkeypatch, node_records):
    """regions of up to 32 positions whose every lag is searched go to expand_small_kernel (teams of 16 / 32 lanes);
    with the class switched off, or its limits moved, the general kernel expands them: same trajectories, same
    per-region records (reference-Python fixtures), both equal to the oracle"""
    rng = np.random.default_rng(1632)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(9, 420, size=40)]
    seqs += ["GGGAAACCC", "GCGCGCGCGCGCGC", "GGGNNNNCCC", "AUAUAUAUAUAUAUAUAUAU", "".join(rng.choice(list("ACGUN"), 150, p=[.23, .23, .23, .23, .08]))]
    runs = {}
    for lim in ("16,32", "0,0", "8,16", "0,32", "16,16"):
        monkeypatch.setenv("RAFFT_SMALL", lim)
        runs[lim] = [as_lists(t) for _, t in rafft_amd.fold_batch(seqs, 100, 12, 1000, traj=True)]
        small = [r for r in node_records if len(r["pos"]) <= 32 and 2 * len(r["pos"]) - 1 <= r["nb_mode"]]
        assert len(small) > 20
        for r in small:
            g = R.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"], r["gc"], r["au"], r["gu"])
            assert g["lag"] == r["lags"] and g["cor"] == [r["cor"][k] for k in r["lags"]]
            assert [[a, b, c, d] for a, b, c, d in zip(g["nb"], g["mi"], g["mj"], g["score"])] == r["ws"]
            assert [[g["nb"][k], g["score"][k], g["mi"][k], g["mj"][k], g["ddcal"][k]] for k in g["kept"]] == r["sol"]
    monkeypatch.delenv("RAFFT_SMALL")
    for lim, t in runs.items():
        assert t == runs["0,0"], lim
    for s, t in zip(seqs[:12] + seqs[40:], runs["16,32"][:12] + runs["16,32"][40:]):
        _, o = oracle.fold(s, 100, 12, 1000, traj=True)
        assert t == as_lists(o), len(s)
    st = rafft_amd.last_stats()
    assert st["n_node_expansions"] > 0
    # a loop with few unpaired positions and MANY branches (n <= 16, 17..32 helices): with equal limits ("16,16") the 32-lane class is
    # off and such a region must take the general kernel - it used to be routed to the class that is never launched and was dropped
    hp = "GGGGAAAACCCC"
    seq = "GA" + hp * 10 + "AC" + hp * 10 + "GU"
    db = ".." + "((((....))))" * 10 + ".." + "((((....))))" * 10 + ".."
    pos = [0, 1, 122, 123, 244, 245]
    for lim in ("16,16", "16,32", "0,0"):
        monkeypatch.setenv("RAFFT_SMALL", lim)
        g = R.expand_node(seq, db, pos, 100, 0, 0.0)
        o = oracle.expand_node(seq, db, pos, 100, 0, 0.0)
        assert g["lag"] == o["lag"] and g["ddcal"] == o["ddcal"] and g["kept"] == o["kept"] and len(o["kept"]) > 0, lim
    monkeypatch.delenv("RAFFT_SMALL")


def test_gpu_beam_region_lists_not_resident(monkeypatch):
    """the beam step keeps every member's regions-with-a-choice in LDS; members that do not fit are decoded
    from the productive-region list in HBM instead - same trajectories either way, and both equal the oracle"""
    rng = np.random.default_rng(33)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in (70, 180, 333, 512, 900)]
    a = rafft_amd.fold_batch(seqs, 100, 30, 200, traj=True)
    monkeypatch.setenv("RAFFT_RL_CAP", "6")
    monkeypatch.setenv("RAFFT_MAT_TILE", "3")      # and materialize_kernel in several tiles per structure
    b = rafft_amd.fold_batch(seqs, 100, 30, 200, traj=True)
    monkeypatch.delenv("RAFFT_RL_CAP")
    monkeypatch.delenv("RAFFT_MAT_TILE")
    for s, (f1, t1), (f2, t2) in zip(seqs, a, b):
        assert as_lists(t1) == as_lists(t2)
        if len(s) <= 333:
            _, o = oracle.fold(s, 100, 30, 200, traj=True)
            assert as_lists(t1) == as_lists(o)


@pytest.mark.parametrize("mb,ms", [(50, 50), (3, 40), (1000, 1), (17, 200)])
def test_gpu_small_max_branch_large_beam_vs_oracle(mb, ms):
    """LDS carve-up of the beam step must hold for any (max_branch, max_stack) mix"""
    rng = np.random.default_rng(41)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in (90, 150, 260, 420)]
    got = rafft_amd.fold_batch(seqs, 100, ms, mb, traj=True)
    for s, (fin, traj) in zip(seqs, got):
        _, o = oracle.fold(s, 100, ms, mb, traj=True)
        assert as_lists(traj) == as_lists(o), (len(s), mb, ms)


def test_gpu_randomised_parameters_vs_oracle():
    """random lengths x (nb_mode, max_stack, max_branch, min_hp, weights): every code path of the
    expand kernel (popcount / FFT correlation, skip / bitonic / radix-select ranking, mask / chunked
    window_slide) and of the beam step against the oracle, full trajectories"""
    rng = np.random.default_rng(77)
    combos = [(5, 1, 2, 3, (3.0, 2.0, 1.0)), (30, 7, 100, 2, (3.0, 2.0, 1.0)), (100, 50, 1000, 3, (3.0, 2.0, 1.0)),
              (300, 7, 1000, 6, (1.0, 1.0, 1.0)), (100, 20, 40, 3, (2.5, 1.75, 0.5)), (64, 3, 5, 4, (3.0, 2.0, 0.0)),
              (1000, 10, 1000, 3, (3.0, 2.0, 1.0))]
    for nb_mode, ms, mb, hp, (gc, au, gu) in combos:
        lens = [int(x) for x in rng.integers(8, 330, size=10)] + [int(rng.integers(400, 900))]
        seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
        got = rafft_amd.fold_batch(seqs, nb_mode, ms, mb, hp, 0.0, True, 37.0, gc, au, gu)
        for s, (fin, traj) in zip(seqs, got):
            _, o = oracle.fold(s, nb_mode, ms, mb, hp, 0.0, True, 37.0, gc, au, gu)
            assert as_lists(traj) == as_lists(o), (len(s), nb_mode, ms, mb, hp, gc, au, gu)


@pytest.mark.parametrize("nb_mode,mb", [(0, 100), (100, 0), (1, 1)])
def test_gpu_degenerate_parameters_vs_oracle(nb_mode, mb):
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in (30, 80, 200)]
    got = rafft_amd.fold_batch(seqs, nb_mode, 5, mb, traj=True)
    for s, (fin, traj) in zip(seqs, got):
        _, o = oracle.fold(s, nb_mode, 5, mb, traj=True)
        assert as_lists(traj) == as_lists(o), (len(s), nb_mode, mb)


def test_gpu_large_nb_mode_on_long_sequences_vs_oracle():
    """nb_mode far beyond the default on sequences that reach the class with the 128-KiB FFT buffers (regions of more than
    1024 positions): the per-lag arrays then live in the free half of the FFT area instead of making the call an error
    (the reference takes any nb_mode, rafft/rafft.py:219-221)"""
    rng = np.random.default_rng(1500)
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in (1500, 2600, 300)]
    for nb_mode, ms in ((1000, 2), (2047, 1), (450, 3)):
        got = rafft_amd.fold_batch(seqs, nb_mode, ms, 1000, traj=True)
        for s, (fin, traj) in zip(seqs, got):
            _, o = oracle.fold(s, nb_mode, ms, 1000, traj=True)
            assert as_lists(traj) == as_lists(o), (len(s), nb_mode, ms)


def test_gpu_job_whose_candidate_table_would_outgrow_its_31_bit_slot_ids_is_split(monkeypatch):
    """a child slot is named by 2 x candidate + side in 31 bits (node lists hold -(slot + 1)): a job planned for more than 2^30
    candidate records is folded in halves.  (Found the hard way: with the dot-bracket arena gone the byte budget of a wave stopped
    keeping BASELINE configs[3] on one GPU below that - a negative slot, a memory fault.)  The hook lowers the limit to the 1 M
    floor, so 200 sequences of 200 nt become eight waves; same beams."""
    rng = np.random.default_rng(230)
    seqs = ["".join(rng.choice(list("ACGU"), 200)) for _ in range(200)]
    want = [[(x.str_struct, x.dcal) for x in r] for r in rafft_amd.fold_batch(seqs, 100, 50, 1000)]
    launches_one = rafft_amd.last_stats()["n_expand_launches"]
    monkeypatch.setenv("RAFFT_TEST_CAND_LIMIT", "1")
    got = [[(x.str_struct, x.dcal) for x in r] for r in rafft_amd.fold_batch(seqs, 100, 50, 1000)]
    st = rafft_amd.last_stats()
    monkeypatch.delenv("RAFFT_TEST_CAND_LIMIT")
    assert got == want
    # (launches of the one-wavefront expand kernel, summed over the waves of the call)
    assert st["n_expand_launches"] > 2 * launches_one and st["n_regrows"] == 0, (st["n_expand_launches"], launches_one)


def test_gpu_more_productive_regions_than_the_short_lists_hold(monkeypatch):
    """a structure with more productive regions than materialize_kernel's short LDS lists hold (64; here the hook makes it 3)
    is no error: the wave is folded again with the long lists - same trajectories, one regrowth on record - and later waves
    with the same parameters start with the long lists (no second double fold)"""
    from rafft_amd import _native
    _native.lib().rafft_shutdown()          # a fresh scheduler: it remembers which parameter sets needed the long lists
    rng = np.random.default_rng(256)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in (300, 500, 120, 800)]
    want = [as_lists(t) for _, t in rafft_amd.fold_batch(seqs, 100, 8, 1000, traj=True)]
    monkeypatch.setenv("RAFFT_TEST_MAX_PROD", "3")
    got = [as_lists(t) for _, t in rafft_amd.fold_batch(seqs, 100, 8, 1000, traj=True)]
    st = rafft_amd.last_stats()
    assert st["n_regrows"] >= 1 and st["n_regrows_prod"] >= 1
    again = [as_lists(t) for _, t in rafft_amd.fold_batch(seqs, 100, 8, 1000, traj=True)]
    assert rafft_amd.last_stats()["n_regrows"] == 0
    assert rafft_amd.last_stats()["n_waves_long_lists"] >= 1          # ... and the library says so
    monkeypatch.delenv("RAFFT_TEST_MAX_PROD")
    assert got == want and again == want
    # the flag decays: eight waves in a row that never needed the long lists, and the short ones (and the four-structures-per-
    # wavefront materialize kernel with them) are back - one outlier batch does not slow a process down for good
    n_long = 0
    for _ in range(10):
        assert [as_lists(t) for _, t in rafft_amd.fold_batch(seqs, 100, 8, 1000, traj=True)] == want
        n_long += rafft_amd.last_stats()["n_waves_long_lists"]
    assert 6 <= n_long < 20 and rafft_amd.last_stats()["n_waves_long_lists"] == 0      # (the fold above was the first of the eight)
    _native.lib().rafft_shutdown()


def test_gpu_long_stems_vs_oracle():
    """perfect and wobbled hairpin stems of 15, 16, 17, 20, 33 and 60 pairs (round 5: a contiguous stem of up to 16 pairs takes its
    stacking energies from the packed strands, one look-up per pair, and its pair hash from two mixes; longer stems take the pair-by-pair
    forms - both sides of the line, nested and side by side, bare and with flanks): full trajectories equal the oracle's"""
    rng = np.random.default_rng(1617)
    comp = {"A": "U", "U": "A", "G": "C", "C": "G"}
    seqs = []
    for nb in (15, 16, 17, 20, 33, 60):
        left = "".join(rng.choice(list("GCAU"), nb))
        right = "".join(comp[c] for c in reversed(left))
        wob = "".join(("U" if (c == "C" and rng.random() < 0.3) else c) for c in right)       # G-C -> G.U here and there
        seqs += ["G" * nb + "GAAA" + "C" * nb, left + "UUCG" + right, "AC" + left + "GCAA" + wob + "UUA",
                 left + "GAAA" + right + "AAAA" + left[::-1] + "UUUU" + "".join(comp[c] for c in left)]
    from _oracle_pool import fold_many
    want = fold_many([(s, 100, 6, 1000, True) for s in seqs])
    got = rafft_amd.fold_batch(seqs, 100, 6, 1000, traj=True)
    for k, (fin, traj) in enumerate(got):
        assert [[(x.str_struct, x.dcal) for x in st] for st in traj] == want[k], (k, seqs[k])
    assert any(x.str_struct.count("(") >= 60 for fin, _ in got for x in fin)       # the 60-pair stems do form


def test_gpu_seen_tables_sized_from_the_length_and_grown(monkeypatch):
    """round 5: a sequence's `seen` set starts in a table sized from its length (seen_slots0, rafft_api.hip) so that the benchmark set
    folds without a rehash; a set that outgrows its table is still rehashed into one of twice the size inside beam_step_kernel.  Both
    sides: random sequences of 60-900 nt with the default tables, with fixed 8192-slot tables (RAFFT_SEEN_FIXED=1: everything beyond
    ~200 nt grows once or twice) and at max_stack 150 (three times the children per step the sizing was measured at) - same beams, and
    the oracle's for the shorter ones"""
    rng = np.random.default_rng(77)
    seqs = ["".join(rng.choice(list("ACGU"), int(L))) for L in (60, 90, 120, 150, 220, 260, 330, 420, 640, 900)]
    base = rafft_amd.fold_batch(seqs, 100, 50, 1000)
    from _oracle_pool import fold_many
    want = fold_many([(s, 100, 50, 1000, False) for s in seqs[:6]])
    for k in range(6):
        assert [(x.str_struct, x.dcal) for x in base[k]] == want[k], k
    monkeypatch.setenv("RAFFT_SEEN_FIXED", "1")
    fixed = rafft_amd.fold_batch(seqs, 100, 50, 1000)
    monkeypatch.delenv("RAFFT_SEEN_FIXED")
    assert [[(x.str_struct, x.dcal) for x in f] for f in fixed] == [[(x.str_struct, x.dcal) for x in f] for f in base]
    wide = rafft_amd.fold_batch(seqs[:8], 100, 150, 1000)
    monkeypatch.setenv("RAFFT_SEEN_FIXED", "1")
    wide_fixed = rafft_amd.fold_batch(seqs[:8], 100, 150, 1000)
    assert [[(x.str_struct, x.dcal) for x in f] for f in wide] == [[(x.str_struct, x.dcal) for x in f] for f in wide_fixed]


def test_gpu_beam_wider_than_the_workgroup_vs_oracle(monkeypatch):
    """round 5: the beam step's flat prepass reads one structure row per thread and numbers the (member, region) items by a prefix sum
    over the members - a beam wider than the workgroup takes several rounds of both (256-thread kernel forced with RAFFT_WIDE_BELOW=0,
    max_stack 300 and 700; the 1024-thread kernel at max_stack 1100): final beams equal the oracle's"""
    rng = np.random.default_rng(4242)
    seqs = ["".join(rng.choice(list("ACGU"), int(L))) for L in (70, 90, 110, 130, 150, 180)]
    from _oracle_pool import fold_many
    for ms, wide_below in ((300, "0"), (700, "0"), (1100, None)):
        if wide_below is None:
            monkeypatch.delenv("RAFFT_WIDE_BELOW", raising=False)
        else:
            monkeypatch.setenv("RAFFT_WIDE_BELOW", wide_below)
        sub = seqs if ms == 300 else seqs[:3]
        want = fold_many([(s, 100, ms, 1000, False) for s in sub])
        got = rafft_amd.fold_batch(sub, 100, ms, 1000)
        for k in range(len(sub)):
            assert [(x.str_struct, x.dcal) for x in got[k]] == want[k], (ms, k)
        assert max(len(g) for g in got) > 256 or ms == 300
