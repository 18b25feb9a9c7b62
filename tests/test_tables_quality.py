"""How far the built-in Turner tables can be trusted (CPU only).

The 11 505 reference-held energy rows are reproduced exactly (tests/test_oracle.py) - by construction, the tables
were fitted to them.  These tests keep the OUT-OF-SAMPLE numbers honest: a by-sequence hold-out of the fit, and the
share of fold decisions that involve a table entry no row pins (tools/unpinned_stats.py has the full-size numbers,
profiles/r02_unpinned_lookups.json).  With a ViennaRNA parameter file loaded (rafft_load_params) none of this
applies: every entry is ViennaRNA's."""
import json
import os

import numpy as np

import oracle
from conftest import ROOT


def test_holdout_of_the_table_fit_one_fold():
    """fit on 4/5 of the sequences, evaluate the structures of the other 1/5 (tools/turner_fit/holdout.py):
    structures whose entries were all exercised in training are right (>98 %), structures touching an entry the
    training rows never exercised are right about half of the time - the unpinned entries are educated guesses"""
    from tools.turner_fit import holdout
    r = holdout.run(k=5, seed=0, folds=[0])["total"]
    assert r["test"] > 2000
    assert r["wrong_rate_when_all_seen"] < 0.02, r
    assert 0.25 < r["wrong_rate_when_touching_unseen"] < 0.75, r
    assert r["wrong_rate"] < 0.07, r
    committed = json.load(open(os.path.join(ROOT, "profiles", "r02_turner_holdout.json")))
    f0 = committed["folds"][0]
    assert f0["test"] == r["test"] and f0["exact_when_all_seen"] == r["exact_when_all_seen"] \
        and f0["exact_when_touching_unseen"] == r["exact_when_touching_unseen"]
    assert abs(committed["total"]["wrong_rate"] - 0.047) < 0.01


def test_share_of_fold_decisions_touching_unpinned_entries(bench_rows):
    """tracked oracle folds of every 40th benchmark sequence (n=100, ms=50): a few per cent of the dE evaluations
    involve an unpinned entry (committed full-size figure: 1.7 %), and the rate is not zero - 'drop-in' must not be
    read as 'bit-exact with ViennaRNA' for the built-in tables"""
    oracle.set_pinned(os.path.join(ROOT, "params", "turner2004_fitted.json"))
    oracle.track(True)
    try:
        c = {}
        n_final = n_final_unp = 0
        for r in bench_rows[::40]:
            if len(r["seq"]) > 400:
                continue
            oracle.fold(r["seq"], 100, 50, 1000, counters=c)
            n_final += len(c["final_unpinned"])
            n_final_unp += sum(1 for x in c["final_unpinned"] if x)
        share = c["dE_unpinned"] / (c["evals"] - c["children"])
        assert 0.003 < share < 0.06, share
        assert 0 < n_final_unp < 0.25 * n_final
        # a structure made of pinned entries only is reported as such, one with a guessed 2x2 loop is not
        d, n = oracle.eval_structure_tracked("GGGGAAAACCCC", "((((....))))")
        assert n == 0
    finally:
        oracle.track(False)
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_unpinned_lookups.json")))
    assert 0.01 < full["cfg3_benchmark_set"]["dE_share"] < 0.03
    assert full["cfg3_vs_published_lowest_energy"]["identical"] == 2141
