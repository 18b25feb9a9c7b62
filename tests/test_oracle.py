"""The oracle (oracle/rafft_oracle.c) against the reference's own artefacts.

CPU-only.  Pins: 11 505 energy triples (benchmark_results/*_scores.csv), golden
vectors produced by the reference's Python (tools/make_golden.py), and the
reference's example trajectories (example/rafft.out, rafft_20.out).
"""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLD

EX = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"


def traj_text(seq, traj):
    out = [seq]
    for i, st in enumerate(traj):
        out.append("# {:-^20}".format(i))
        for s in st:
            out.append(f"{s.str_struct} {s.energy:6.1f}")
    return "\n".join(out) + "\n"


def test_energy_kats_exact(energy_kats):
    bad = [(s, st, d, oracle.eval_structure(s, st)) for s, st, d in energy_kats
           if oracle.eval_structure(s, st) != d]
    assert len(energy_kats) == 11505
    assert not bad, bad[:3]


@pytest.mark.parametrize("ms,fname", [(5, "example_rafft.out"), (20, "example_rafft_20.out")])
def test_reference_example_trajectories(ms, fname):
    _, traj = oracle.fold(EX, 100, ms, 1000, traj=True)
    assert traj_text(EX, traj) == open(os.path.join(GOLD, fname)).read()


def test_fold_trajectories_match_reference_python(fold_cases):
    for case in fold_cases:
        _, traj = oracle.fold(case["seq"], traj=True, **case["params"])
        got = [[[s.str_struct, s.dcal] for s in st] for st in traj]
        assert got == case["traj"], (case["seq"], case["params"])


def test_node_expansion_matches_reference_python(node_records):
    for r in node_records:
        cor = oracle.autocor(r["seq"], r["pos"], r["gc"], r["au"], r["gu"])
        np.testing.assert_array_equal(cor, np.array(r["cor"]))  # exact: integer weights, direct convolution
        ex = oracle.expand_node(r["seq"], r["db"], r["pos"], r["nb_mode"], r["min_hp"], r["min_nrj"],
                                r["gc"], r["au"], r["gu"])
        assert ex["lag"] == r["lags"]
        ws = [[a, b, c, d] for a, b, c, d in zip(ex["nb"], ex["mi"], ex["mj"], ex["score"])]
        assert ws == r["ws"]
        sol = [[ex["nb"][k], ex["score"][k], ex["mi"][k], ex["mj"][k], ex["ddcal"][k]] for k in ex["kept"]]
        assert sol == r["sol"]


def test_error_behaviour():
    with pytest.raises(KeyError):
        oracle.fold("acgu")
    with pytest.raises(KeyError):
        oracle.fold("ACGT")
    with pytest.raises(np.exceptions.AxisError):
        oracle.fold("")


def test_oracle_under_address_sanitizer():
    """the C oracle folds and evaluates under ASan + UBSan without a report (CPU build only)"""
    import subprocess
    here = os.path.join(os.path.dirname(GOLD), "..", "oracle")
    subprocess.check_call(["make", "-s", "-C", here, "asan_driver"])
    out = subprocess.run([os.path.join(here, "asan_driver")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("asan ok")
