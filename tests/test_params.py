"""Energy-parameter plumbing, CPU only: the product's ViennaRNA parameter-file reader/writer (rafft_load_params /
rafft_save_params, rafft_amd/csrc/rafft_params.h) against the tests' own Python reader (tests/_par_reader.py).
No fold runs here - loading, inspecting and saving a parameter set needs no GPU."""
import json
import os
import sys
import types

import numpy as np
import pytest

import rafft_amd
from rafft_amd import _native, params
from conftest import ROOT
import _par_reader as PR


@pytest.fixture(autouse=True)
def builtin_params_afterwards():
    yield
    params.reset_params()


@pytest.fixture()
def synthetic_par(tmp_path):
    """the built-in tables with made-up enthalpies and a few changed values, as a ViennaRNA parameter file"""
    params.reset_params()
    p0 = tmp_path / "builtin.par"
    params.save_params(p0)
    par = PR.add_synthetic_enthalpies(PR.read_par(p0), seed=3)
    par["stack"] = par["stack"].copy(); par["stack"][0, 1] -= 20; par["stack"][1, 0] -= 20
    par["int22"] = par["int22"].copy(); par["int22"][2, 3, 1, 2, 3, 0] += 40
    par["hairpin"] = par["hairpin"].copy(); par["hairpin"][7] += 30
    par["Tetraloops"] = par["Tetraloops"] + [("GAAAAC", 120, 500)]
    path = tmp_path / "synthetic.par"
    PR.write_par(par, path, comment="synthetic: built-in 37 C values, invented enthalpies - loader tests only")
    return path, par


def test_builtin_tables_survive_save_and_independent_read(tmp_path):
    """params/turner2004_fitted.json -> header -> library -> rafft_save_params -> Python reader: same numbers"""
    J = json.load(open(os.path.join(ROOT, "params", "turner2004_fitted.json")))["tables"]
    p = tmp_path / "b.par"
    params.save_params(p)
    par = PR.read_par(p)
    assert params.params_info() == {"source": "built-in Turner 2004, 37 C (params/turner2004_tables.h)", "has_enthalpies": False}
    assert "stack_enthalpies" not in par
    assert np.array_equal(par["stack"][:6, :6], np.array(J["stack"]))
    assert np.array_equal(par["int22"], np.array(J["int22"]))
    assert np.array_equal(par["int21"][:6, :6, 1:, 1:, 1:], np.array(J["int21"]))
    assert np.array_equal(par["int11"][:6, :6, 1:, 1:], np.array(J["int11"]))
    assert np.array_equal(par["mismatch_multi"][:6, 1:, 1:], np.array(J["mismatch_multi"]))
    assert list(par["hairpin"]) == J["hairpin"] and list(par["bulge"]) == J["bulge"] and list(par["interior"]) == J["interior"]
    assert (par["ml_closing"], par["ml_intern"], par["ml_base"], par["terminal_au"], par["ninio"], par["max_ninio"]) == \
           (J["ml_closing"], J["ml_intern"], J["ml_base"], J["terminal_au"], J["ninio"], J["max_ninio"])
    assert {s: e for s, e, _ in par["Tetraloops"]} == J["tetraloops"] and abs(par["lxc"] - J["lxc"]) < 1e-9


def test_load_then_save_round_trip_and_entry_values(synthetic_par, tmp_path):
    path, par = synthetic_par
    params.load_params(path)
    info = params.params_info()
    assert info["has_enthalpies"] and info["source"] == str(path)
    rng = np.random.default_rng(0)
    # entries by ViennaRNA array shape: pair axes 0..7 (file rows are pairs 1..7), base axes 0..4
    for _ in range(300):
        t, u = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        a, b, c, d = (int(x) for x in rng.integers(0, 5, size=4))
        for dh in (False, True):
            sfx = "_enthalpies" if dh else ""
            assert params.param_value("stack", (t + 1) * 8 + (u + 1), dh) == par["stack" + sfx][t, u]
            assert params.param_value("mismatch_exterior", ((t + 1) * 5 + a) * 5 + b, dh) == par["mismatch_exterior" + sfx][t, a, b]
            assert params.param_value("dangle3", (t + 1) * 5 + a, dh) == par["dangle3" + sfx][t, a]
            assert params.param_value("int11", (((t + 1) * 8 + u + 1) * 5 + a) * 5 + b, dh) == par["int11" + sfx][t, u, a, b]
            assert params.param_value("int21", ((((t + 1) * 8 + u + 1) * 5 + a) * 5 + b) * 5 + c, dh) == par["int21" + sfx][t, u, a, b, c]
            if a and b and c and d:
                assert params.param_value("int22", (((((t + 1) * 8 + u + 1) * 5 + a) * 5 + b) * 5 + c) * 5 + d, dh) == \
                       par["int22" + sfx][t, u, a - 1, b - 1, c - 1, d - 1]
    assert params.param_value("hairpin", 7) == par["hairpin"][7] and params.param_value("ml_closing", 0, True) == par["ml_closing_dH"]
    # a 2x2 entry with an N base = maximum over the concrete bases at that place (ViennaRNA update_nst)
    assert params.param_value("int22", (((((2 + 1) * 8 + 3 + 1) * 5 + 0) * 5 + 2) * 5 + 3) * 5 + 1) == par["int22"][2, 3, :, 1, 2, 0].max()
    out = tmp_path / "again.par"
    params.save_params(out)
    again = PR.read_par(out)
    for k, v in par.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(again[k], v), k
        else:
            assert again[k] == v, k
    params.reset_params()
    assert not params.params_info()["has_enthalpies"]


def test_parameter_file_tokens_and_errors(synthetic_par, tmp_path):
    path, par = synthetic_par
    txt = open(path).read()
    # DEF keeps the built-in value, INF is ViennaRNA's INF, comments may span lines
    first = txt.index("# stack\n")
    row = txt.index("\n", txt.index("*/", first)) + 1
    eol = txt.index("\n", row)
    cells = txt[row:eol].split()
    cells[0], cells[1] = "DEF", "INF"
    params.load_params_text(txt[:row] + " ".join(cells) + " /* multi\nline */" + txt[eol:], "tokens")
    assert params.param_value("stack", 1 * 8 + 1) == -240            # built-in CG/CG stack kept
    assert params.param_value("stack", 1 * 8 + 2) == 10000000
    assert params.params_info()["source"] == "tokens"
    with pytest.raises(_native.RafftError, match="header"):
        params.load_params_text(txt.replace("## RNAfold parameter file v2.0", "## something else"))
    with pytest.raises(_native.RafftError, match="values, expected"):
        params.load_params_text(txt[:row] + txt[eol + 1:])                  # one stack row short
    with pytest.raises(_native.RafftError, match="bad token"):
        params.load_params_text(txt[:row] + " ".join(["12x"] + cells[2:] + cells[:1]) + txt[eol:])
    with pytest.raises(_native.RafftError, match="cannot open"):
        params.load_params(tmp_path / "nope.par")
    # a failed load leaves the previous set in place
    assert params.params_info()["source"] == "tokens"
    # a file without enthalpy sections loads, but cannot be rescaled
    params.load_params_text("\n# ".join(s for s in txt.split("\n# ") if not s.split("\n")[0].strip().endswith("_enthalpies")))
    assert not params.params_info()["has_enthalpies"]


def test_viennarna_is_touched_once_up_front(synthetic_par, monkeypatch):
    """a host with ViennaRNA gets ViennaRNA's own tables: `RNA.params_save` to a temporary file before the first fold
    (here a stand-in RNA module that writes the synthetic file); RAFFT_PARAMS takes precedence"""
    path, par = synthetic_par
    calls = []
    rna = types.ModuleType("RNA")
    rna.__version__ = "9.9.9-standin"

    def params_save(fname):
        calls.append(fname)
        open(fname, "w").write(open(path).read())
    rna.params_save = params_save
    monkeypatch.setitem(sys.modules, "RNA", rna)
    monkeypatch.setattr(params, "_auto_done", False)
    params.ensure_default_params()
    params.ensure_default_params()
    assert len(calls) == 1 and not os.path.exists(calls[0])
    info = params.params_info()
    assert info["has_enthalpies"] and "9.9.9-standin" in info["source"]
    assert params.param_value("hairpin", 7) == par["hairpin"][7]
    params.reset_params()
    monkeypatch.setattr(params, "_auto_done", False)
    monkeypatch.setenv("RAFFT_PARAMS", str(path))
    params.ensure_default_params()
    assert params.params_info()["source"] == str(path) and len(calls) == 1
    # and without ViennaRNA nothing changes
    params.reset_params()
    monkeypatch.delenv("RAFFT_PARAMS")
    monkeypatch.delitem(sys.modules, "RNA")
    monkeypatch.setattr(params, "_auto_done", False)
    params.ensure_default_params()
    assert not params.params_info()["has_enthalpies"]


def test_other_temperature_needs_enthalpies():
    """the built-in tables are 37 C only: any other temp is an error, never silently 37 C energies"""
    params.reset_params()
    with pytest.raises(_native.RafftError) as e:
        rafft_amd.fold("GGGAAACCC", temp=25.0)
    assert e.value.code == _native.ERR_TEMP


def test_python_rescale_matches_viennarna_formula(synthetic_par):
    """tests/_par_reader.tables_at: G(T) = dH - (dH - G37) * (T + K0) / Tmeasure, truncated toward zero"""
    path, par = synthetic_par
    T = PR.tables_at(par, 25.0)
    g, h = int(par["stack"][0, 0]), int(par["stack_enthalpies"][0, 0])
    assert T["stack"][1, 1] == int(h - (h - g) * ((25.0 + 273.15) / 310.15))
    assert np.array_equal(PR.tables_at(par, 37.0)["int21"][1:, 1:], par["int21"][:6, :6])
    assert (PR.tables_at(par, 60.0)["dangle5"] <= 0).all() and (PR.tables_at(par, 60.0)["mismatch_multi"] <= 0).all()
