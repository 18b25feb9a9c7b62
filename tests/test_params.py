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


def test_unreadable_parameter_file_is_an_error_at_every_fold(monkeypatch, tmp_path):
    """RAFFT_PARAMS names a file that is missing or malformed: every fold fails loudly - never a silent fall back to the built-in
    tables (whose unexercised entries are rule / model values; the caller asked for ViennaRNA's own, rafft/utils.py:17-21)"""
    params.reset_params()
    bad = tmp_path / "broken.par"
    bad.write_text("## RNAfold parameter file v2.0\n\n# stack\n 1 2 x3\n")
    for target in (str(tmp_path / "missing.par"), str(bad)):
        monkeypatch.setenv("RAFFT_PARAMS", target)
        monkeypatch.setattr(params, "_auto_done", False)
        for _ in range(2):                       # the second call fails like the first
            with pytest.raises(RuntimeError, match="RAFFT_PARAMS"):
                params.ensure_default_params()
        assert not params.params_info()["has_enthalpies"]       # nothing was loaded
    monkeypatch.delenv("RAFFT_PARAMS")
    monkeypatch.setenv("RAFFT_NO_VIENNARNA", "1")
    monkeypatch.setattr(params, "_auto_done", False)
    params.ensure_default_params()               # without the variable: the built-in set, as before


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


def test_saved_file_has_the_layout_viennarna_writes(synthetic_par, tmp_path):
    """The text `rafft_save_params` writes, section by section, against the layout of ViennaRNA 2.x parameter files
    (`RNA.params_save`, misc/rna_turner2004.par; format "## RNAfold parameter file v2.0", ViennaRNA
    src/ViennaRNA/params/io.c): header line first; sections in ViennaRNA's order, every energy array followed by its
    `_enthalpies` twin; one row per innermost-but-one index; the NN / N rows and columns of non-standard pairs and bases -
    which the fold never reads - present with ViennaRNA's shapes (7 pair rows, 5 base columns; int22 alone has 6 x 6 x 4^4);
    comments only as /* */; `# END` last.  Read back by a reader that shares no code with the library."""
    path, par = synthetic_par
    params.load_params(path)
    out = tmp_path / "layout.par"
    params.save_params(out)
    params.reset_params()
    lines = open(out).read().split("\n")
    assert lines[0] == "## RNAfold parameter file v2.0"
    heads = [(i, l[2:].strip()) for i, l in enumerate(lines) if l.startswith("# ")]
    names = [n for _, n in heads]
    arrays = ["stack", "mismatch_hairpin", "mismatch_interior", "mismatch_interior_1n", "mismatch_interior_23", "mismatch_multi",
              "mismatch_exterior", "dangle5", "dangle3", "int11", "int21", "int22", "hairpin", "bulge", "interior"]
    want = [x for a in arrays for x in (a, a + "_enthalpies")] + ["NINIO", "ML_params", "Misc", "Hexaloops", "Tetraloops", "Triloops", "END"]
    assert names == want
    # rows of every array section: (rows, values per row) with comments stripped
    strip = lambda l: __import__("re").sub(r"/\*.*?\*/", " ", l).split()
    rows_of = {}
    for (i, n), (j, _) in zip(heads, heads[1:]):
        body = [strip(l) for l in lines[i + 1:j]]
        rows_of[n] = [r for r in body if r]
    shape_rows = {"stack": (7, 7), "mismatch_hairpin": (35, 5), "mismatch_interior": (35, 5), "mismatch_interior_1n": (35, 5),
                  "mismatch_interior_23": (35, 5), "mismatch_multi": (35, 5), "mismatch_exterior": (35, 5), "dangle5": (7, 5), "dangle3": (7, 5),
                  "int11": (7 * 7 * 5, 5), "int21": (7 * 7 * 5 * 5, 5), "int22": (6 * 6 * 4 * 4 * 4, 4)}
    for n, (nr, nc) in shape_rows.items():
        for sfx in ("", "_enthalpies"):
            assert len(rows_of[n + sfx]) == nr and all(len(r) == nc for r in rows_of[n + sfx]), (n + sfx, len(rows_of[n + sfx]))
    for n in ("hairpin", "bulge", "interior"):
        for sfx in ("", "_enthalpies"):
            toks = [t for r in rows_of[n + sfx] for t in r]
            assert len(toks) == 31 and toks[0] == "INF"                  # size 0 cannot occur: ViennaRNA writes INF there
    assert [len(r) for r in rows_of["NINIO"]] == [3] and [len(r) for r in rows_of["ML_params"]] == [6] and [len(r) for r in rows_of["Misc"]] == [6]
    for n, ln in (("Hexaloops", 8), ("Tetraloops", 6), ("Triloops", 5)):
        assert rows_of[n] and all(len(r) == 3 and len(r[0]) == ln and set(r[0]) <= set("ACGU") for r in rows_of[n])
    # every token is an integer, INF or (Misc: lxc) a decimal; nothing but /* */ comments between sections
    for n, rows in rows_of.items():
        for r in rows:
            for t in (r[1:] if n in ("Hexaloops", "Tetraloops", "Triloops") else r):
                assert t == "INF" or __import__("re").fullmatch(r"-?\d+(\.\d+)?", t), (n, t)
    # non-standard rows / columns (pair NN = last pair row, base N = first base column): present, and what the independent reader reads back
    again = PR.read_par(out)
    for k, v in par.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(again[k], v), k
    assert again["stack"].shape == (7, 7) and again["mismatch_multi"].shape == (7, 5, 5) and again["int11"].shape == (7, 7, 5, 5)


def test_malformed_blocks_are_rejected_with_their_line(synthetic_par):
    path, par = synthetic_par
    lines = open(path).read().split("\n")
    i = lines.index("# int11")
    first_row = next(k for k in range(i + 1, len(lines)) if lines[k].strip() and not lines[k].strip().startswith("/*"))

    def load(ls, tag):
        with pytest.raises(_native.RafftError) as ei:
            params.load_params_text("\n".join(ls), tag)
        return str(ei.value)
    bad = list(lines); bad[first_row] = bad[first_row].replace(bad[first_row].split()[0], "x1y", 1)
    assert f"line {first_row + 1}: bad token 'x1y'" in load(bad, "badtoken")
    bad = list(lines); del bad[first_row + 2]
    msg = load(bad, "shortrow")
    assert "section '# int11' (line %d)" % (i + 1) in msg and "1220 values, expected 1225" in msg and "block ends on line" in msg
    bad = list(lines); bad.insert(first_row + 1, "  1 2 3")
    msg = load(bad, "surplus")
    assert "1228 values, expected 1225" in msg and "first surplus value on line" in msg
    last = max(k for k, l in enumerate(lines) if "*/" in l)
    bad = lines[:last + 2] + ["/* never closed"] + [l for l in lines[last + 2:] if "*/" not in l]
    assert f"line {last + 3}: unterminated comment" in load(bad, "comment")
    assert params.params_info()["source"].startswith("built-in")         # failed loads changed nothing


def test_parameter_reader_under_sanitizers(synthetic_par, tmp_path):
    """rafft_params.h (reader, writer, temperature rescaling - the library's one parser of untrusted input) built host-only with
    ASan + UBSan and fed well-formed files, the malformed corpus of the tests above, truncations, byte flips and oversized
    files: no sanitizer report, every file either accepted (and stable under write + re-read) or rejected with a message.
    (CPU build only: GPU AddressSanitizer is not available; this code needs no GPU.)"""
    import subprocess
    path, par = synthetic_par
    exe = tmp_path / "params_san_driver"
    src = os.path.join(ROOT, "tests", "san", "params_san_driver.cpp")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-host-only", "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer",
                           "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all", src, "-o", str(exe)])
    text = open(path).read()
    builtin = tmp_path / "builtin_only.par"
    params.reset_params()
    params.save_params(builtin)
    lines = text.split("\n")
    corpus = {"good_synthetic": text, "good_builtin": open(builtin).read(), "empty": "", "header_only": "## RNAfold parameter file v2.0\n",
              "no_header": "\n".join(l for l in lines if not l.startswith("##"))}
    i11 = lines.index("# int11")
    row = next(k for k in range(i11 + 1, len(lines)) if lines[k].strip() and not lines[k].strip().startswith("/*"))
    for tag, tok in (("badtoken", "x1y"), ("huge", "1e300"), ("nan", "nan"), ("inf", "-inf"), ("big_int", "99999999999999999999"), ("hex", "0x7fffffff")):
        bad = list(lines); bad[row] = bad[row].replace(bad[row].split()[0], tok, 1)
        corpus[tag] = "\n".join(bad)
    bad = list(lines); del bad[row + 2]; corpus["shortrow"] = "\n".join(bad)
    bad = list(lines); bad.insert(row + 1, "  1 2 3"); corpus["surplus"] = "\n".join(bad)
    last = max(k for k, l in enumerate(lines) if "*/" in l)
    corpus["comment"] = "\n".join(lines[:last + 2] + ["/* never closed"] + [l for l in lines[last + 2:] if "*/" not in l])
    corpus["misc_lxc"] = text.replace("107.856000", "1e999")
    corpus["long_special"] = text.replace("# Tetraloops\n", "# Tetraloops\n" + "G" * 100000 + " 1 2\n")
    corpus["many_special"] = text.replace("# Tetraloops\n", "# Tetraloops\n" + "GAAAAC 1 2\n" * 5000)
    corpus["oversized_block"] = text.replace("# stack\n", "# stack\n" + " ".join(["7"] * 2000000) + "\n")
    corpus["long_line"] = text + "\n# junk\n" + "9 " * 3000000
    corpus["many_sections"] = text.replace("# END", "\n".join(f"# s{k}\n{k}" for k in range(50000)) + "\n# END")
    corpus["crlf"] = text.replace("\n", "\r\n")
    corpus["nul_bytes"] = text[:5000] + "\0\0\0" + text[5000:]
    rng = np.random.default_rng(11)
    for k, cut in enumerate(sorted(rng.integers(0, len(text), size=150))):      # truncated files
        corpus[f"trunc{k:03d}"] = text[:int(cut)]
    raw = bytearray(text.encode())
    files = []
    for tag, body in corpus.items():
        f = tmp_path / f"c_{tag}.par"
        f.write_text(body)
        files.append(str(f))
    for k in range(150):                                                         # byte flips
        b = bytearray(raw)
        for pos in rng.integers(0, len(b), size=int(rng.integers(1, 8))):
            b[int(pos)] = int(rng.integers(0, 256))
        f = tmp_path / f"c_flip{k:03d}.par"
        f.write_bytes(bytes(b))
        files.append(str(f))
    out = subprocess.run([str(exe)] + files, capture_output=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    stdout, stderr = out.stdout.decode("latin-1"), out.stderr.decode("latin-1")      # (messages quote the offending bytes)
    assert out.returncode == 0, (stdout[-1500:], stderr[-3000:])
    verdicts = dict(zip([os.path.basename(f)[2:-4] for f in files], stdout.strip().split("\n")))
    assert len(verdicts) == len(files)
    assert verdicts["good_synthetic"].startswith("ok ") and verdicts["good_builtin"].startswith("ok ") and verdicts["crlf"].startswith("ok ")
    for tag in ("empty", "no_header", "badtoken", "huge", "nan", "inf", "big_int", "hex", "shortrow", "surplus", "comment", "misc_lxc", "long_special",
                "many_special", "oversized_block"):
        assert verdicts[tag].startswith("rejected: "), (tag, verdicts[tag])
    assert all(v.startswith(("ok ", "rejected: ")) for v in verdicts.values())
