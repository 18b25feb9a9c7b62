"""Independent Python reader/writer of ViennaRNA 2.x parameter files and of ViennaRNA's temperature
rescaling (test infrastructure: the checker of rafft_amd/csrc/rafft_params.h, which is the product's).

Format ("## RNAfold parameter file v2.0", what RNA.params_save writes / misc/rna_turner2004.par holds):
sections `# name`, C comments, INF / DEF tokens; array sections are row-major over
  stack 7x7 (pairs CG GC GU UG AU UA NN), mismatch_* 7x5x5 (pair, N A C G U, N A C G U), dangle5/3 7x5,
  int11 7x7x5x5, int21 7x7x5x5x5, int22 6x6x4x4x4x4 (pairs without NN, bases without N), hairpin/bulge/interior 31,
each followed by a `_enthalpies` twin; NINIO (m, m_dH, max), ML_params (cu cu_dH cc cc_dH ci ci_dH),
Misc (DuplexInit dH TerminalAU dH LXC 0), Hexaloops/Tetraloops/Triloops (sequence energy enthalpy).

`tables_at(par, temp)` returns the tables in the oracle's layout (pair types 0..6, base codes 0..4) rescaled as
ViennaRNA's get_scaled_params does for model details dangles=2: G(T) = dH - (dH - G37) * (T + 273.15) / 310.15
truncated toward zero; dangles and multi/exterior mismatches clipped to <= 0; lxc * (T + 273.15) / 310.15."""
import re

import numpy as np

INF = 10000000
SHAPES = {"stack": (7, 7), "mismatch_hairpin": (7, 5, 5), "mismatch_interior": (7, 5, 5), "mismatch_interior_1n": (7, 5, 5),
          "mismatch_interior_23": (7, 5, 5), "mismatch_multi": (7, 5, 5), "mismatch_exterior": (7, 5, 5),
          "dangle5": (7, 5), "dangle3": (7, 5), "int11": (7, 7, 5, 5), "int21": (7, 7, 5, 5, 5),
          "int22": (6, 6, 4, 4, 4, 4), "hairpin": (31,), "bulge": (31,), "interior": (31,)}
SPECIAL = {"Triloops": 5, "Tetraloops": 6, "Hexaloops": 8}


def read_par(path):
    txt = open(path).read()
    assert "## RNAfold parameter file v2.0" in txt
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    secs, cur = {}, None
    for line in txt.splitlines():
        line = line.strip()
        if not line or line.startswith("##"):
            continue
        if line.startswith("#"):
            name = line[1:].split()[0]
            if name == "END":
                break
            cur = secs.setdefault(name, [])
            continue
        if cur is not None:
            cur.extend(line.split())
    par = {}
    for base, shape in SHAPES.items():
        for sfx in ("", "_enthalpies"):
            if base + sfx not in secs:
                continue
            vals = [INF if t == "INF" else int(float(t)) for t in secs[base + sfx]]
            par[base + sfx] = np.array(vals, dtype=np.int64).reshape(shape)
    t = secs["NINIO"]
    par["ninio"], par["ninio_dH"], par["max_ninio"] = int(t[0]), int(t[1]), int(t[2])
    t = [int(x) for x in secs["ML_params"][:6]]
    par["ml_base"], par["ml_base_dH"], par["ml_closing"], par["ml_closing_dH"], par["ml_intern"], par["ml_intern_dH"] = t
    t = secs["Misc"]
    par["duplex_init"], par["duplex_init_dH"], par["terminal_au"], par["terminal_au_dH"] = (int(x) for x in t[:4])
    par["lxc"] = float(t[4])
    for name, ln in SPECIAL.items():
        t = secs.get(name, [])
        par[name] = [(t[i], int(t[i + 1]), int(t[i + 2])) for i in range(0, len(t), 3)]
        assert all(len(s) == ln for s, _, _ in par[name])
    return par


def write_par(par, path, comment="synthetic"):
    """Writes `par` (the dict read_par returns) in the same format."""
    pn = ["CG", "GC", "GU", "UG", "AU", "UA", "NN"]
    out = ["## RNAfold parameter file v2.0", "", f"/* {comment} */"]

    def rows(a):
        a = np.asarray(a)
        flat = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(-1, 10 if a.size % 10 == 0 else 1)
        return ["".join("   INF" if v >= INF else f"{int(v):6d}" for v in r) for r in flat]

    for base in SHAPES:
        for sfx in ("", "_enthalpies"):
            if base + sfx not in par:
                continue
            out += ["", f"# {base}{sfx}", f"/* {' '.join(pn)} */"]
            a = par[base + sfx]
            if a.ndim == 1:
                out += ["".join("   INF" if v >= INF else f"{int(v):6d}" for v in a[i:i + 10]) for i in range(0, 31, 10)]
            else:
                out += rows(a)
    out += ["", "# NINIO", "/* Ninio = MIN(max, m*|n1-n2| */", f"{par['ninio']:6d} {par['ninio_dH']:6d} {par['max_ninio']:6d}"]
    out += ["", "# ML_params", "/* cu cu_dH cc cc_dH ci ci_dH */",
            " ".join(f"{par[k]:6d}" for k in ("ml_base", "ml_base_dH", "ml_closing", "ml_closing_dH", "ml_intern", "ml_intern_dH"))]
    out += ["", "# Misc", "/* all parameters are pairs of 'energy enthalpy' */",
            f"{par['duplex_init']:6d} {par['duplex_init_dH']:6d} {par['terminal_au']:6d} {par['terminal_au_dH']:6d} {par['lxc']:12.6f} {0:6d}"]
    for name in ("Hexaloops", "Tetraloops", "Triloops"):
        out += ["", f"# {name}"] + [f"{s} {e:6d} {h:6d}" for s, e, h in par[name]]
    out += ["", "# END", ""]
    open(path, "w").write("\n".join(out))


def add_synthetic_enthalpies(par, seed=1):
    """dH for every energy: a deterministic made-up value (mechanism tests only - these are NOT thermodynamic data)."""
    rng = np.random.default_rng(seed)
    out = dict(par)
    for base in SHAPES:
        g = np.asarray(par[base])
        dh = 3 * g + 10 * rng.integers(-40, 41, size=g.shape)
        dh[g >= INF] = INF
        out[base + "_enthalpies"] = dh
    for k in ("ninio", "ml_base", "ml_closing", "ml_intern", "terminal_au", "duplex_init"):
        out[k + "_dH"] = int(3 * par[k] + 10 * rng.integers(-40, 41))
    for name in SPECIAL:
        out[name] = [(s, e, int(3 * e + 10 * rng.integers(-40, 41))) for s, e, _ in par[name]]
    return out


def _rescale(g, dh, tempf):
    g = np.asarray(g, dtype=np.float64)
    dh = np.asarray(dh, dtype=np.float64)
    r = np.trunc(dh - (dh - g) * tempf).astype(np.int64)
    return np.where(np.asarray(g) >= INF, INF, r)


def _expand_int22(a):
    """6x6x4x4x4x4 -> 7x7x5x5x5x5 in oracle layout (pair 0 unused; base 0 = N: maximum over the concrete bases)."""
    full = np.zeros((7, 7, 5, 5, 5, 5), dtype=np.int64)
    full[1:, 1:, 1:, 1:, 1:, 1:] = a
    for ax in (2, 3, 4, 5):
        idx = [slice(None)] * 6
        idx[ax] = slice(1, 5)
        mx = full[tuple(idx)].max(axis=ax)
        idx[ax] = 0
        full[tuple(idx)] = mx
    full[0] = 0
    full[:, 0] = 0
    return full


def tables_at(par, temp):
    at37 = temp == 37.0
    tempf = (temp + 273.15) / (37.0 + 273.15)

    def sc(name):
        g = np.asarray(par[name])
        if at37:
            return g.copy()
        return _rescale(g, par[name + "_enthalpies"], tempf)

    def scs(k):
        return int(par[k]) if at37 else int(_rescale(par[k], par[k + "_dH"], tempf))

    T = {}
    st = np.zeros((7, 7), dtype=np.int64)
    st[1:, 1:] = sc("stack")[:6, :6]
    T["stack"] = st
    for k in ("hairpin", "bulge", "interior"):
        T[k] = sc(k)
    for k in ("mismatch_hairpin", "mismatch_interior", "mismatch_interior_1n", "mismatch_interior_23", "mismatch_multi",
              "mismatch_exterior"):
        a = np.zeros((7, 5, 5), dtype=np.int64)
        a[1:] = sc(k)[:6]
        if k in ("mismatch_multi", "mismatch_exterior"):
            a = np.minimum(a, 0)
        T[k] = np.minimum(a, 30000)
    for k in ("dangle5", "dangle3"):
        a = np.zeros((7, 5), dtype=np.int64)
        a[1:] = sc(k)[:6]
        T[k] = np.minimum(a, 0)
    a = np.zeros((7, 7, 5, 5), dtype=np.int64)
    a[1:, 1:] = sc("int11")[:6, :6]
    T["int11"] = np.minimum(a, 30000)
    a = np.zeros((7, 7, 5, 5, 5), dtype=np.int64)
    a[1:, 1:] = sc("int21")[:6, :6]
    T["int21"] = np.minimum(a, 30000)
    g22 = _expand_int22(np.asarray(par["int22"]))
    if at37:
        T["int22"] = g22
    else:
        T["int22"] = _rescale(g22, _expand_int22(np.asarray(par["int22_enthalpies"])), tempf)
        T["int22"][0] = 0
        T["int22"][:, 0] = 0
    T["int22"] = np.minimum(T["int22"], 30000)
    T["scalars"] = dict(ml_base=scs("ml_base"), ml_closing=scs("ml_closing"), ml_intern=scs("ml_intern"), ninio=scs("ninio"),
                        max_ninio=int(par["max_ninio"]), term_au=scs("terminal_au"),
                        lxc=float(par["lxc"]) if at37 else float(par["lxc"]) * tempf)
    T["special"] = {}
    for kind, name in enumerate(("Triloops", "Tetraloops", "Hexaloops")):
        seen, ent = set(), []
        for s, e, h in par[name]:
            if s in seen:
                continue
            seen.add(s)
            ent.append((s, int(e) if at37 else int(_rescale(e, h, tempf))))
        T["special"][kind] = ent
    return T
