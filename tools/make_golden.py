"""Generate golden vectors by importing the reference's own Python (container only).

The reference (rafft/utils.py:7) imports ViennaRNA's `RNA` module at import time;
ViennaRNA is not installed here (ordinary ModuleNotFoundError, no permission
denial - SURVEY.md 8c).  A stand-in `RNA` module is injected whose
`fold_compound(seq, md).eval_structure(db)` returns the restated Turner-2004
energy (oracle/rafft_oracle.c:eval_pt, itself pinned by the reference's 11 505
energy triples).  Everything else - encode, correlation (scipy), lag ranking,
window_slide, candidate filtering/sorting, node splitting, beam/BFS control
flow, dedupe, max_branch quirk - is executed by the REFERENCE code.  The
resulting vectors therefore pin the oracle's control flow to the reference and
its energies to the KAT-pinned evaluator.

Outputs (tests/golden/):
  fold_traj.json.gz     full trajectories for a set of (sequence, params)
  node_expand.json.gz   per-node records: correlation profile, ranked lags,
                        window_slide tuples, kept candidates
  example_rafft.out, example_rafft_20.out   copies of the reference's example
                        output data files (expected outputs)
  bench_inputs.tsv.gz   name, sequence of benchmark_cleaned_all_length.csv and the
                        reference's published result rows for it
"""
import sys, os, types, json, gzip, csv, shutil
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import oracle as ORC  # noqa: E402

# ---- stand-in RNA module -------------------------------------------------
RNA = types.ModuleType("RNA")


class md:  # noqa: N801
    temperature = 37.0


class _FC:
    def __init__(self, seq, model):
        self.seq = seq
        assert abs(model.temperature - 37.0) < 1e-9

    def eval_structure(self, db):
        d = ORC.eval_structure(self.seq, db)
        return float(np.float32(np.float32(d) / 100.0))


RNA.md = md
RNA.fold_compound = lambda seq, model: _FC(seq, model)
sys.modules["RNA"] = RNA
sys.path.insert(0, "/root/reference")
import rafft.rafft as R  # noqa: E402
import rafft.utils as U  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
records = []
_budget = [0]
_orig_create = R.create_childs


def rec_create_childs(upair, cur_str, gp):
    n = len(upair.pos_list)
    noncontig = any(upair.pos_list[i + 1] - upair.pos_list[i] != 1 for i in range(n - 1))
    if _budget[0] > 0 and (noncontig or _budget[0] % 3 == 0) and n >= 2:
        cor_l = U.auto_cor(upair.forward, upair.backward)
        cor = [float(c) for _, c in cor_l]
        cs = sorted(cor_l, key=lambda el: el[1])
        ranked = cs[::-1][:gp.nb_mode]
        ws = [R.window_slide(upair.forward, upair.backward, pos, upair.pos_list, gp.min_hp) for pos, _ in ranked]
        sol = R.find_best_consecutives(cs, upair, cur_str, gp)
        records.append(dict(
            seq=gp.sequence, db=cur_str.str_struct, pos=list(map(int, upair.pos_list)),
            nb_mode=gp.nb_mode, min_hp=gp.min_hp, min_nrj=gp.min_nrj, gc=gp.gc_wei, au=gp.au_wei, gu=gp.gu_wei,
            cor=cor, lags=[int(p) for p, _ in ranked],
            ws=[[int(a), int(b), int(c), float(d)] for a, b, c, d in ws],
            sol=[[int(s[0]), float(s[1]), int(s[2]), int(s[3]), int(round(s[4] * 100))] for s in sol]))
    if _budget[0] > 0:
        _budget[0] -= 1
    return _orig_create(upair, cur_str, gp)


R.create_childs = rec_create_childs


def run(seq, rec_nodes=0, **kw):
    _budget[0] = rec_nodes
    fin, traj = R.fold(seq, traj=True, **kw)
    return dict(seq=seq, params=kw,
                traj=[[[s.str_struct, int(round(float(s.energy) * 100))] for s in st] for st in traj])


def main():
    rng = np.random.default_rng(20241220)
    bench = list(csv.reader(open("/root/reference/benchmark_results/benchmark_cleaned_all_length.csv")))
    ex = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"
    trna = bench[11][0]
    cases = []
    cases.append(run(ex, 40, nb_mode=100, max_stack=5, max_branch=1000))
    cases.append(run(ex, 40, nb_mode=100, max_stack=20, max_branch=1000))
    cases.append(run(ex, 0, nb_mode=100, max_stack=1, max_branch=100))
    cases.append(run(ex, 0, nb_mode=10, max_stack=50, max_branch=1000))
    cases.append(run(ex, 0, nb_mode=100, max_stack=50, max_branch=7))      # max_branch quirk (vi)
    cases.append(run(ex, 0, nb_mode=100, max_stack=10, max_branch=1))
    cases.append(run(ex, 20, nb_mode=100, max_stack=10, max_branch=1000, min_hp=5))
    cases.append(run(ex, 20, nb_mode=100, max_stack=10, max_branch=1000, min_nrj=-2.5))
    cases.append(run(ex, 20, nb_mode=100, max_stack=10, max_branch=1000, gc_wei=1.0, au_wei=1.0, gu_wei=1.0))
    cases.append(run(ex, 20, nb_mode=50, max_stack=10, max_branch=1000, gc_wei=3.0, au_wei=2.0, gu_wei=0.0))
    cases.append(run(trna, 30, nb_mode=100, max_stack=1, max_branch=1000))   # BASELINE cfg1
    cases.append(run(trna, 30, nb_mode=100, max_stack=50, max_branch=1000))
    # tiny / edge sequences
    for s in ["A", "AU", "GC", "GGGAAACCC", "GGGGAAAACCCC", "ACGUACGUACGU", "GGGGGGGGGG", "GCGCGCGCGCGCGC",
              "GGGNNNNCCC", "NNNNN", "GGGAAAUCCCGGGAAAUCCC", "AUAUAUAUAUAUAUAUAUAU"]:
        cases.append(run(s, 5, nb_mode=100, max_stack=5, max_branch=1000))
    # benchmark sequences (short ones at the headline config, some longer at smaller beams)
    idx = sorted(range(len(bench)), key=lambda i: len(bench[i][0]))
    for i in idx[:6] + idx[200:204] + idx[1000:1003]:
        cases.append(run(bench[i][0], 10, nb_mode=100, max_stack=50, max_branch=1000))
    for i in idx[1800:1802] + idx[2200:2201]:
        cases.append(run(bench[i][0], 10, nb_mode=100, max_stack=10, max_branch=1000))
    # random sequences (cfg2-like, reduced count) and with N
    for L in (30, 60, 100, 200):
        for _ in range(3):
            s = "".join(rng.choice(list("ACGU"), L))
            cases.append(run(s, 8, nb_mode=100, max_stack=50 if L <= 100 else 20, max_branch=1000))
    s = "".join(rng.choice(list("ACGUN"), 80, p=[.24, .24, .24, .24, .04]))
    cases.append(run(s, 8, nb_mode=100, max_stack=20, max_branch=1000))
    s = "".join(rng.choice(list("ACGU"), 400))
    cases.append(run(s, 10, nb_mode=100, max_stack=5, max_branch=1000))

    with gzip.GzipFile(os.path.join(GOLD, "fold_traj.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(cases, separators=(",", ":")).encode())
    with gzip.GzipFile(os.path.join(GOLD, "node_expand.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(records, separators=(",", ":")).encode())
    print(len(cases), "fold cases;", len(records), "node records")

    shutil.copy("/root/reference/example/rafft.out", os.path.join(GOLD, "example_rafft.out"))
    shutil.copy("/root/reference/example/rafft_20.out", os.path.join(GOLD, "example_rafft_20.out"))

    bench_fixture(bench)


def bench_fixture(bench=None):
    """tests/golden/bench_inputs.tsv.gz: name, sequence, the reference's published result rows
    (structure, dcal) for the three RAFFT runs, then the known structure and the published
    (pvv, sens) of those rows (scoring.py:121-128 columns)."""
    if bench is None:
        bench = list(csv.reader(open("/root/reference/benchmark_results/benchmark_cleaned_all_length.csv")))

    def rows(f):
        return {r["seq"]: (r["struct"], round(float(r["nrj"]) * 100), r["pvv"], r["sens"]) for r in
                csv.DictReader(open("/root/reference/benchmark_results/" + f))}
    best = rows("fft_100n_50ms_best_nrj_scores.csv")
    ppv = rows("fft_100n_50ms_scores.csv")
    ppv200 = rows("fft_200n_200ms_scores.csv")
    with gzip.GzipFile(os.path.join(GOLD, "bench_inputs.tsv.gz"), "wb", mtime=0) as fh:
        for seq, known, name in bench:
            b, p, q = best[seq], ppv[seq], ppv200[seq]
            fh.write(f"{name}\t{seq}\t{b[0]}\t{b[1]}\t{p[0]}\t{p[1]}\t{q[0]}\t{q[1]}\t{known}\t{b[2]}\t{b[3]}\t{p[2]}\t{p[3]}\t{q[2]}\t{q[3]}\n".encode())


if __name__ == "__main__":
    main()


def kinetics_golden():
    """tests/golden/kinetics.json.gz: the reference's rafft_kin.kinetics on its own example outputs."""
    import matplotlib
    matplotlib.use("Agg")
    from rafft.rafft_kin import kinetics
    out = {}
    for name, mt, ns in (("example_rafft_20.out", 40, 100), ("example_rafft.out", 30, 50)):
        fp, seq = U.parse_rafft_output(os.path.join(GOLD, name))
        traj, times, sl, eq = kinetics(fp, mt, ns)
        out[name] = dict(max_time=mt, n_steps=ns, times=[float(t) for t in times],
                         trajectory=[[float(x) for x in row] for row in traj],
                         struct_list=[s.str_struct for s in sl],
                         equi=[[e[0], float(e[1]), float(e[2]), int(e[3])] for e in eq])
    with gzip.GzipFile(os.path.join(GOLD, "kinetics.json.gz"), "wb", mtime=0) as fh:
        fh.write(json.dumps(out, separators=(",", ":")).encode())


if __name__ == "__main__" and "--kinetics" in sys.argv:
    kinetics_golden()


if __name__ == "__main__" and "--bench-fixture" in sys.argv:
    bench_fixture()
