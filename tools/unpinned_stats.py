"""How often does a fold depend on a table entry that no reference-held energy row pins?  

The built-in Turner tables (params/turner2004_fitted.json) mark every entry that one of the reference's 11 505
(sequence, structure, energy) rows exercises as `pinned`; the rest are rule/prior values (mostly 1x1 / 2x1 / 2x2
interior-loop entries).  The oracle - bit-identical to the GPU path on full beams (tests/test_gpu_configs.py) -
tracks every table look-up; this script folds BASELINE configs[1] and configs[2] with tracking on and reports

  dE_share           share of dE evaluations (rafft/rafft.py:98) whose VALUE involves an unpinned entry
                     (entries shared by the structure with and without the stem cancel)
  final_share        share of final-beam structures whose energy involves an unpinned entry
  best_share         share of sequences whose lowest-energy final structure does
  seq_any_dE         share of sequences with at least one such dE evaluation anywhere in the fold

and, for the benchmark set, the comparison with the reference's published lowest-energy structures
(fft_100n_50ms_best_nrj_scores.csv): for every sequence where ours differs - is the reference's structure in our
final beam, is its energy (our tables) lower/equal/higher than our best, do the two touch unpinned entries.

    python tools/unpinned_stats.py  -> profiles/r04_unpinned_lookups.json
"""
import gzip, json, os, sys, time
import multiprocessing as mp
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
FITTED = os.path.join(ROOT, "params", "turner2004_fitted.json")


def _init():
    import oracle
    oracle.set_pinned(FITTED)
    oracle.track(True)


def _one(task):
    import oracle
    seq, ref_best = task
    c = {}
    fin = oracle.fold(seq, 100, 50, 1000, counters=c)
    fu = c["final_unpinned"]
    best = min(range(len(fin)), key=lambda k: fin[k].dcal)
    out = dict(evals=c["evals"] - c["children"], dE_unp=c["dE_unpinned"], n_final=len(fin), final_unp=sum(1 for x in fu if x),
               best_unp=int(fu[best] > 0))
    if ref_best is not None:
        rdb, rd = ref_best
        out["ref_same"] = int(fin[best].str_struct == rdb)
        if not out["ref_same"]:
            d, nunp = oracle.eval_structure_tracked(seq, rdb)
            oracle.track(True)
            out.update(ref_in_beam=int(any(x.str_struct == rdb for x in fin)), ref_dcal_published=rd, ref_dcal_ours=d,
                       our_best_dcal=fin[best].dcal, ref_touches_unpinned=int(nunp > 0), L=len(seq))
    return out


def summarise(rows):
    ev = sum(r["evals"] for r in rows)
    return dict(sequences=len(rows), dE_evaluations=ev, dE_share=sum(r["dE_unp"] for r in rows) / ev,
                final_structures=sum(r["n_final"] for r in rows),
                final_share=sum(r["final_unp"] for r in rows) / sum(r["n_final"] for r in rows),
                best_share=sum(r["best_unp"] for r in rows) / len(rows),
                seq_any_dE=sum(1 for r in rows if r["dE_unp"]) / len(rows))


def main():
    rows = [l.rstrip("\n").split("\t") for l in gzip.open(os.path.join(ROOT, "tests", "golden", "bench_inputs.tsv.gz"), "rt")]
    cfg3 = [(r[1], (r[2], int(r[3]))) for r in rows]
    rng = np.random.default_rng(200)
    cfg2 = [("".join(rng.choice(list("ACGU"), 200)), None) for _ in range(1000)]
    out = {"params": "nb_mode 100, max_stack 50, max_branch 1000", "tables": "params/turner2004_fitted.json (built-in)"}
    with mp.get_context("fork").Pool(len(os.sched_getaffinity(0)), initializer=_init) as pool:
        for name, tasks in (("cfg2_random_L200", cfg2), ("cfg3_benchmark_set", cfg3)):
            t0 = time.time()
            order = sorted(range(len(tasks)), key=lambda i: -len(tasks[i][0]))
            res = pool.map(_one, [tasks[i] for i in order], chunksize=1)
            out[name] = summarise(res)
            print(name, out[name], f"{time.time() - t0:.0f} s", file=sys.stderr, flush=True)
            if name == "cfg3_benchmark_set":
                diff = [r for r in res if not r["ref_same"]]
                cmp_ = lambda r: "lower" if r["ref_dcal_ours"] < r["our_best_dcal"] else "equal" if r["ref_dcal_ours"] == r["our_best_dcal"] else "higher"
                tab = {}
                for r in diff:
                    key = (("in beam" if r["ref_in_beam"] else "not in beam"), cmp_(r),
                           "evaluates as published" if r["ref_dcal_ours"] == r["ref_dcal_published"] else "evaluates differently",
                           "unpinned involved" if (r["ref_touches_unpinned"] or r["best_unp"]) else "pinned only")
                    tab["; ".join(key)] = tab.get("; ".join(key), 0) + 1
                out["cfg3_vs_published_lowest_energy"] = dict(
                    identical=len(res) - len(diff), different=len(diff),
                    breakdown_of_different=dict(sorted(tab.items(), key=lambda kv: -kv[1])),
                    reference_structure_in_our_beam=sum(r["ref_in_beam"] for r in diff),
                    reference_structure_lower_than_our_best=sum(1 for r in diff if r["ref_dcal_ours"] < r["our_best_dcal"]),
                    reference_energy_reproduced=sum(1 for r in diff if r["ref_dcal_ours"] == r["ref_dcal_published"]),
                    any_unpinned_entry_involved=sum(1 for r in diff if r["ref_touches_unpinned"] or r["best_unp"]))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_unpinned_lookups.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
