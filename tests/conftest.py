import gzip
import json
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_json_gz(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def fold_cases():
    return load_json_gz("fold_traj.json.gz")


@pytest.fixture(scope="session")
def node_records():
    return load_json_gz("node_expand.json.gz")


@pytest.fixture(scope="session")
def energy_kats():
    out = []
    with gzip.open(os.path.join(GOLD, "energy_kats.tsv.gz"), "rt") as fh:
        for line in fh:
            s, st, d = line.split()
            out.append((s, st, int(d)))
    return out


@pytest.fixture(scope="session")
def bench_rows():
    out = []
    with gzip.open(os.path.join(GOLD, "bench_inputs.tsv.gz"), "rt") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            out.append(dict(name=f[0], seq=f[1], best=(f[2], int(f[3])), ppv=(f[4], int(f[5])), ppv200=(f[6], int(f[7])),
                            known=f[8], best_scores=(float(f[9]), float(f[10])), ppv_scores=(float(f[11]), float(f[12])),
                            ppv200_scores=(float(f[13]), float(f[14]))))
    return out


def long_fixture(suffix=""):
    """reference-Python runs above n = 2381, where scipy's convolve takes its fp64 FFT branch (tools/make_golden_long.py)"""
    t = load_json_gz(f"fold_traj_long{suffix}.json.gz")
    n = load_json_gz(f"node_expand_long{suffix}.json.gz")
    return t["sequences"], t["cases"], n["records"]


def long_record_args(seqs, r):
    """(sequence, dot-bracket, positions) of one region record of the long fixtures"""
    s = seqs[r["seq"]]
    db = ["."] * len(s)
    for a, b in r["db_pairs"]:
        db[a], db[b] = "(", ")"
    pos = [x for st, ln in r["pos"] for x in range(st, st + ln)]
    return s, "".join(db), pos


def long_record_agreement(ex, r):
    """how an exact expansion `ex` (oracle.expand_node / rafft_amd.rafft.expand_node) agrees with the reference's run
    through scipy's FFT: (same top-nb_mode SET, same ranked ORDER, same window_slide tuples on the common lags,
    same kept candidates in the same order)"""
    same_set = sorted(ex["lag"]) == sorted(r["lags"])
    same_order = ex["lag"] == r["lags"]
    wsg = {l: [a, b, c, d] for l, a, b, c, d in zip(ex["lag"], ex["nb"], ex["mi"], ex["mj"], ex["score"])}
    wsr = dict(zip(r["lags"], r["ws"]))
    same_ws = all(wsg[l] == wsr[l] for l in wsg if l in wsr)
    sol = [[ex["nb"][k], ex["score"][k], ex["mi"][k], ex["mj"][k], ex["ddcal"][k]] for k in ex["kept"]]
    return same_set, same_order, same_ws, sol == r["sol"]


# measured agreement of the EXACT correlation with the reference's scipy-FFT run, per fixture:
# (records, same set, same order, same window_slide on common lags, same kept candidates) - see DESIGN.md 2.2
LONG_AGREEMENT = {"": (55, 50, 1, 55, 52), "_ms50": (177, 159, 3, 177, 159)}
