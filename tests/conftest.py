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
