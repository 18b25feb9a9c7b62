"""Load the (seq, struct, dcal) known-answer triples from the reference's
benchmark CSVs (container only) or from the committed fixture."""
import csv, gzip, os

REF = "/root/reference/benchmark_results/"
FILES = ["fft_100n_50ms_best_nrj_scores.csv", "fft_100n_50ms_scores.csv",
         "fft_200n_200ms_scores.csv", "mfe_scores.csv", "mxfold_scores.csv"]
FIXTURE = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "energy_kats.tsv.gz")


def load_from_reference():
    kats = {}
    for f in FILES:
        for r in csv.DictReader(open(REF + f)):
            k = (r["seq"], r["struct"])
            d = round(float(r["nrj"]) * 100)
            if k in kats:
                assert kats[k] == d
            else:
                kats[k] = d
    return [(s, st, d) for (s, st), d in kats.items()]


def load_fixture():
    out = []
    with gzip.open(FIXTURE, "rt") as fh:
        for line in fh:
            s, st, d = line.split()
            out.append((s, st, int(d)))
    return out


def write_fixture():
    ks = load_from_reference()
    ks.sort(key=lambda x: (len(x[0]), x[0], x[1]))
    with gzip.GzipFile(FIXTURE, "wb", mtime=0) as gz:
        for s, st, d in ks:
            gz.write(f"{s}\t{st}\t{d}\n".encode())
    return len(ks)


if __name__ == "__main__":
    print(write_fixture())
