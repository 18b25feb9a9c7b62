"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic of each kernel.

Units/corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read, so it is doubled;
WRITE_SIZE is taken as is.  Our kernels gather 1-2-byte elements, an access width the guide
calls uncalibrated, so the absolute figure is indicative.

usage: python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic.json
"""
import csv, glob, json, sys, collections


def per_kernel(dirname, counter):
    f = glob.glob(dirname + "/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main(fetch_dir, write_dir, out):
    fe = per_kernel(fetch_dir, "FETCH_SIZE")
    wr = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0)); w, nw = wr.get(k, (0.0, 0))
        res[k] = {"launches_profiled": nf, "fetch_bytes_per_launch": round(2 * f * 1024), "write_bytes_per_launch": round(w * 1024),
                  "hbm_bytes_per_launch": round((2 * f + w) * 1024), "raw_FETCH_SIZE_KiB": round(f, 3), "raw_WRITE_SIZE_KiB": round(w, 3)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1",
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main(*sys.argv[1:4])
