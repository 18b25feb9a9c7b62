"""CU-weighted shares from a rocprofv3 kernel trace (t_kernel_trace.csv).

A sum of kernel durations overstates the latency-bound launches of the long-tail wave: `beam_step_kernel<1024>` on the 8 longest
sequences is 8 workgroups on 256 CUs, running BESIDE the bulk wave's kernels.  Each dispatch is weighted with the share of the
chip its grid can occupy at most: min(1, workgroups / CUs) (an upper bound: workgroups of sequences that are finished exit at
once).  usage: python tools/cu_share.py t_kernel_trace.csv [n_cus] > out.json"""
import csv, json, re, sys, collections
n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dur = collections.defaultdict(float); w = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    wgs = 1
    for ax in "XYZ":
        wgs *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
    dur[name] += d; w[name] += d * min(1.0, wgs / n_cu); calls[name] += 1
td, tw = sum(dur.values()), sum(w.values())
out = {"source": "rocprofv3 --kernel-trace of bench.py --steps 20 --warmup 5; weight of a dispatch = min(1, workgroups / %d CUs)" % n_cu,
       "kernels": {k: {"calls": calls[k], "ms": round(dur[k] / 1e6, 3), "share_of_summed_durations": round(dur[k] / td, 4),
                       "cu_weighted_ms": round(w[k] / 1e6, 3), "cu_weighted_share": round(w[k] / tw, 4)}
                   for k in sorted(dur, key=lambda k: -dur[k])}}
print(json.dumps(out, indent=1))
