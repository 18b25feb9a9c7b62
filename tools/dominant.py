"""Which kernel is "dominant" depends on the clock (VERDICT r4 #3b).  Three of them side by side, per kernel, from the rocprofv3 passes
of tools/profile_r05.sh:
  pipelined_summed_ms   sum of the durations in the kernel trace of the driver's command (a kernel that queues for CUs behind the
                        persistent workgroups of another wave is charged for the wait)
  cu_weighted_ms        the same, every dispatch weighted with the share of the chip its grid can occupy (tools/cu_share.py)
  serial_ms_per_batch   a trace with every kernel alone on the chip (RAFFT_SERIAL=1, one wave at a time), per benchmark batch
usage: python tools/dominant.py gpurun_out/r05_prof profiles r05   -> profiles/r05_dominant.json (bench.py prints it as roofline.dominant_by)"""
import csv, json, os, sys


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"].split("(")[0].replace("void ", "")] = (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]))
    return out


def main(src, dst, tag, serial_batches=60, pipelined_batches=25):
    pip = stats(os.path.join(src, "trace", "t_kernel_stats.csv"))
    ser = stats(os.path.join(src, "serial", "t_kernel_stats.csv")) if os.path.exists(os.path.join(src, "serial", "t_kernel_stats.csv")) else {}
    cu = {}
    if os.path.exists(os.path.join(src, "cu_share.json")):
        for k, v in json.load(open(os.path.join(src, "cu_share.json")))["kernels"].items():
            cu[k] = v["cu_weighted_ms"]
    ks = {}
    for k in sorted(set(pip) | set(ser), key=lambda k: -(ser.get(k, (0, 0))[0])):
        if k.startswith("__amd"):
            continue
        ks[k] = {"pipelined_summed_ms": round(pip.get(k, (0, 0))[0], 3), "pipelined_calls": pip.get(k, (0, 0))[1],
                 "pipelined_ms_per_batch": round(pip.get(k, (0, 0))[0] / pipelined_batches, 4),
                 "cu_weighted_ms": round(cu.get(k, 0.0), 3),
                 "serial_ms_per_batch": round(ser.get(k, (0, 0))[0] / serial_batches, 4), "serial_calls": ser.get(k, (0, 0))[1]}
    top = lambda f: max(ks, key=lambda k: ks[k][f]) if ks else None
    out = {"source": f"tools/profile_{tag}.sh: kernel trace of `bench.py --steps 20 --warmup 5 --no-extras` without the pre-warm ({pipelined_batches} batches), "
                     f"tools/cu_share.py on it, and a serial trace (RAFFT_SERIAL=1 RAFFT_SPLIT=0, tools/ab_bench.py 20: {serial_batches} batches)",
           "top_pipelined_summed": top("pipelined_summed_ms"), "top_cu_weighted": top("cu_weighted_ms"), "top_serial": top("serial_ms_per_batch"),
           "kernels": ks}
    json.dump(out, open(os.path.join(dst, f"{tag}_dominant.json"), "w"), indent=1)
    if ser:
        import shutil
        shutil.copy(os.path.join(src, "serial", "t_kernel_stats.csv"), os.path.join(dst, f"{tag}_serial_kernel_stats.csv"))
    print(json.dumps({k: out[k] for k in ("top_pipelined_summed", "top_cu_weighted", "top_serial")}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
