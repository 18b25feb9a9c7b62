"""rocprofv3 passes of tools/profile_r02.sh -> profiles/: kernel statistics of the timed loop, per-launch HBM
traffic (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction as MI355X_MICROARCH.md prescribes) and the
instruction-issue roofline of every kernel from the SQ counters.

issue roofline: wave-instructions issued (VALU + SALU + LDS + SMEM + VMEM + branch) / (CUs * 4 SIMDs * clock * kernel
time) - the fraction of SIMD issue slots (one wave-instruction per SIMD per cycle) the kernel fills; beside it the
VALU-only figure priced at 2 cycles per wave64 VALU instruction on the SIMD-32 (MI355X_MICROARCH.md), and the shares
of wave cycles spent active / parked (s_waitcnt) / issue-stalled.

Lanes per vector instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 4: thread-cycles tick per quad-cycle of a wave64 instruction) and
LDS bank-conflict cycles come from a fifth pass when present.

usage: python tools/pmc_summary.py gpurun_out/r04_prof profiles r04      (run in the build container after tools/profile_r04.sh on the GPU box:
       the commit comes from git, the digest of rafft_amd/csrc from the tree that ran; a `cfg4` sub-directory - the configs[3] shard,
       3 calls of tools/ab_cfg4.py - becomes profiles/<tag>_cfg4_*)"""
import collections, csv, json, os, shutil, sys

N_CU, SIMD_PER_CU = 256, 4
N_BATCHES = 4          # bench.py --steps 3 --warmup 1 in every PMC pass (3 for the configs[3] shard: tools/ab_cfg4.py folds it three times)


def counters(dirname):
    f = os.path.join(dirname, "p_counter_collection.csv")
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, {k: len(v) for k, v in disp.items()}, dur


def head_commit():
    import subprocess
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except OSError:
        return None


def main(src, dst, tag, n_batches=N_BATCHES, digest=None, workload="BASELINE configs[2], bench.py --steps 3 --warmup 1 --no-extras with synchronous calls (BENCH_DEPTH=1, BENCH_PREWARM_S=0)"):
    global N_BATCHES
    N_BATCHES = n_batches
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
    if os.path.exists(os.path.join(src, "cu_share.json")):
        shutil.copy(os.path.join(src, "cu_share.json"), os.path.join(dst, f"{tag}_cu_share.json"))
    if os.path.exists(os.path.join(src, "trace_bench.json")):
        bench = json.load(open(os.path.join(src, "trace_bench.json")))
        json.dump(bench, open(os.path.join(dst, f"{tag}_trace_bench.json"), "w"), indent=1)
    if digest is None and os.path.exists(os.path.join(src, "csrc_digest.txt")):
        digest = open(os.path.join(src, "csrc_digest.txt")).read().strip()
    fe, nf, _ = counters(os.path.join(src, "pmc_fetch"))
    wr, nw, _ = counters(os.path.join(src, "pmc_write"))
    sq, ns, dur = counters(os.path.join(src, "pmc_sq"))
    sq2, ns2, dur2 = ({}, {}, {})
    if os.path.exists(os.path.join(src, "pmc_sq2", "p_counter_collection.csv")):
        sq2, ns2, dur2 = counters(os.path.join(src, "pmc_sq2"))
    lanes = {}
    if os.path.exists(os.path.join(src, "pmc_lanes", "p_counter_collection.csv")):
        ln, _, _ = counters(os.path.join(src, "pmc_lanes"))
        for k, c in ln.items():
            if c.get("SQ_INSTS_VALU"):
                lanes[k] = {"lanes_per_valu_instruction": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_INSTS_VALU"], 2),
                            "lds_bank_conflict_cycles": int(c.get("SQ_LDS_BANK_CONFLICT", 0)), "lds_idx_active_cycles": int(c.get("SQ_LDS_IDX_ACTIVE", 0)),
                            "lds_instructions": int(c.get("SQ_INSTS_LDS", 0))}
    clk_mhz = None
    for r in csv.DictReader(open(os.path.join(src, "trace", "t_agent_info.csv"))):
        if r.get("Agent_Type", "").upper() == "GPU" or r.get("Name", "").startswith("gfx"):
            for key in ("Max_Engine_Clk_Fcompute", "Max_Engine_Clk_FCompute", "Max_Clock_Frequency"):
                if r.get(key):
                    clk_mhz = float(r[key]); break
    clk = (clk_mhz or 2400.0) * 1e6
    kernels, issue = {}, {}
    for k in sorted(set(fe) | set(wr)):
        f = fe[k]["FETCH_SIZE"] / max(nf.get(k, 1), 1) if k in fe else 0.0
        w = wr[k]["WRITE_SIZE"] / max(nw.get(k, 1), 1) if k in wr else 0.0
        kernels[k] = {"launches_profiled": nf.get(k, 0), "hbm_bytes_per_batch": round((2 * f + w) * 1024 * nf.get(k, 0) / N_BATCHES), "fetch_bytes_per_launch": round(2 * f * 1024), "write_bytes_per_launch": round(w * 1024),
                      "hbm_bytes_per_launch": round((2 * f + w) * 1024), "raw_FETCH_SIZE_KiB": round(f, 3), "raw_WRITE_SIZE_KiB": round(w, 3)}
    for k, c in sq.items():
        t = dur[k] * 1e-9
        if t <= 0 or c["SQ_WAVE_CYCLES"] <= 0:
            continue
        extra = sq2.get(k, {})
        scale = (dur[k] / dur2[k]) if k in dur2 and dur2[k] > 0 else 1.0      # the second SQ pass is another run of the same work
        n_inst = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"] + scale * (
            extra.get("SQ_INSTS_SMEM", 0) + extra.get("SQ_INSTS_VMEM_RD", 0) + extra.get("SQ_INSTS_VMEM_WR", 0))
        slots = N_CU * SIMD_PER_CU * clk * t
        issue[k] = {"launches_profiled": ns[k], "kernel_seconds": round(t, 6),
                    "wave_instructions": int(n_inst), "issue_frac": round(n_inst / slots, 4),
                    "valu_busy_frac_at_2_cycles_per_wave64_op": round(2 * c["SQ_INSTS_VALU"] / slots, 4),
                    # measured pipe occupancy (second pass): SQ_ACTIVE_INST_* tick in quad-cycles while an instruction of that
                    # kind executes - for VALU it equals the instruction count, i.e. a vector instruction holds its SIMD 4 cycles
                    "pipe_busy_frac_measured": {n[len("SQ_ACTIVE_INST_"):].lower(): round(4.0 * scale * extra[n] / slots, 4)
                                                for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS") if n in extra},
                    "insts": {n: int(c[n]) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")},
                    "insts_second_pass": {n: int(v) for n, v in extra.items() if n.startswith("SQ_INSTS")},
                    "wave_cycle_shares": {"active": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                                          "parked_waitcnt_or_barrier": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                                          "issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)},
                    # SQ_WAVE_CYCLES (like SQ_BUSY_CYCLES and the SQ_WAIT_* counters) ticks once per 4 cycles
                    "mean_waves_resident_per_simd": round(4.0 * c["SQ_WAVE_CYCLES"] / slots, 3)}
    ex = next((k for k in issue if k.startswith("void expand_kernel<64")), None)
    out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* | SQ_* second set | lanes + LDS conflicts), each in its own run, "
                     "%s; tools/profile_%s.sh" % (workload, tag.split("_")[0]),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md)",
           "commit": head_commit(), "csrc_digest": digest, "batches_profiled": n_batches,
           "whole_batch_hbm_bytes": sum(v["hbm_bytes_per_batch"] for v in kernels.values()),
           "clock_hz_used": clk, "kernels": kernels, "issue": issue, "lanes_and_lds": lanes,
           "issue_roofline": dict(kernel=ex, **{k: issue[ex][k] for k in ("issue_frac", "valu_busy_frac_at_2_cycles_per_wave64_op", "pipe_busy_frac_measured",
                                                                             "wave_cycle_shares", "mean_waves_resident_per_simd")}) if ex else None}
    json.dump(out, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    for k, v in sorted(issue.items(), key=lambda kv: -kv[1]["kernel_seconds"])[:8]:
        print(k[:44].ljust(44), v["kernel_seconds"], v["issue_frac"], v["valu_busy_frac_at_2_cycles_per_wave64_op"], v["wave_cycle_shares"], v["mean_waves_resident_per_simd"])
    for k, v in kernels.items():
        if "expand" in k or "beam" in k or "material" in k:
            print(k[:44].ljust(44), v["hbm_bytes_per_launch"], v["launches_profiled"])


if __name__ == "__main__":
    src, dst, tag = sys.argv[1:4]
    main(src, dst, tag)
    c4 = os.path.join(src, "cfg4")
    if os.path.exists(os.path.join(c4, "pmc_fetch", "p_counter_collection.csv")):
        dg = open(os.path.join(src, "csrc_digest.txt")).read().strip() if os.path.exists(os.path.join(src, "csrc_digest.txt")) else None
        print("---- configs[3] shard")
        main(c4, dst, tag + "_cfg4", 3, dg, "BASELINE configs[3], one GPU's LPT shard (2048 sequences, L 100..3000, ms=200), 3 calls of tools/ab_cfg4.py")
        for name in ("trace_run.log",):
            if os.path.exists(os.path.join(c4, name)):
                shutil.copy(os.path.join(c4, name), os.path.join(dst, f"{tag}_cfg4_{name}"))
    if os.path.exists(os.path.join(src, "bench_cfg4_n1.json")) and os.path.getsize(os.path.join(src, "bench_cfg4_n1.json")):
        shutil.copy(os.path.join(src, "bench_cfg4_n1.json"), os.path.join(dst, f"{tag}_bench_cfg4_n1.json"))
