#!/usr/bin/env python3
"""bench.py - sequences/second of the RAFFT fold hot path on MI355X.

Workload (BASELINE.json configs[2], the config the metric is quoted on and the
reference's own published run, benchmark_results/bench_fft.py:8): the 2296
sequences of benchmark_cleaned_all_length.csv, nb_mode n=100, max_stack ms=50,
max_branch=1000 (CLI default).  One "step" = one pass of the whole hot path
(rafft_fold_batch through the C-ABI) over that batch.  With N ranks every rank
folds its own replica of the batch (independent sequences, no collective):
weak scaling, value = N * 2296 * K / max-over-ranks time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel
(expand_kernel): algorithmic bytes (per region 3n + 16*min(K,2n-1),
per structure 3L) per launch / mean launch duration measured with HIP events on
the library's stream.  `cpu_baseline` times the CPU oracle (oracle/rafft_oracle.c,
a port of the reference algorithm) on a bounded sample with one process per core.
"""
import argparse
import ctypes as C
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def load_bench_sequences():
    seqs = []
    with gzip.open(os.path.join(ROOT, "tests", "golden", "bench_inputs.tsv.gz"), "rt") as fh:
        for line in fh:
            seqs.append(line.split("\t")[1])
    return seqs


def _cpu_worker(args):
    import oracle
    seq, n, ms, mb = args
    fin = oracle.fold(seq, n, ms, mb)
    return [(x.str_struct, x.dcal) for x in fin]


def cpu_baseline(seqs, n, ms, mb, budget_s=20.0):
    """Oracle (kind=port) on the host cores, one worker process per core
    (mirrors benchmark_results/bench_fft.py:17-21), on a bounded sample."""
    import multiprocessing as mp
    import oracle
    oracle.oracle.build()
    # the GPU box exposes many logical CPUs but a job owns a 16-core share: never oversubscribe it
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    # sample: every k-th sequence of the same workload, sized from a short probe
    t0 = time.time()
    probe = seqs[::97][:16]
    for s in probe:
        oracle.fold(s, n, ms, mb)
    per_seq = (time.time() - t0) / len(probe)
    want = int(max(cores * 4, min(len(seqs), budget_s * cores / max(per_seq, 1e-6))))
    stride = max(1, len(seqs) // want)
    sample = seqs[::stride]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        t0 = time.time()
        finals = pool.map(_cpu_worker, [(s, n, ms, mb) for s in sample], chunksize=4)
        el = time.time() - t0
    return {"value": round(len(sample) / el, 2), "unit": "sequences/s", "cores": cores, "kind": "port",
            "sample": f"every {stride}-th sequence of the workload ({len(sample)} seqs, {el:.1f} s wall), "
                      f"oracle/rafft_oracle.c, Pool({cores})"}, dict(zip(sample, finals))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nb-mode", type=int, default=100)
    ap.add_argument("--max-stack", type=int, default=50)
    ap.add_argument("--max-branch", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_SAME_GPU"):      # rehearsal of the N>1 path on a one-GPU box
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CPU baseline first, on rank 0 at N=1 only, BEFORE this process touches the GPU
    # (fork-based pool; the timed GPU region below is unaffected)
    seqs = load_bench_sequences()
    cpu, cpu_finals = None, {}
    if world == 1 and not args.no_cpu_baseline:
        cpu, cpu_finals = cpu_baseline(seqs, args.nb_mode, args.max_stack, args.max_branch)

    import torch
    import torch.distributed as dist
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL ("nccl") carries only the barrier and the max-reduction of the elapsed time: the fold itself
        # needs no collective.  BENCH_BACKEND=gloo lets the same path be rehearsed without RCCL.
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    from rafft_amd import _native as N
    from rafft_amd.rafft import _params
    lib = N.lib()
    N.check(lib.rafft_init(local_rank))
    p = _params(args.nb_mode, args.max_stack, args.max_branch, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
    n = len(seqs)
    enc = [s.encode() for s in seqs]
    arr = (C.c_char_p * n)(*enc)
    lens = (C.c_int * n)(*[len(e) for e in enc])

    def step():
        res = C.POINTER(N.Result)()
        N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, local_rank, C.byref(res)))
        ok = all(res.contents.seq[i].status == 0 for i in range(0, n, 97))
        lib.rafft_free_result(res)
        assert ok

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # parity of the measured path against the CPU oracle on the baseline's sample (outside the timed region):
    # the WHOLE final beam (every structure, in order, with its exact dcal) identical; energy MAE in kcal/mol over
    # all beam rows (the metric's second half)
    parity = None
    if cpu_finals:
        res = C.POINTER(N.Result)()
        N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, local_rank, C.byref(res)))
        same, same_beam, abs_err, cnt, rows = 0, 0, 0.0, 0, 0
        for i, s in enumerate(seqs):
            if s not in cpu_finals:
                continue
            sr = res.contents.seq[i]
            w = sr.length + 1
            raw = C.string_at(sr.db, sr.n_structs * w).decode()
            beam = [(raw[k * w:k * w + sr.length], sr.dcal[k]) for k in range(sr.n_structs)]
            want = cpu_finals[s]
            same += int(beam[0] == want[0])
            same_beam += int(beam == want)
            for (_, gd), (_, wd) in zip(beam, want):
                abs_err += abs(gd - wd) / 100.0
                rows += 1
            cnt += 1
        lib.rafft_free_result(res)
        parity = {"sequences_compared": cnt, "final_beam_identical": same_beam, "beam_rows_compared": rows,
                  "lowest_energy_structure_identical": same,
                  "energy_mae_kcal_per_mol": abs_err / max(rows, 1)}

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    agg = {}
    for _ in range(args.steps):
        step()
        st = N.Stats()
        lib.rafft_get_stats(C.byref(st))
        for k, v in st.as_dict().items():
            agg[k] = agg.get(k, 0) + v
    barrier()
    el = time.perf_counter() - t0
    # per-stage kernel times: one extra call, outside the timed region, with every stage bracketed by HIP events
    # (the timed steps carry events only around the dominant kernel - a pair around every kernel costs ~1.3 ms)
    stage_ms = None
    if rank == 0:
        os.environ["RAFFT_SPANS"] = "2"
        res = C.POINTER(N.Result)()
        N.check(lib.rafft_fold_batch(C.byref(p), n, arr, lens, local_rank, C.byref(res)))
        lib.rafft_free_result(res)
        del os.environ["RAFFT_SPANS"]
        st = N.Stats()
        lib.rafft_get_stats(C.byref(st))
        stage_ms = {k: round(v, 3) for k, v in st.as_dict().items() if k.startswith("ms_")}
    # informational, outside the timed region: the same call on a 4x larger batch (the set replicated 4 times in
    # ONE rafft_fold_batch call).  A batch advances in lock-step folding steps whose number is set by its longest
    # sequence, so the fixed per-step latency is amortised over more sequences.  Never part of `value`.
    scaling_info = None
    if world == 1 and not args.no_cpu_baseline:
        R = 4
        arr4 = (C.c_char_p * (n * R))(*(enc * R))
        lens4 = (C.c_int * (n * R))(*([len(e) for e in enc] * R))
        best = None
        for _ in range(3):
            res = C.POINTER(N.Result)()
            t1 = time.perf_counter()
            N.check(lib.rafft_fold_batch(C.byref(p), n * R, arr4, lens4, local_rank, C.byref(res)))
            dt = time.perf_counter() - t1
            lib.rafft_free_result(res)
            best = dt if best is None else min(best, dt)
        scaling_info = {"replicas_in_one_call": R, "sequences": n * R, "ms": round(best * 1e3, 3),
                        "sequences_per_s": round(n * R / best, 1)}
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
        # separate runs, gfx950 correction applied by tools/pmc_traffic.py); counters cannot be read in-process.
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            k = tj["kernels"].get("void expand_kernel<64, false>") or tj["kernels"].get("void expand_kernel<64>")
            if k:
                traffic, traffic_src = k["hbm_bytes_per_launch"], "profiles/r01_traffic.json: " + tj["correction"]
        launches = max(1, agg["n_expand_launches"])
        dur_s = agg["ms_expand"] / 1e3 / launches
        bytes_per_launch = agg["alg_bytes_expand"] / launches
        achieved = bytes_per_launch / dur_s / 1e9 if dur_s > 0 else 0.0
        out = {
            "metric": "sequences/sec (whole node) on benchmark set, beam N=100; kcal/mol MAE vs CPU",
            "value": round(world * n * args.steps / el, 2),
            "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 FFT -> exact int counts, f64 scores, i32 dcal energies",
            "data": "benchmark_cleaned_all_length.csv sequences (committed fixture tests/golden/bench_inputs.tsv.gz); "
                    "each rank folds its own replica",
            "config": {"workload": "BASELINE configs[2]: 2296 seqs of benchmark_cleaned_all_length.csv "
                                   "(L 28..2968), nb_mode n=100, max_stack ms=50, max_branch=1000, 1 GPU per rank",
                       "nb_mode": args.nb_mode, "max_stack": args.max_stack, "max_branch": args.max_branch,
                       "sequences_per_rank": n, "parallelism": f"replica x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": "expand_kernel<64,false> (regions with FFT size <= 512)", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(bytes_per_launch, 1),
                         "mean_launch_ms": round(dur_s * 1e3, 4), "launches_per_step": launches / args.steps},
            "kernel_ms_per_step": {k: round(agg[k] / args.steps, 3) for k in ("ms_total", "ms_expand")},
            "stage_ms_untimed_pass": stage_ms,
            "memoization": {"regions_created": agg["n_nodes_created"] // args.steps,
                            "regions_expanded": agg["n_node_expansions"] // args.steps},
            "cpu_baseline": cpu,
            "parity_vs_cpu": parity,
            "larger_batch_info": scaling_info,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
