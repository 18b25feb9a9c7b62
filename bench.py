#!/usr/bin/env python3
"""bench.py - sequences/second of the RAFFT fold hot path on MI355X.

Workload (BASELINE.json configs[2], the config the metric is quoted on and the reference's own published run,
benchmark_results/bench_fft.py:8): the 2296 sequences of benchmark_cleaned_all_length.csv, nb_mode n=100,
max_stack ms=50, max_branch=1000 (CLI default).  One "step" = one pass of the whole hot path over that set.

N = 1: every step folds the whole set on the one GPU.
N > 1: WEAK scaling, per-GPU work fixed: the global batch of a step is N copies of the set (N x 2296 sequences),
       LPT-sharded over the ranks (rafft_amd/sharding.py - what the reference does with a process pool,
       benchmark_results/bench_fft.py:17-22), each rank folds its shard, no data-path collective (RCCL carries only the
       barriers and the max-reduction of the elapsed time): value = N * 2296 * K / max-over-ranks time.  The gathered
       result of one extra pass is parity-checked on rank 0 outside the timed region.  `strong_sharded_value` (ONE copy
       of the set sharded over the ranks - rounds 2 and 3 reported this as `value` at N > 1) and `cfg4_sharded`
       (BASELINE configs[3]: 16 384 random sequences L 100..3000, ms=200, LPT-sharded) ride along as extra keys.
       Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) it starts its N ranks itself - what
       the reference's driver does with Pool(int(argv[1])), benchmark_results/bench_fft.py:17 - before anything touches
       the GPU, relays rank 0's line and fails if any rank does.

Steps are issued through the library's asynchronous C-ABI (rafft_fold_submit / rafft_fold_wait) with up to twenty-one batches
in flight - continuous batching: queued batches with identical parameters are folded as ONE wave by the library's
scheduler (fewer, fuller kernel launches), and the nearly empty last folding steps of step k (only the longest sequences still fold)
run beside the busy first steps of step k+1.  Every step is waited for, and its result freed, inside the timed
region.  `ms_per_call_sequential` is the latency of one synchronous rafft_fold_batch call for comparison.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (expand_kernel<64>): algorithmic bytes
(per region 3n + 16*min(K,2n-1), per structure 3L) per launch / mean launch duration measured with HIP
events on the library's stream; `issue_frac` is its instruction-issue roofline from the SQ counters of profiles/.
`cpu_baseline` times the CPU oracle (oracle/rafft_oracle.c, a port of the reference algorithm) with one process per
core on the same workload.
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
PIPELINE_DEPTH = int(os.environ.get("BENCH_DEPTH", "21"))   # batches in flight: three waves of up to seven merged batches (profiling passes use 1)


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


class Folder:
    """the C-ABI, as a caller in the reference's language would bind it"""

    def __init__(self, lib, N, params, seqs, device):
        self.lib, self.N, self.p, self.device = lib, N, params, device
        self.n = len(seqs)
        self.enc = [s.encode() for s in seqs]
        self.arr = (C.c_char_p * self.n)(*self.enc)
        self.lens = (C.c_int * self.n)(*[len(e) for e in self.enc])
        self.agg = {}

    def submit(self):
        job = C.c_void_p()
        self.N.check(self.lib.rafft_fold_submit(C.byref(self.p), self.n, self.arr, self.lens, self.device, C.byref(job)))
        return job

    def wait(self, job, keep=False, stats=True):
        res = C.POINTER(self.N.Result)()
        self.N.check(self.lib.rafft_fold_wait(job, C.byref(res)))
        if stats:
            st = self.N.Stats()
            self.lib.rafft_get_stats(C.byref(st))
            for k, v in st.as_dict().items():
                self.agg[k] = self.agg.get(k, 0) + v
        if keep:
            return res
        ok = res.contents.n_failed == 0
        self.lib.rafft_free_result(res)
        assert ok

    def run(self, steps, depth=PIPELINE_DEPTH):
        """`steps` passes, `depth` in flight; every pass waited for and freed before this returns"""
        if self.n == 0:
            return
        if depth == 1:          # one call at a time: the synchronous entry point (rafft_fold_batch), as a blocking caller would use it
            for _ in range(steps):
                res = C.POINTER(self.N.Result)()
                self.N.check(self.lib.rafft_fold_batch(C.byref(self.p), self.n, self.arr, self.lens, self.device, C.byref(res)))
                st = self.N.Stats()
                self.lib.rafft_get_stats(C.byref(st))
                for k, v in st.as_dict().items():
                    self.agg[k] = self.agg.get(k, 0) + v
                ok = res.contents.n_failed == 0
                self.lib.rafft_free_result(res)
                assert ok
            return
        q = []
        for _ in range(steps):
            q.append(self.submit())
            if len(q) >= depth:
                self.wait(q.pop(0))
        while q:
            self.wait(q.pop(0))

    def beams(self, res):
        out = []
        for i in range(self.n):
            sr = res.contents.seq[i]
            w = sr.length + 1
            raw = C.string_at(sr.db, sr.n_structs * w).decode()
            out.append([(raw[k * w:k * w + sr.length], sr.dcal[k]) for k in range(sr.n_structs)])
        return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without torchrun: start the N ranks as child processes (one per GPU, rendezvous on
    127.0.0.1), relay rank 0's JSON line, exit non-zero if any rank fails.  Runs before this process imports torch or
    touches the GPU (mirrors benchmark_results/bench_fft.py:17-21: the reference's driver takes N and starts its own
    workers)."""
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n_ranks))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, errs = [], []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # (ranks > 0 keep their stderr in a temporary file: a rank that dies at start-up says why)
        ef = tempfile.TemporaryFile(mode="w+t") if r else None
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef, text=True))
    # every child is polled: the first one to fail ends the run (its siblings would sit in the rendezvous or a barrier until the
    # process group's timeout - tens of minutes), and the whole run has a deadline of its own
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))
    out0, failed = [], None
    import threading
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)      # rank 0's lines, as they come
    reader.start()
    while True:
        time.sleep(0.1)
        open0 = reader.is_alive()
        rcs = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(rcs) if c not in (None, 0)]
        if bad:
            failed = f"ranks failed (rank, exit code): {bad}"
            break
        if all(c == 0 for c in rcs) and not open0:
            break
        if time.time() > deadline:
            failed = "timed out (BENCH_LAUNCH_TIMEOUT_S)"
            break
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for r, ef in enumerate(errs):
            if ef is not None:
                ef.seek(0)
                tail = ef.read()[-1500:]
                if tail.strip():
                    print(f"--- rank {r} stderr (tail) ---\n{tail}", file=sys.stderr)
    lines = [l for l in out0 if l.startswith("{")]
    if failed or not lines:
        print(f"bench.py --gpus {n_ranks}: {failed or 'no JSON line'}; rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        sys.exit(1)
    line = json.loads(lines[-1])
    assert line["n_gpus"] == n_ranks, (line["n_gpus"], n_ranks)
    print(lines[-1].rstrip("\n"), flush=True)
    sys.exit(0)


def csrc_digest():
    """digest of the device/host sources of the library: the PMC summary under profiles/ carries the digest of the tree it was
    measured on (tools/profile_r04.sh), so a summary older than the build is flagged (`traffic_stale`) - the GPU box has no .git"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rafft_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nb-mode", type=int, default=100)
    ap.add_argument("--max-stack", type=int, default=50)
    ap.add_argument("--max-branch", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_SAME_GPU"):      # rehearsal of the N>1 path on a one-GPU box
        local_rank = 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_TEST_DIE_RANK") == str(rank) and world > 1:      # test hook: this rank dies at start-up
        print("BENCH_TEST_DIE_RANK: this rank exits before the rendezvous", file=sys.stderr)
        sys.exit(3)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or with "
              f"torch.distributed.run --nproc-per-node N", file=sys.stderr)
        sys.exit(2)
    # CPU baseline first, on rank 0 at N=1 only, BEFORE this process touches the GPU
    # (fork-based pool; the timed GPU region below is unaffected)
    seqs = load_bench_sequences()
    cpu, cpu_finals = None, {}
    if world == 1 and not args.no_cpu_baseline:
        cpu, cpu_finals = cpu_baseline(seqs, args.nb_mode, args.max_stack, args.max_branch)

    import torch
    import torch.distributed as dist
    backend, gloo = None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL ("nccl") carries only the barrier and the max-reduction of the elapsed time: the fold itself
        # needs no collective.  BENCH_BACKEND=gloo lets the same path be rehearsed without RCCL.
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            gloo = dist.new_group(backend="gloo")       # host-side gather of result records (outside the timed region)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    from rafft_amd import _native as N
    from rafft_amd import sharding
    from rafft_amd.rafft import _params
    import rafft_amd
    lib = N.lib()
    N.check(lib.rafft_init(local_rank))
    p = _params(args.nb_mode, args.max_stack, args.max_branch, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
    n = len(seqs)
    full = Folder(lib, N, p, seqs, local_rank)
    if world > 1:
        # weak scaling: the global batch of a step is `world` copies of the set, LPT-sharded (global index i = copy i // n of
        # sequence i % n); `strong`: ONE copy sharded over the ranks (what rounds 2 and 3 timed at N > 1)
        shard_idx = sharding.lpt_shards([len(s) for s in seqs] * world, world)[rank]
        mine = Folder(lib, N, p, [seqs[i % n] for i in shard_idx], local_rank)
        strong = Folder(lib, N, p, [seqs[i] for i in sharding.lpt_shards([len(s) for s in seqs], world)[rank]], local_rank)
    else:
        shard_idx, mine, strong = list(range(n)), full, None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # parity of the measured path against the CPU oracle on the baseline's sample (outside the timed region):
    # the WHOLE final beam (every structure, in order, with its exact dcal) identical; energy MAE in kcal/mol over
    # all beam rows (the metric's second half)
    parity = None
    if cpu_finals:
        res = full.wait(full.submit(), keep=True, stats=False)
        beams = full.beams(res)
        lib.rafft_free_result(res)
        same, same_beam, abs_err, cnt, rows = 0, 0, 0.0, 0, 0
        for s, beam in zip(seqs, beams):
            if s not in cpu_finals:
                continue
            want = cpu_finals[s]
            same += int(beam[0] == want[0])
            same_beam += int(beam == want)
            for (_, gd), (_, wd) in zip(beam, want):
                abs_err += abs(gd - wd) / 100.0
                rows += 1
            cnt += 1
        parity = {"sequences_compared": cnt, "final_beam_identical": same_beam, "beam_rows_compared": rows,
                  "lowest_energy_structure_identical": same,
                  "energy_mae_kcal_per_mol": abs_err / max(rows, 1)}

    # ---------------------------------------------------------------- the timed region
    # Steps in flight: PIPELINE_DEPTH on every rank - a step hands every rank as many sequences as the single GPU folds
    # (weak scaling), so every rank's scheduler sees the N = 1 stream of batches.
    DEPTH_RUN = PIPELINE_DEPTH
    # (a fresh box hands over a GPU in its low-power state: half a second of the same work, untimed, before the W warm-up
    #  steps, so that the clocks have ramped whatever W is)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < float(os.environ.get("BENCH_PREWARM_S", "0.5")):
        mine.run(DEPTH_RUN, depth=DEPTH_RUN)
    mine.run(args.warmup, depth=DEPTH_RUN)
    agg_untimed = dict(mine.agg)        # (pre-warm and warmup launches: with the timed ones they are what rocprofv3 averages over)
    mine.agg = {}
    barrier()
    if os.environ.get("RAFFT_TRACE_ALLOC"):
        print(f"[bench] timed region starts t={time.monotonic():.3f}", file=sys.stderr, flush=True)
    def alloc_counters():
        a = (C.c_ulonglong * 5)()
        lib.rafft_alloc_counters(C.byref(a))
        return list(a)

    a0 = alloc_counters()
    t0 = time.perf_counter()
    mine.run(args.steps, depth=DEPTH_RUN)
    el_own = time.perf_counter() - t0          # this rank's own steps done (before the barrier)
    barrier()
    el = allmax(time.perf_counter() - t0)
    a1 = alloc_counters()
    # what the process group really was: ranks seen, the device each one folded on, its sequences per step and its own elapsed time
    # (a SCALE record can then be checked for N real GPUs and for the rank that set the pace)
    ranks_info = None
    if world > 1:
        try:
            dev_id = str(torch.cuda.get_device_properties(torch.cuda.current_device()).uuid)
        except Exception:
            dev_id = f"ordinal {torch.cuda.current_device()}"
        mine_rec = {"rank": rank, "device_ordinal": int(torch.cuda.current_device()), "device": dev_id, "sequences_per_step": mine.n, "elapsed_s": round(el_own, 6)}
        recs = [None] * world
        dist.all_gather_object(recs, mine_rec, group=gloo)
        els = [r["elapsed_s"] for r in recs]
        ranks_info = {"world_size_seen": dist.get_world_size(), "distinct_devices": len({r["device"] for r in recs}),
                      "device_ordinals": [r["device_ordinal"] for r in recs], "sequences_per_step_by_rank": [r["sequences_per_step"] for r in recs],
                      "elapsed_s_by_rank": els, "elapsed_s_max": max(els), "elapsed_s_mean": round(sum(els) / len(els), 6), "elapsed_s_min": min(els),
                      "elapsed_s_with_barrier_max_over_ranks": round(el, 6)}
    # the library's allocations inside the timed region (rank 0): none when the workspaces were sized by the untimed steps
    timed_allocs = {"device_buffers": a1[0] - a0[0], "device_MB": round((a1[1] - a0[1]) / 1e6, 1),
                    "pinned_chunks": a1[3] - a0[3], "pinned_MB": round((a1[4] - a0[4]) / 1e6, 1),
                    "slowest_device_allocation_ms_since_start": round(a1[2] / 1e3, 3)}
    if os.environ.get("RAFFT_TRACE_ALLOC"):
        print(f"[bench] timed region ends t={time.monotonic():.3f}", file=sys.stderr, flush=True)
    agg = dict(mine.agg)

    extras = {}
    if not args.no_extras:
        # steady state (outside `value`): at least two seconds of the same sharded steps, so that a short timed region - whose
        # start-up and drain are a visible part at N >= 4, where every rank folds one or two merged waves - has a figure beside it
        # that is neither
        barrier()
        t1 = time.perf_counter()
        n_ss = 0
        while True:
            mine.run(2 * DEPTH_RUN, depth=DEPTH_RUN)
            n_ss += 2 * DEPTH_RUN
            flag = allmax(1.0 if time.perf_counter() - t1 < 2.0 else 0.0)
            if flag == 0.0:
                break
        barrier()
        extras["steady_state_value"] = round(n * world * n_ss / allmax(time.perf_counter() - t1), 2)
        extras["steady_state_steps"] = n_ss
        # latency of one synchronous call (no second batch in flight)
        mine.run(2, depth=1)
        barrier()
        t1 = time.perf_counter()
        mine.run(max(3, args.steps // 2), depth=1)
        barrier()
        extras["ms_per_call_sequential"] = round(allmax(time.perf_counter() - t1) / max(3, args.steps // 2) * 1e3, 3)
        # the same through rafft_fold_submit + rafft_fold_wait, one at a time: what an asynchronous or Python caller with ONE batch pays
        # (a lone submit lingers up to RAFFT_LINGER_US = 150 us for followers before its waves are admitted; rafft_fold_batch never does)
        barrier()
        t1 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            mine.wait(mine.submit())
        barrier()
        extras["ms_per_call_submit_wait"] = round(allmax(time.perf_counter() - t1) / max(3, args.steps // 2) * 1e3, 3)
        if world > 1:
            # (a) strong scaling: ONE copy of the set LPT-sharded over the ranks, `world` times as many steps in flight so that a
            # rank's scheduler still merges its 1/N-sized shards into full waves (rounds 2 and 3 reported this as `value`)
            strong.run(2 * world, depth=PIPELINE_DEPTH * world)
            barrier()
            t1 = time.perf_counter()
            strong.run(args.steps, depth=PIPELINE_DEPTH * world)
            barrier()
            el_strong = allmax(time.perf_counter() - t1)
            extras["strong_sharded_value"] = round(n * args.steps / el_strong, 2)
            # the same two figures side by side under names that say what they are (`value` is the weak one, as `scaling` says)
            extras["strong_value"] = extras["strong_sharded_value"]
            extras["strong_steps"] = args.steps
            extras["strong_sequences_per_step"] = n
            extras["strong_ms_per_step"] = round(el_strong / args.steps * 1e3, 3)
            # (b) the gathered result of the sharded path (every copy of every sequence, whichever rank folded it) == rank 0's own
            # fold of the whole set
            res = mine.wait(mine.submit(), keep=True, stats=False)
            payload = list(zip(shard_idx, mine.beams(res)))
            lib.rafft_free_result(res)
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(payload, gathered, dst=0, group=gloo)
            if rank == 0:
                got = [None] * (n * world)
                for part in gathered:
                    for i, b in part:
                        got[i] = b
                res = full.wait(full.submit(), keep=True, stats=False)
                want = full.beams(res)
                lib.rafft_free_result(res)
                extras["sharded_parity"] = {"sequences": n * world, "final_beam_identical_to_single_gpu_fold": sum(int(a == want[i % n]) for i, a in enumerate(got))}
        # (c) BASELINE configs[3]: 16 384 random sequences, L ~ U[100, 3000], ms=200, LPT-sharded over the ranks
        if (world > 1 and not os.environ.get("BENCH_SKIP_CFG4")) or os.environ.get("BENCH_CFG4"):
            import numpy as np
            rng = np.random.default_rng(3000)
            lens4 = rng.integers(100, 3001, size=16384)
            sh4 = set(sharding.lpt_shards([int(x) for x in lens4], world)[rank])
            seqs4 = []
            for i, ln in enumerate(lens4):
                c = rng.choice(4, int(ln))
                if i in sh4:
                    seqs4.append("".join("ACGU"[k] for k in c))
            p4 = _params(args.nb_mode, 200, args.max_branch, 3, 0.0, False, 37.0, 3.0, 2.0, 1.0)
            f4 = Folder(lib, N, p4, seqs4, local_rank)
            f4.run(1, depth=1)                     # first call sizes the HBM workspace
            barrier()
            t1 = time.perf_counter()
            f4.run(2, depth=1)
            barrier()
            el4 = allmax(time.perf_counter() - t1)
            extras["cfg4_sharded"] = {"sequences": 16384, "max_stack": 200, "passes": 2, "ms_per_pass": round(el4 / 2 * 1e3, 1),
                                      "sequences_per_s": round(16384 * 2 / el4, 1), "sequences_on_rank0": len(seqs4)}
            del f4

    # per-stage kernel times: one extra call, outside the timed region, with every stage bracketed by HIP events
    # (the timed steps carry events only around the dominant kernel - a pair around every kernel costs ~1.3 ms)
    stage_ms = None
    if rank == 0 and not args.no_extras:
        os.environ["RAFFT_SPANS"] = "2"
        full.agg = {}
        full.wait(full.submit())
        del os.environ["RAFFT_SPANS"]
        stage_ms = {k: round(v, 3) for k, v in full.agg.items() if k.startswith("ms_")}
    # the drop-in Python API on the same workload (outside the timed region): rafft_amd.fold_batch returns lazily
    # materialised rows, so it costs what the C call costs
    py_api = None
    if rank == 0 and world == 1 and not args.no_extras:
        rafft_amd.fold_batch(seqs, args.nb_mode, args.max_stack, args.max_branch)
        t1 = time.perf_counter()
        for _ in range(5):
            r = rafft_amd.fold_batch(seqs, args.nb_mode, args.max_stack, args.max_branch)
            first = r[0][0].str_struct
        t_seq = (time.perf_counter() - t1) / 5
        t1 = time.perf_counter()
        q = []
        n_py = 2 * PIPELINE_DEPTH           # (as many batches in flight as the timed loop keeps, twice over)
        for _ in range(n_py):
            q.append(rafft_amd.submit_batch(seqs, args.nb_mode, args.max_stack, args.max_branch))
            if len(q) >= PIPELINE_DEPTH:
                q.pop(0).result()
        while q:
            q.pop(0).result()
        t_pipe = (time.perf_counter() - t1) / n_py
        py_api = {"fold_batch_ms": round(t_seq * 1e3, 3), "fold_batch_sequences_per_s": round(n / t_seq, 1),
                  "submit_batch_pipelined_sequences_per_s": round(n / t_pipe, 1), "first_structure": first[:24] + "..."}
    # how many FINAL structures carry an energy that read a rule / model value of the built-in interior-loop tables (DESIGN.md 2.1:
    # such an entry is right in ~9 of 10 cases; no reference-held energy row pins it) - per workload, in the record and not only in the
    # design notes.  configs[2] (this set), configs[1] (1000 random sequences of 200 nt, ms=50), a sample of configs[3] (ms=200)
    guessed = None
    if rank == 0 and world == 1 and not args.no_extras:
        import numpy as np

        def guessed_share(seq_list, ms, every=1):
            res_ = rafft_amd.fold_batch(seq_list, args.nb_mode, ms, args.max_branch)
            ss, dbs = [], []
            for sq, fin in zip(seq_list, res_):
                for x in list(fin)[::every]:
                    ss.append(sq)
                    dbs.append(x.str_struct)
            _, stt, gs = rafft_amd.rafft.eval_structures_info(ss, dbs)
            lowest = [int(g) for g in gs]       # (every structure listed; the lowest-energy one of a sequence is its first)
            first = []
            k = 0
            for sq, fin in zip(seq_list, res_):
                cnt = len(list(fin)[::every])
                if cnt:
                    first.append(lowest[k])
                k += cnt
            return {"final_structures": len(dbs), "with_guessed_entry": int(sum(gs)), "share": round(sum(gs) / max(1, len(dbs)), 4),
                    "lowest_energy_structures_with_guessed_entry_share": round(sum(first) / max(1, len(first)), 4), "eval_failures": int(sum(1 for v in stt if v))}
        rng2 = np.random.default_rng(200)
        cfg2 = ["".join("ACGU"[k] for k in rng2.integers(0, 4, 200)) for _ in range(1000)]
        rng4 = np.random.default_rng(3000)
        lens4 = rng4.integers(100, 3001, size=16384)[:24]
        cfg4s = ["".join("ACGU"[k] for k in rng4.integers(0, 4, int(ln))) for ln in lens4]
        guessed = {"configs[2] benchmark set (ms=50)": guessed_share(seqs, args.max_stack),
                   "configs[1] 1000 random x 200 nt (ms=50)": guessed_share(cfg2, 50),
                   "configs[3] sample: 24 random sequences of 100..3000 nt (ms=200)": guessed_share(cfg4s, 200)}
    # informational, outside the timed region: the same call on a 4x larger batch (the set replicated 4 times in
    # ONE rafft_fold_batch call).  Never part of `value`.
    scaling_info = None
    if world == 1 and not args.no_cpu_baseline and not args.no_extras:
        R = 4
        big = Folder(lib, N, p, seqs * R, local_rank)
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            big.run(1, depth=1)
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        scaling_info = {"replicas_in_one_call": R, "sequences": n * R, "ms": round(best * 1e3, 3),
                        "sequences_per_s": round(n * R / best, 1)}

    if rank == 0:
        # HBM traffic and SQ counters of the dominant kernel come from rocprofv3 PMC passes (separate runs,
        # gfx950 correction applied by tools/pmc_traffic.py); counters cannot be read in-process.
        traffic, traffic_src, issue, traffic_batch, tj = None, None, None, None, None
        for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                k = next((v for kn, v in tj["kernels"].items() if kn.startswith("void expand_kernel<64")), None)
                if k:
                    traffic, traffic_src = k["hbm_bytes_per_launch"], f"profiles/{name}: " + tj["correction"]
                    traffic_batch = k.get("hbm_bytes_per_batch")
                    issue = tj.get("issue_roofline")
                break
        # which kernel is "dominant" depends on the clock: summed durations of the pipelined trace (inflated for kernels that queue for
        # CUs), the same weighted with the share of the chip a launch can occupy, or a serial trace (every kernel alone on the chip).
        # The committed profile has all three (tools/profile_r05.sh -> tools/dominant.py); the roofline above is for `roofline.kernel`,
        # and `named_kernel_is_top_serial` says whether that is the top row of the serial statistics.
        dominant_by = None
        for name in ("r05_dominant.json",):
            dpath = os.path.join(ROOT, "profiles", name)
            if os.path.exists(dpath):
                dominant_by = json.load(open(dpath))
                dominant_by["source_file"] = "profiles/" + name
                dominant_by["named_kernel_is_top_serial"] = str(dominant_by.get("top_serial", "")).startswith("expand_kernel<64")
        launches = max(1, agg.get("n_expand_launches", 0))
        if traffic is not None and traffic_batch:
            # the PMC passes run synchronous calls (7 launches of this kernel per batch); the timed loop folds several
            # batches per wave, i.e. fewer and bigger launches: per launch LIKE `achieved` = bytes per batch / launches per step
            traffic = round(traffic_batch * args.steps / launches)
            traffic_src += "; per batch in the PMC run, divided by the launches per step of the timed loop"
        # rocprofv3 --stats averages over EVERY launch of the process - warmup waves (W batches: a smaller wave than the timed loop's
        # seven) included: the same average from the library's HIP events, to set beside profiles/*_kernel_stats.csv
        launches_all = launches + agg_untimed.get("n_expand_launches", 0)
        dur_all_ms = (agg.get("ms_expand", 0.0) + agg_untimed.get("ms_expand", 0.0)) / max(1, launches_all)
        dur_s = agg.get("ms_expand", 0.0) / 1e3 / launches
        bytes_per_launch = agg.get("alg_bytes_expand", 0) / launches
        achieved = bytes_per_launch / dur_s / 1e9 if dur_s > 0 else 0.0
        # every kernel family of the step beside the dominant one: algorithmic bytes and its time in the untimed
        # pass that brackets every stage with HIP events; counter bytes and issue fraction from the tracked PMC summary
        kernels = None
        if stage_ms and full.agg:
            fa = full.agg

            def pmc(prefix):
                if not tj:
                    return None, None
                kk = next((v for kn, v in tj["kernels"].items() if kn.replace("void ", "").startswith(prefix)), None)
                ii = next((v for kn, v in tj.get("issue", {}).items() if kn.replace("void ", "").startswith(prefix)), None)
                return (kk or {}).get("hbm_bytes_per_batch"), (ii or {}).get("issue_frac")
            rows = [("expand_kernel<64,true,16>", "expand_kernel<64", fa.get("alg_bytes_expand", 0), fa.get("ms_expand", 0.0)),
                    ("expand_small_kernel<16|32>", "expand_small_kernel", fa.get("alg_bytes_expand_small", 0), fa.get("ms_expand_c1", 0.0)),
                    ("expand_kernel<256>", "expand_kernel<256", fa.get("alg_bytes_expand_c2", 0), fa.get("ms_expand_c2", 0.0)),
                    ("expand_kernel<512>", "expand_kernel<512", fa.get("alg_bytes_expand_c3", 0), fa.get("ms_expand_c3", 0.0)),
                    ("beam_step_kernel + materialize_kernel + dedupe_kernel (2L+8 per new structure)", "beam_step_kernel<256", fa.get("alg_bytes_beam", 0),
                     fa.get("ms_beam", 0.0) + fa.get("ms_materialize", 0.0))]
            kernels = []
            for label, prefix, ab, ms in rows:
                cb, fr = pmc(prefix)
                kernels.append({"kernel": label, "alg_bytes_per_step": int(ab), "ms_per_step_untimed_pass": round(ms, 3),
                                "achieved_gbs": round(ab / ms / 1e6, 2) if ms > 0 else None,
                                "frac": round(ab / ms / 1e6 / HBM_PEAK_GBS, 6) if ms > 0 else None,
                                "counter_bytes_per_batch": cb, "issue_frac": fr})
        out = {
            "metric": "sequences/sec (whole node) on benchmark set, beam N=100; kcal/mol MAE vs CPU",
            "value": round(n * world * args.steps / el, 2),
            "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "scaling_note": "per-GPU work fixed, as the bench contract asks of a path that shards: a step folds n_gpus copies of the set, "
                            "LPT-sharded, value = weak_value = n_gpus x 2296 x steps / time.  The reference's own driver (one copy of the file "
                            "over Pool(N), benchmark_results/bench_fft.py:17-22) is the STRONG figure: `strong_value` beside it with its own step "
                            "count (rounds 2 and 3 printed that one as `value` at N > 1).  N = 1 values are like-for-like across all rounds; "
                            "`vs_baseline` is null (no published number), so no ratio is formed from either",
            "dtype": "f32 FFT -> exact int counts, f64 scores, i32 dcal energies",
            "data": "benchmark_cleaned_all_length.csv sequences (committed fixture tests/golden/bench_inputs.tsv.gz)"
                    + (f"; a step folds {world} copies of the set, LPT-sharded over the ranks" if world > 1 else ""),
            "config": {"workload": "BASELINE configs[2]: 2296 seqs of benchmark_cleaned_all_length.csv "
                                   "(L 28..2968), nb_mode n=100, max_stack ms=50, max_branch=1000",
                       "nb_mode": args.nb_mode, "max_stack": args.max_stack, "max_branch": args.max_branch,
                       "sequences_per_step": n * world, "sequences_on_rank0": mine.n,
                       "parallelism": (f"LPT sequence shards x{world}, no collective" if world > 1 else "1 GPU"),
                       "batches_in_flight": DEPTH_RUN,
                       "scheduler": "queued batches with equal parameters are folded as ONE wave of up to "
                                    + os.environ.get("RAFFT_MERGE_SEQS", "16384") + " sequences (7 steps of this workload), three waves at a time; "
                                    "ms_per_call_sequential is one synchronous call"},
            "roofline": {"bound": "hbm", "kernel": "expand_kernel<64,true,16> (regions of 33..256 positions, up to 128 branches; 16 one-wavefront teams per workgroup)", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(bytes_per_launch, 1),
                         "mean_launch_ms": round(dur_s * 1e3, 4), "launches_per_step": launches / args.steps,
                         "mean_launch_ms_all_launches": round(dur_all_ms, 4), "launches_incl_untimed": launches_all,
                         "issue_roofline": issue, "dominant_by": dominant_by,
                         "traffic_age": (tj or {}).get("commit"),
                         # the PMC summary was measured on another build of rafft_amd/csrc than the one that just ran
                         "traffic_stale": (tj or {}).get("csrc_digest") != csrc_digest(),
                         "kernels": kernels},
            "kernel_ms_per_step": {"ms_expand": round(agg.get("ms_expand", 0.0) / args.steps, 3),
                                   "batch_latency_ms_mean": round(agg.get("ms_total", 0.0) / args.steps, 3)},
            "stage_ms_untimed_pass": stage_ms,
            "memoization": {"region_instances": agg.get("n_node_instances", 0) // args.steps,      # (structure, region) pairs
                            "regions_created": agg.get("n_nodes_created", 0) // args.steps,      # records written: first pick of a (parent region, candidate, side)
                            "regions_aliased": agg.get("n_nodes_aliased", 0) // args.steps,      # ... that turned out to be a known loop (another path)
                            "regions_expanded": agg.get("n_node_expansions", 0) // args.steps,
                            # children the combine step accepted (each one a probe + an insert in its sequence's `seen` set, rafft.py:196-200)
                            "children_accepted": agg.get("n_children", 0) // args.steps, "structures_materialized": agg.get("n_structs", 0) // args.steps,
                            # built-in tables: stem energies evaluated / involving a rule or model value / kept candidates that do
                            "dE_evaluations": agg.get("n_dE_evals", 0) // args.steps, "dE_with_guessed_entry": agg.get("n_dE_guessed", 0) // args.steps,
                            "kept_candidates_with_guessed_entry": agg.get("n_kept_guessed", 0) // args.steps},
            "weak_value": round(n * world * args.steps / el, 2), "weak_steps": args.steps, "weak_sequences_per_step": n * world,
            "ranks": ranks_info,
            "allocations_in_timed_region": timed_allocs,
            # waves of the timed region that overflowed an HBM arena and were folded again with larger ones (0 in a healthy run)
            "regrows_in_timed_region": int(agg.get("n_regrows", 0)),
            "cpu_baseline": cpu,
            "parity_vs_cpu": (dict(parity, guessed_share_of_final_structures=guessed) if parity is not None else
                              ({"guessed_share_of_final_structures": guessed} if guessed is not None else None)),
            "python_api": py_api,
            "larger_batch_info": scaling_info,
        }
        out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
