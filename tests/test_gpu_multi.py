"""The N > 1 path on the one GPU of the test box: two ranks (child processes, gloo) sharing the card.
The 8-GPU run is the driver's; this covers its code path - LPT shards, every rank folding its shard through the real
libraffthip.so, host-side gather on rank 0 - and bench.py's multi-rank mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rafft_amd
from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import torch.distributed as dist
import rafft_amd
from rafft_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
seqs = json.load(open({seqfile!r}))
res = sharding.fold_sharded(seqs, device=0, nb_mode=100, max_stack=20, max_branch=1000, traj=True)
if dist.get_rank() == 0:
    json.dump([[[(x.str_struct, x.dcal) for x in st] for st in traj] for fin, traj in res], open({outfile!r}, "w"))
    json.dump(rafft_amd.last_stats(), open({outfile!r} + ".stats", "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_gpu_two_ranks_one_gpu_sharded_fold(tmp_path):
    rng = np.random.default_rng(4)
    lens = [int(x) for x in rng.integers(20, 300, size=120)] + [900, 1400]
    seqs = ["".join(rng.choice(list("ACGU"), n)) for n in lens]
    seqfile, outfile = tmp_path / "seqs.json", tmp_path / "out.json"
    seqfile.write_text(json.dumps(seqs))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, seqfile=str(seqfile), outfile=str(outfile)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = json.loads(outfile.read_text())
    want = rafft_amd.fold_batch(seqs, 100, 20, 1000, traj=True)
    assert [[[tuple(x) for x in st] for st in t] for t in got] == [[[(x.str_struct, x.dcal) for x in st] for st in traj] for fin, traj in want]
    st = json.loads((tmp_path / "out.json.stats").read_text())
    assert 0 < st["n_structs"] and st["n_regrows"] == 0           # rank 0 really folded (its shard) on the GPU


def _check_two_rank_line(out):
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 4
    # a step folds two copies of the set, LPT-sharded: every rank holds about one set's worth of sequences
    assert out["config"]["sequences_per_step"] == 2 * 2296 and 2000 < out["config"]["sequences_on_rank0"] < 2600
    assert out["sharded_parity"] == {"sequences": 2 * 2296, "final_beam_identical_to_single_gpu_fold": 2 * 2296}
    assert out["strong_sharded_value"] > 0 and out["roofline"]["frac"] > 0
    # (round 5) the N > 1 line explains itself: weak and strong figures under their own names with their step counts, and what the
    # process group really was - ranks seen, the device of each (here both ranks share the one card), sequences and elapsed time by rank
    assert out["weak_value"] == out["value"] and out["weak_steps"] == 4 and out["weak_sequences_per_step"] == 2 * 2296
    assert out["strong_value"] == out["strong_sharded_value"] and out["strong_steps"] == 4 and out["strong_sequences_per_step"] == 2296
    rk = out["ranks"]
    assert rk["world_size_seen"] == 2 and rk["distinct_devices"] == 1 and rk["device_ordinals"] == [0, 0]
    assert sum(rk["sequences_per_step_by_rank"]) == 2 * 2296 and len(rk["elapsed_s_by_rank"]) == 2
    assert rk["elapsed_s_min"] <= rk["elapsed_s_mean"] <= rk["elapsed_s_max"] <= rk["elapsed_s_with_barrier_max_over_ranks"] + 1e-3


def test_gpu_bench_two_ranks_sharded_mode():
    """bench.py as the driver starts it for N > 1 (here: 2 ranks on the one GPU, gloo instead of RCCL): weak scaling over LPT
    shards of two copies of the set, the gathered result equal to a single-GPU fold of the whole set"""
    env = dict(os.environ, BENCH_SAME_GPU="1", BENCH_BACKEND="gloo", BENCH_SKIP_CFG4="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29657", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    _check_two_rank_line(json.loads(line))


def test_gpu_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no torchrun, WORLD_SIZE unset) starts its two ranks itself, relays rank 0's
    line with n_gpus == 2 (benchmark_results/bench_fft.py:17: the reference's driver takes N and starts its own workers)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_SAME_GPU="1", BENCH_BACKEND="gloo", BENCH_SKIP_CFG4="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check_two_rank_line(json.loads(lines[0]))


def test_gpu_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus 2 under a launcher that started one rank: fails loudly instead of folding on one GPU and calling it two"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
