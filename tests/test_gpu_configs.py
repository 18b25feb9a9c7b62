"""Every BASELINE config end to end on the GPU, against the oracle at full size.

cfg1  bin/rafft as a process (CLI -> rafft_amd -> ctypes -> libraffthip.so), all output formats
cfg3  all 2296 benchmark sequences, n=100 ms=50: the FULL final beam of every sequence vs the oracle
cfg3+ the two ~2.9-knt 23S sequences and random 2-3-knt sequences: full trajectories (expand_kernel<512>
      at FFT size 8192, beam_step_kernel<1024>)
cfg4  one real LPT shard (1/8) of the 16 384-sequence mixed-length set at ms=200

The oracle side runs in spawned worker processes (tests/_oracle_pool.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import rafft as R, sharding, utils
from conftest import GOLD, ROOT
from _oracle_pool import fold_many
from test_gpu_scale import check_structures

pytestmark = pytest.mark.gpu

TRNA = "GGGGAAUUAGCUCAAAUGGUAGAGCGCUCGCUUAGCAUGCGAGAGGUAGCGGGAUCGAUGCCCGCAUUCUCCACCA"
EX = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"
RAFFT = os.path.join(ROOT, "bin", "rafft")


def beam_key(beam):
    return [(x.str_struct, x.dcal) for x in beam]


def traj_key(traj):
    return [beam_key(st) for st in traj]


def run_cli(*argv):
    r = subprocess.run([sys.executable, RAFFT, *argv], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def fmt_final(seq, beam):
    return "\n".join([seq] + [f"{db} {utils.Structure(db, d).energy:6.1f}" for db, d in beam]) + "\n"


def test_gpu_cfg1_cli_process_end_to_end(tmp_path):
    """BASELINE configs[0]: bin/rafft started as a process on the GPU box; stdout compared byte for byte with
    the oracle's result in the reference's formats (bin/rafft:59-79)"""
    # final format, CLI defaults (n=100, ms=1, max_branch=1000)
    o1 = [(x.str_struct, x.dcal) for x in oracle.fold(TRNA, 100, 1, 1000)]
    assert run_cli("-s", TRNA) == fmt_final(TRNA, o1)
    o50 = [(x.str_struct, x.dcal) for x in oracle.fold(TRNA, 100, 50, 1000)]
    assert run_cli("-s", TRNA, "-n", "100", "-ms", "50") == fmt_final(TRNA, o50)
    # --traj: the reference's own example outputs
    assert run_cli("-s", EX, "-ms", "5", "--traj") == open(os.path.join(GOLD, "example_rafft.out")).read()
    assert run_cli("-s", EX, "-ms", "20", "-tr") == open(os.path.join(GOLD, "example_rafft_20.out")).read()
    # --bench rows (the format bench_fft.py:8 collects)
    want = "".join(f"{TRNA} {len(TRNA)} {db} {utils.Structure(db, d).energy:6.1f} {db.count('(')}\n" for db, d in o50)
    assert run_cli("-s", TRNA, "-n", "100", "-ms", "50", "--bench") == want
    # -sf: FASTA with T (one record: lines joined, T->U, bin/rafft:42) and --batch (several records, one GPU batch)
    fa = tmp_path / "one.fa"
    dna = TRNA.replace("U", "T")
    fa.write_text(">tRNA\n" + dna[:40] + "\n" + dna[40:] + "\n")
    assert run_cli("-sf", str(fa)) == fmt_final(TRNA, o1)
    rng = np.random.default_rng(76)
    more = ["".join(rng.choice(list("ACGU"), int(n))) for n in (30, 76, 140)]
    fb = tmp_path / "many.fa"
    fb.write_text("".join(f">s{k}\n{s}\n" for k, s in enumerate([TRNA, EX] + more)))
    got = run_cli("-sf", str(fb), "--batch", "-ms", "10", "--bench")
    want = ""
    for s in [TRNA, EX] + more:
        for x in oracle.fold(s, 100, 10, 1000):
            want += f"{s} {len(s)} {x.str_struct} {utils.Structure(x.str_struct, x.dcal).energy:6.1f} {x.str_struct.count('(')}\n"
    assert got == want
    # errors surface as the reference's exceptions (non-zero exit, KeyError in the traceback)
    r = subprocess.run([sys.executable, RAFFT, "-s", "ACGT"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "KeyError" in r.stderr


def test_gpu_cfg3_full_beam_all_2296_vs_oracle(bench_rows):
    """BASELINE configs[2]: every sequence of the benchmark set, n=100 ms=50 max_branch=1000 - the whole final
    beam (structures in order + exact dcal) equals the oracle's, for all 2296"""
    seqs = [r["seq"] for r in bench_rows]
    want = fold_many([(s, 100, 50, 1000, False) for s in seqs])
    got = rafft_amd.fold_batch(seqs, 100, 50, 1000)
    bad = [i for i, (g, w) in enumerate(zip(got, want)) if beam_key(g) != w]
    assert not bad, (len(bad), [(i, len(seqs[i])) for i in bad[:10]])
    n_struct = sum(len(w) for w in want)
    assert n_struct > 100000          # ~50 per sequence: it is the whole beam that was compared


@pytest.fixture(scope="module")
def long_seqs(bench_rows):
    seqs = sorted((r["seq"] for r in bench_rows), key=len)[-2:]         # the two 23S, 2915 and 2968 nt
    rng = np.random.default_rng(2968)
    seqs += ["".join(rng.choice(list("ACGU"), n)) for n in (2000, 2500, 3000)]
    return seqs


def test_gpu_long_sequences_full_trajectory_vs_oracle(long_seqs):
    """sequences of 2-3 knt: regions of the 512-thread expand class (FFT size 4096/8192) and the 1024-thread
    beam step, full trajectory (every beam of every folding step) against the oracle at ms=50 and ms=200"""
    tasks = [(s, 100, 50, 1000, True) for s in long_seqs] + [(long_seqs[1], 100, 200, 1000, True), (long_seqs[4], 100, 200, 1000, True)]
    want = fold_many(tasks)
    got50 = rafft_amd.fold_batch(long_seqs, 100, 50, 1000, traj=True)
    for k, (fin, traj) in enumerate(got50):
        assert traj_key(traj) == want[k], (k, len(long_seqs[k]))
        assert len(traj) >= 15
    got200 = rafft_amd.fold_batch([long_seqs[1], long_seqs[4]], 100, 200, 1000, traj=True)
    for k, (fin, traj) in enumerate(got200):
        assert traj_key(traj) == want[len(long_seqs) + k], (k, 200)


def test_gpu_cfg4_real_lpt_shard_ms200():
    """BASELINE configs[3]: 16 384 random sequences, L ~ U[100,3000], ms=200, LPT-sharded over 8 GPUs - ONE real
    shard (2048 sequences) on this GPU: size-independent properties on every result, whole-structure energy
    re-evaluation of every final structure, and the full final beam against the oracle for every sequence <= 400 nt"""
    rng = np.random.default_rng(3000)
    lens = rng.integers(100, 3001, size=16384)
    shard = sharding.lpt_shards([int(x) for x in lens], 8)[0]
    assert len(shard) == 2048
    # bases of shard members only (the generator is advanced identically for every sequence, so the set is well defined)
    seqs_all = {}
    want_idx = set(shard)
    for i, n in enumerate(lens):
        s = rng.choice(4, int(n))
        if i in want_idx:
            seqs_all[i] = "".join("ACGU"[c] for c in s)
    seqs = [seqs_all[i] for i in shard]
    small = [k for k, s in enumerate(seqs) if len(s) <= 400]
    assert len(small) > 100
    want = fold_many([(seqs[k], 100, 200, 1000, False) for k in small])
    res = rafft_amd.fold_batch(seqs, 100, 200, 1000)
    for s, beam in zip(seqs, res):
        check_structures(s, beam, 200)
    flat = [(s, x.str_struct, x.dcal) for s, beam in zip(seqs, res) for x in beam]
    step = 200000                         # bounded host/device buffers per evaluation call
    for a in range(0, len(flat), step):
        part = flat[a:a + step]
        got, st = R.eval_structures([f[0] for f in part], [f[1] for f in part])
        assert not any(st) and got == [f[2] for f in part]
    bad = [k for k, w in zip(small, want) if beam_key(res[k]) != w]
    assert not bad, (len(bad), [len(seqs[k]) for k in bad[:10]])


def _cfg4_set():
    """BASELINE configs[3]: 16 384 random sequences, L ~ U[100, 3000] (default_rng(3000): lengths first, then the bases of
    every sequence in turn) and their LPT shards over 8 GPUs"""
    rng = np.random.default_rng(3000)
    lens = rng.integers(100, 3001, size=16384)
    seqs = ["".join("ACGU"[c] for c in rng.choice(4, int(n))) for n in lens]
    return seqs, sharding.lpt_shards([int(x) for x in lens], 8)


def test_gpu_cfg4_all_eight_shards_ms200():
    """BASELINE configs[3] whole: every one of the eight LPT shards (2048 sequences each, ms=200) folded on this GPU.  On all 16 384
    results: beam sorted by energy, no structure twice, and the energy of EVERY final structure (3.2 M of them) re-evaluated from its
    dot-bracket by the whole-structure kernel (balanced, canonical pairs only, exact dcal - a checksum of the incrementally summed dE
    over every folding step).  Against the oracle: the full final beam of a stratified sample that includes members of 1000-3000 nt
    (two per 400-nt length band, from different shards) - the size classes the headline workload barely touches."""
    import ctypes as C
    from rafft_amd import _native as N
    seqs, shards = _cfg4_set()
    assert sorted(i for sh in shards for i in sh) == list(range(16384)) and all(len(sh) == 2048 for sh in shards)
    # the oracle sample first (worker processes, while the GPU folds): by length band, members of different shards
    shard_of = {i: k for k, sh in enumerate(shards) for i in sh}
    sample = []
    for b, lo in enumerate((100, 500, 900, 1300, 1700, 2100, 2500, 2800)):
        cand = [i for i in range(16384) if lo <= len(seqs[i]) < lo + 200]
        sample += [next(i for i in cand if shard_of[i] == k) for k in (b, (3 * b + 5) % 8)]
    assert max(len(seqs[i]) for i in sample) >= 2800 and len({shard_of[i] for i in sample}) >= 6
    import threading
    want = {}
    th = threading.Thread(target=lambda: want.update(zip(sample, fold_many([(seqs[i], 100, 200, 1000, False) for i in sample]))))
    th.start()
    lib = N.lib()
    n_struct = 0
    got_sample = {}
    for k, sh in enumerate(shards):
        mine = [seqs[i] for i in sh]
        res = rafft_amd.fold_batch(mine, 100, 200, 1000)
        assert rafft_amd.last_stats()["n_regrows"] <= 1, (k, rafft_amd.last_stats())
        enc = [s.encode() for s in mine]
        seq_addr, db_addr, dcal_all = [], [], []
        for j, s in enumerate(mine):
            L, sizes, rows, dcal = res.raw(j)
            assert L == len(s) and sizes == [len(dcal)] and 1 <= len(dcal) <= 200
            assert (np.diff(dcal) >= 0).all()                                       # sorted by energy (rafft.py:207)
            assert len({r.tobytes() for r in rows}) == len(rows)                       # `seen` dedupe (rafft.py:196-200)
            base = rows.ctypes.data
            db_addr.append(base + np.arange(len(dcal), dtype=np.uint64) * np.uint64(L + 1))
            seq_addr.append(np.full(len(dcal), C.cast(C.c_char_p(enc[j]), C.c_void_p).value, dtype=np.uint64))
            dcal_all.append(np.array(dcal))
            if sh[j] in want or sh[j] in sample:
                got_sample[sh[j]] = beam_key(res[j])
        seq_addr, db_addr, dcal_all = np.concatenate(seq_addr), np.concatenate(db_addr), np.concatenate(dcal_all)
        n_struct += len(dcal_all)
        step = 150000                          # bounded host/device buffers per evaluation call
        for a in range(0, len(dcal_all), step):
            n = min(step, len(dcal_all) - a)
            sa, da = np.ascontiguousarray(seq_addr[a:a + n]), np.ascontiguousarray(db_addr[a:a + n])
            out, st = (C.c_int * n)(), (C.c_int * n)()
            N.check(lib.rafft_eval_structures(n, (C.c_char_p * n).from_buffer(sa), (C.c_char_p * n).from_buffer(da), out, st))
            assert not np.frombuffer(st, dtype=np.int32).any(), k
            assert (np.frombuffer(out, dtype=np.int32) == dcal_all[a:a + n]).all(), k
        del res
    assert n_struct > 3_000_000
    th.join()
    bad = [(i, len(seqs[i])) for i in sample if got_sample[i] != want[i]]
    assert not bad, bad


def test_gpu_cfg5_streamed_graph_text_and_sidecar(tmp_path):
    """BASELINE configs[4] through the CLI: one 400-nt sequence, beam 1000, `--traj` text and binary side-car streamed
    from the flat result buffers - equal to the oracle's trajectory formatted the reference's way, and
    readable by rafft_kin's readers"""
    rng = np.random.default_rng(400)
    s = "".join(rng.choice(list("ACGU"), 400))
    out, side = tmp_path / "ffg.out", tmp_path / "ffg.bin"
    run_cli("-s", s, "-ms", "1000", "--traj", "-o", str(out), "--sidecar", str(side))
    _, o = oracle.fold(s, 100, 1000, 1000, traj=True)
    want = utils.format_trajectory(s, [[utils.Structure(x.str_struct, x.dcal) for x in st] for st in o])
    assert out.read_text() == want
    fp, sq = utils.read_sidecar(str(side))
    assert sq == s and [[(x.str_struct, x.dcal) for x in st] for st in fp] == [[(x.str_struct, x.dcal) for x in st] for st in o]
    steps, seq = utils.parse_rafft_output(str(out))
    assert seq == s and [len(x) for x in steps] == [len(x) for x in o]


def test_gpu_batch_front_end_csv_in_bench_rows_out(tmp_path, bench_rows):
    """what benchmark_results/bench_fft.py does with a process pool - the benchmark CSV in, `--bench` rows
    of n=100 ms=50 out - as ONE process and one GPU batch (every 12th row here; the full set is the cfg3 test)"""
    rows = bench_rows[::12]
    csvf = tmp_path / "bench.csv"
    csvf.write_text("seq,struct,name\n" + "".join(f"{r['seq']},{r['known']},{r['name']}\n" for r in rows))
    outf = tmp_path / "rows.txt"
    run_cli("-sf", str(csvf), "--batch", "--bench", "-n", "100", "-ms", "50", "-o", str(outf))
    want = fold_many([(r["seq"], 100, 50, 1000, False) for r in rows])
    exp = "".join(f"{r['seq']} {len(r['seq'])} {db} {utils.Structure(db, d).energy:6.1f} {db.count('(')}\n"
                  for r, beam in zip(rows, want) for db, d in beam)
    assert outf.read_text() == exp
