"""CPU-only tests of the host side: C-ABI library loads and exports every declared symbol,
the Python mirror of the reference interface, the CLI formats, LPT sharding and the
world_size-2 gather path (gloo)."""
import ctypes
import io
import os
import re
import subprocess
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import _native, cli, sharding, utils
from conftest import GOLD, ROOT

EX = "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU"


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rafft_hip.h")).read()
    declared = set(re.findall(r"\b(rafft_[a-z_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = _native.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.rafft_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_native.Params) == 4 * 4 + 8 + 4 + 4 + 8 + 3 * 8
    assert ctypes.sizeof(_native.SeqResult) == 16 + 4 * 8
    assert ctypes.sizeof(_native.Stats) == 9 * 8 + 24 * 8


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.RafftError) as e:
        rafft_amd.fold("GGGAAACCC")
    assert e.value.code == _native.ERR_NO_DEVICE


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "rafft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")) and f != "turner2004_tables.h":
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "from oracle" not in txt, f


def test_structure_and_format_helpers(tmp_path):
    fin, traj = oracle.fold(EX, 100, 5, 1000, traj=True)
    mine = [[utils.Structure(s.str_struct, s.dcal) for s in st] for st in traj]
    txt = utils.format_trajectory(EX, mine)
    assert txt == open(os.path.join(GOLD, "example_rafft.out")).read()
    p = tmp_path / "x.out"
    p.write_text(txt)
    steps, seq = utils.parse_rafft_output(str(p))
    assert seq == EX and [len(s) for s in steps] == [len(s) for s in traj]
    assert steps[1][0].str_struct == traj[1][0].str_struct and abs(steps[1][0].energy - (-14.0)) < 1e-9
    s = mine[2][0]
    assert utils.dot_bracket(s.pair_list, len(EX)) == s.str_struct
    assert s.energy == float(np.float32(np.float32(s.dcal) / 100.0))


def _oracle_fold_batch(seqs, n_mode=100, max_stack=1, max_branch=100, min_hp=3, min_nrj=0.0, traj=False, temp=37.0,
                       gc=3.0, au=2.0, gu=1.0, **kw):
    return [oracle.fold(s, n_mode, max_stack, max_branch, min_hp, min_nrj, traj, temp, gc, au, gu) for s in seqs]


def test_cli_formats_match_reference(tmp_path):
    """final / --bench / --traj text formats of bin/rafft:59-79 (fold injected: no GPU here)"""
    def run(argv):
        buf = io.StringIO()
        with redirect_stdout(buf):
            cli.main(argv, fold_batch=_oracle_fold_batch)
        return buf.getvalue()
    assert run(["-s", EX, "-ms", "5", "--traj"]) == open(os.path.join(GOLD, "example_rafft.out")).read()
    fin = run(["-s", EX, "-ms", "5"]).splitlines()
    assert fin[0] == EX and fin[1].endswith(" -24.0") and len(fin) == 6
    b = run(["-s", EX, "-ms", "2", "--bench"]).splitlines()
    f = b[0].split()
    assert f[0] == EX and f[1] == "82" and f[3] == "-24.0" and int(f[4]) == f[2].count("(") and len(b) == 2
    fa = tmp_path / "s.fa"
    fa.write_text(">x\nGGGTTTGCGG\nTGTAAGTGCA\n")
    assert run(["-sf", str(fa)]).splitlines()[0] == "GGGUUUGCGGUGUAAGUGCA"
    with pytest.raises(AssertionError):
        cli.main([], fold_batch=_oracle_fold_batch)


def test_lpt_shards_balanced_and_complete():
    rng = np.random.default_rng(1)
    lens = list(rng.integers(28, 3000, size=500))
    for n in (1, 2, 4, 8):
        sh = sharding.lpt_shards(lens, n)
        assert sorted(i for s in sh for i in s) == list(range(len(lens)))
        loads = [sum(lens[i] ** 2 for i in s) for s in sh]
        assert max(loads) <= 1.15 * (sum(loads) / n) + max(lens) ** 2


WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import torch.distributed as dist
import oracle
from rafft_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
seqs = json.load(open({seqfile!r}))
fold = lambda ss, **kw: [oracle.fold(s, 100, 5, 1000) for s in ss]
res = sharding.fold_sharded(seqs, fold_fn=fold)
if dist.get_rank() == 0:
    json.dump([[(x.str_struct, x.dcal) for x in r] for r in res], open({outfile!r}, "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_world_size_2_gloo_sharded_fold(tmp_path):
    """N>1 path: two ranks, LPT shards, gather on rank 0 - results identical to a single-rank run"""
    import json
    rng = np.random.default_rng(2)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(20, 120, size=12)]
    seqfile, outfile = tmp_path / "seqs.json", tmp_path / "out.json"
    seqfile.write_text(json.dumps(seqs))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, seqfile=str(seqfile), outfile=str(outfile)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = json.loads(outfile.read_text())
    want = [[(x.str_struct, x.dcal) for x in oracle.fold(s, 100, 5, 1000)] for s in seqs]
    assert [[tuple(x) for x in r] for r in got] == want


def test_kinetics_matches_reference_python():
    """rafft_amd.rafft_kin.kinetics == the reference's kinetics on its example fast-folding graphs"""
    from conftest import load_json_gz
    from rafft_amd import rafft_kin
    gold = load_json_gz("kinetics.json.gz")
    for name, g in gold.items():
        fp, seq = utils.parse_rafft_output(os.path.join(GOLD, name))
        traj, times, sl, eq = rafft_kin.kinetics(fp, g["max_time"], g["n_steps"])
        assert [s.str_struct for s in sl] == g["struct_list"]
        np.testing.assert_allclose(np.array(times, dtype=float), np.array(g["times"]), rtol=1e-14)
        np.testing.assert_allclose(np.array([np.asarray(r, dtype=float) for r in traj]), np.array(g["trajectory"]), rtol=1e-9, atol=1e-12)
        assert [(e[0], e[3]) for e in eq] == [(e[0], e[3]) for e in g["equi"]]


def test_rafft_kin_cli_table(capsys):
    from rafft_amd import rafft_kin
    rafft_kin.main([os.path.join(GOLD, "example_rafft_20.out"), "-mt", "40"])
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 68
    top = lines[-1].split()
    assert top[0].startswith("(((((.(((") and abs(float(top[1]) - 0.519) < 1e-3 and top[3] == "59"


def test_fast_folding_graph_sidecar_roundtrip(tmp_path, capsys):
    """`rafft --traj --sidecar` writes the graph in binary (exact dcal); `rafft_kin --sidecar`
    reads it and prints the same population table as from the text"""
    from rafft_amd import rafft_kin
    side = tmp_path / "ffg.bin"
    buf = io.StringIO()
    with redirect_stdout(buf):
        cli.main(["-s", EX, "-ms", "20", "--traj", "--sidecar", str(side)], fold_batch=_oracle_fold_batch)
    assert buf.getvalue() == open(os.path.join(GOLD, "example_rafft_20.out")).read()
    fp_bin, seq_bin = utils.read_sidecar(str(side))
    fp_txt, seq_txt = utils.parse_rafft_output(os.path.join(GOLD, "example_rafft_20.out"))
    assert seq_bin == seq_txt == EX
    assert [[s.str_struct for s in st] for st in fp_bin] == [[s.str_struct for s in st] for st in fp_txt]
    _, o = oracle.fold(EX, 100, 20, 1000, traj=True)
    assert [[s.dcal for s in st] for st in fp_bin] == [[s.dcal for s in st] for st in o]       # exact, not 1 decimal
    capsys.readouterr()
    rafft_kin.main([os.path.join(GOLD, "example_rafft_20.out"), "-mt", "40"])
    from_text = capsys.readouterr().out
    rafft_kin.main([str(side), "--sidecar", "-mt", "40"])
    from_bin = capsys.readouterr().out
    t, b2 = from_text.strip().splitlines(), from_bin.strip().splitlines()
    assert from_bin == from_text and len(t) == len(b2) == 68 and b2[-1].split()[3] == "59"   # same one-decimal energies by default
    rafft_kin.main([str(side), "--sidecar", "--exact", "-mt", "40"])
    assert len(capsys.readouterr().out.strip().splitlines()) == 68
    (tmp_path / "bad.bin").write_bytes(b"nonsense")
    with pytest.raises(ValueError):
        utils.read_sidecar(str(tmp_path / "bad.bin"))


def test_scoring_reproduces_reference_columns(bench_rows):
    """PPV / sensitivity of the reference's published structures against the known structures:
    the flexible-pair rule reproduces the pvv/sens columns of its *_scores.csv (2 decimals)"""
    from rafft_amd import scoring
    bad = 0
    for r in bench_rows:
        for key in ("best", "ppv", "ppv200"):
            p, s = scoring.score(r[key][0], r["known"])
            a, b = r[key + "_scores"]
            if abs(p - a) > 0.006 or abs(s - b) > 0.006:
                bad += 1
    assert bad <= 3, bad        # one benchmark entry's CT file differs from the CSV's known structure


def test_kinetics_solvers_against_high_precision_truth():
    """the two solvers behind kinetics_gpu (run here on the CPU through torch) against 60-digit arithmetic on the
    reference's example graphs (tests/golden/kinetics_truth.json.gz, tools/make_kinetics_truth.py) - and the reference's
    own float64 eig/inv output against the same truth: it is exact early and off by ~0.5 at the end, which is why
    the late-time check of the GPU path cannot be the reference's numbers"""
    import torch
    from conftest import load_json_gz
    from rafft_amd import rafft_kin
    truth, gold = load_json_gz("kinetics_truth.json.gz"), load_json_gz("kinetics.json.gz")
    for name, tr in truth.items():
        fp, seq = utils.parse_rafft_output(os.path.join(GOLD, name))
        sl, index = rafft_kin.unique_structures(fp)
        sm = {st.str_struct: (index[st.str_struct], st.energy) for st in sl}
        rate = torch.as_tensor(np.asarray(rafft_kin.get_transition_mat(fp, len(sl), sm), dtype=np.float64))
        en = np.array([st.energy for st in sl])
        p0 = torch.zeros(len(sl), dtype=torch.float64)
        p0[0] = 1.0
        times = np.exp(np.arange(tr["n_steps"]) * (tr["max_time"] / tr["n_steps"]) - 4)
        ks = tr["sample_index"]
        want = np.array(tr["populations"])
        early = [i for i, k in enumerate(ks) if k <= 0.6 * tr["n_steps"]]
        imp = rafft_kin.solve_master_equation(rate, en, p0, times, "implicit", 32)[ks]
        spe = rafft_kin.solve_master_equation(rate, en, p0, times, "spectral")[ks]
        assert np.abs(imp - want)[early].max() < 5e-6 and np.abs(imp - want).max() < 2e-2
        assert np.abs(spe - want)[early].max() < 1e-6
        assert imp.min() > -1e-12 and np.allclose(imp.sum(axis=1), 1.0)
        ref = np.array(gold[name]["trajectory"])[1:][ks]
        assert np.abs(ref - want)[early].max() < 1e-6
        if name == "example_rafft_20.out":
            assert np.abs(ref - want).max() > 0.3                       # the reference's own late-time numbers are noise
            assert abs(want[-1].max() - 0.5316) < 1e-4                  # = README.org:146 (0.531)


def _raw_of(traj_or_beam, traj):
    steps = traj_or_beam if traj else [traj_or_beam]
    rows = np.array([list(s.str_struct.encode()) for st in steps for s in st], dtype=np.uint8)
    L = len(steps[0][0].str_struct)
    return L, [len(st) for st in steps], rows.reshape(-1, L), np.array([s.dcal for st in steps for s in st], dtype=np.int32)


def test_streaming_writers_equal_the_reference_formats(tmp_path):
    """the text formats of bin/rafft:59-79 and the side-car written straight from flat result buffers
    (no Structure objects) are byte-identical to the per-structure formatting"""
    import argparse
    rng = np.random.default_rng(9)
    seqs = [EX, "".join(rng.choice(list("ACGU"), 150)), "GGGAAACCC"]
    for s in seqs:
        fin, traj = oracle.fold(s, 100, 12, 1000, traj=True)
        for mode in ("final", "bench", "traj"):
            args = argparse.Namespace(traj=mode == "traj", bench=mode == "bench")
            res = (fin, traj) if mode == "traj" else fin
            mine = [[utils.Structure(x.str_struct, x.dcal) for x in st] for st in traj]
            want = cli.format_result(s, (mine[-1], mine) if mode == "traj" else mine[-1], args) + "\n"
            buf = io.BytesIO()
            utils.write_result_text(buf, s, _raw_of(traj if mode == "traj" else fin, mode == "traj"), traj=args.traj, bench=args.bench)
            assert buf.getvalue().decode() == want, (mode, len(s))
        side = tmp_path / "s.bin"
        utils.write_sidecar_raw(str(side), s, _raw_of(traj, True))
        fp, sq = utils.read_sidecar(str(side))
        assert sq == s and [[(x.str_struct, x.dcal) for x in st] for st in fp] == [[(x.str_struct, x.dcal) for x in st] for st in traj]
    # ragged columns: pair counts of one and two digits, an energy that needs seven characters
    rows = np.array([list(b"((((((((((....))))))))))"), list(b"((((((((........))))))))"), list(b"........................")], dtype=np.uint8)
    raw = (24, [3], rows, np.array([-100010, -1230, 0], dtype=np.int32))
    buf = io.BytesIO()
    utils.write_result_text(buf, "A" * 24, raw, bench=True)
    st = [utils.Structure(bytes(r).decode(), d) for r, d in zip(rows, raw[3])]
    assert buf.getvalue().decode() == cli.format_result("A" * 24, st, argparse.Namespace(traj=False, bench=True)) + "\n"
    buf = io.BytesIO()
    utils.write_result_text(buf, "A" * 24, raw)
    assert buf.getvalue().decode() == cli.format_result("A" * 24, st, argparse.Namespace(traj=False, bench=False)) + "\n"


def test_batch_front_end_reads_csv_fasta_and_lines(tmp_path):
    """`rafft -sf FILE --batch --bench [-o OUT]` replaces benchmark_results/bench_fft.py:8-22 - CSV with the
    reference's `seq` column (benchmark_cleaned_all_length.csv), FASTA records or one sequence per line in; one
    `seq len structure energy #pairs` row per structure out"""
    seqs = [EX, "GGGGAAAACCCC", "ACGUACGUACGUUUUACGUACGUACGU"]
    csvf = tmp_path / "b.csv"
    csvf.write_text("seq,struct,name\n" + "".join(f"{s.replace('U', 'T') if k == 1 else s},{'.' * len(s)},n{k}\n" for k, s in enumerate(seqs)))
    outf = tmp_path / "rows.txt"
    cli.main(["-sf", str(csvf), "--batch", "--bench", "-ms", "4", "-o", str(outf)], fold_batch=_oracle_fold_batch)
    want = ""
    for s in seqs:
        for x in oracle.fold(s, 100, 4, 1000):
            want += f"{s} {len(s)} {x.str_struct} {x.energy:6.1f} {x.str_struct.count('(')}\n"
    assert outf.read_text() == want
    fa = tmp_path / "b.fa"
    fa.write_text("".join(f">r{k}\n{s[:20]}\n{s[20:]}\n" for k, s in enumerate(seqs)))
    out2 = tmp_path / "rows2.txt"
    cli.main(["-sf", str(fa), "--batch", "--bench", "-ms", "4", "-o", str(out2)], fold_batch=_oracle_fold_batch)
    assert out2.read_text() == want
    ln = tmp_path / "b.txt"
    ln.write_text("\n".join(seqs) + "\n")
    out3 = tmp_path / "rows3.txt"
    cli.main(["-sf", str(ln), "--batch", "--bench", "-ms", "4", "-o", str(out3)], fold_batch=_oracle_fold_batch)
    assert out3.read_text() == want
    with pytest.raises(SystemExit):
        cli.main(["-sf", str(csvf), "--batch", "--csv_column", "nope"], fold_batch=_oracle_fold_batch)


def test_round5_device_helpers_on_the_host(tmp_path):
    """the telescoping pair hash (a stem = two mixes, a pair set hashes the same whatever stems it is assembled from, no collision on the
    pattern a polynomial hash would confuse), the packed-strand stacking table against the plain table for every quadruple of bases and
    random stems, and the special-hairpin filter never hiding a listed loop - rafft_device.h's host-callable forms, built host-only"""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    exe = str(tmp_path / "device_helpers_check")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-host-only", "-O1", "-std=c++17", "-Wno-unused-function", "-Wno-missing-braces",
                           os.path.join(ROOT, "tests", "hostcheck", "device_helpers_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr


def test_bench_self_launch_ends_the_run_when_a_rank_dies(tmp_path):
    """a rank that dies at start-up ends the whole run at once with a non-zero exit (its siblings are terminated instead of sitting in the
    rendezvous until the process group's timeout) and its stderr is shown"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_SAME_GPU="1", BENCH_BACKEND="gloo", BENCH_SKIP_CFG4="1", BENCH_TEST_DIE_RANK="1", BENCH_LAUNCH_TIMEOUT_S="120")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "ranks failed" in r.stderr and "BENCH_TEST_DIE_RANK" in r.stderr
    assert time.time() - t0 < 100


def test_one_config_struct_is_the_only_reader_of_the_environment(tmp_path):
    """every environment switch of the library is a field of Config (rafft_config.h), read_config() there is the ONLY getenv
    caller of the library, defaults / parsing / snapshot equality hold, and the test hooks compile out with -DRAFFT_NO_TEST_HOOKS"""
    import subprocess
    csrc = os.path.join(ROOT, "rafft_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".h")) and name != "rafft_config.h":
            assert "getenv" not in open(os.path.join(csrc, name)).read(), name
    cfg = open(os.path.join(csrc, "rafft_config.h")).read()
    fields = set(re.findall(r"//\s+(RAFFT_[A-Z0-9_]+)", cfg))
    read = set(re.findall(r'"(RAFFT_[A-Z0-9_]+)"', cfg))
    assert read and read <= fields | {"RAFFT_TAPER", "RAFFT_C2_FFT", "RAFFT_SCHED_NAP_US"}, sorted(read - fields)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "rafft_config.h" in doc and "RAFFT_NO_TEST_HOOKS" in doc
    for flags in ([], ["-DRAFFT_NO_TEST_HOOKS"]):
        exe = str(tmp_path / ("config_check" + str(len(flags))))
        subprocess.check_call(["g++", "-std=c++17", "-O1"] + flags + [os.path.join(ROOT, "tests", "hostcheck", "config_check.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr
