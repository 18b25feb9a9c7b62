"""Continuous batching (rafft_fold_submit / rafft_fold_wait): several batches in flight, driven by the library's one
scheduler thread, give exactly the results of synchronous calls - whatever is in flight beside them."""
import threading

import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import _native, rafft as R

pytestmark = pytest.mark.gpu


def key(res, traj):
    if traj:
        return [[[(s.str_struct, s.dcal) for s in st] for st in t] for _, t in res]
    return [[(s.str_struct, s.dcal) for s in beam] for beam in res]


def make_batches(bench_rows):
    rng = np.random.default_rng(17)
    rnd = lambda lens: ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
    long_tail = rnd(rng.integers(30, 150, size=300)) + rnd([1500, 2100])
    return [
        (dict(nb_mode=100, max_stack=50, max_branch=1000, traj=False), [r["seq"] for r in bench_rows[::3]]),
        (dict(nb_mode=100, max_stack=10, max_branch=200, traj=True), long_tail),
        (dict(nb_mode=100, max_stack=20, max_branch=1000, traj=False), rnd(rng.integers(60, 400, size=500))),
        (dict(nb_mode=30, max_stack=5, max_branch=50, traj=True), rnd(rng.integers(10, 90, size=40))),
        (dict(nb_mode=100, max_stack=50, max_branch=1000, traj=False), [r["seq"] for r in bench_rows[1::3]]),
        (dict(nb_mode=100, max_stack=200, max_branch=1000, traj=False), rnd([700, 900, 1200])),
    ]


def test_gpu_batches_in_flight_equal_synchronous_calls(bench_rows):
    batches = make_batches(bench_rows)
    want = [key(rafft_amd.fold_batch(seqs, **kw), kw["traj"]) for kw, seqs in batches]
    # all in flight at once, waited for in reverse order
    pend = [rafft_amd.submit_batch(seqs, **kw) for kw, seqs in batches]
    got = [None] * len(batches)
    for i in reversed(range(len(batches))):
        got[i] = key(pend[i].result(), batches[i][0]["traj"])
    assert got == want
    # a rolling window of two (the bench loop), several rounds
    q, got2 = [], []
    for rnd_ in range(3):
        for kw, seqs in batches:
            q.append((rafft_amd.submit_batch(seqs, **kw), kw["traj"]))
            if len(q) >= 2:
                pb, tr = q.pop(0)
                got2.append(key(pb.result(), tr))
    while q:
        pb, tr = q.pop(0)
        got2.append(key(pb.result(), tr))
    assert got2 == want * 3
    # and a sample against the oracle
    kw, seqs = batches[3]
    for s, (fin, traj) in zip(seqs[:10], rafft_amd.fold_batch(seqs, **kw)):
        _, o = oracle.fold(s, kw["nb_mode"], kw["max_stack"], kw["max_branch"], traj=True)
        assert [[(x.str_struct, x.dcal) for x in st] for st in traj] == [[(x.str_struct, x.dcal) for x in st] for st in o]


def test_gpu_calls_from_several_threads_and_seam_calls_between(bench_rows):
    """fold_batch from three host threads at once (ctypes releases the GIL), whole-structure evaluations (which borrow
    a workspace: they wait for the batches in flight) and a failing submission in between"""
    batches = make_batches(bench_rows)[:3]
    want = [key(rafft_amd.fold_batch(seqs, **kw), kw["traj"]) for kw, seqs in batches]
    got = [None] * 3
    errs = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = key(rafft_amd.fold_batch(batches[i][1], **batches[i][0]), batches[i][0]["traj"])
        except Exception as e:        # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    flat = [(s, beam[0][0]) for s, beam in zip(batches[0][1][:50], want[0][:50])]
    for _ in range(3):
        e, st = R.eval_structures([f[0] for f in flat], [f[1] for f in flat])
        assert not any(st) and e == [w[0][1] for w in want[0][:50]]
        with pytest.raises(_native.RafftError):
            rafft_amd.submit_batch(["GGGAAACCC"], max_stack=0)
    for t in th:
        t.join()
    assert not errs and got == want


def test_gpu_unwaited_job_is_released():
    pb = rafft_amd.submit_batch(["GGGGAAAACCCC"] * 4, max_stack=3)
    del pb                                   # PendingBatch.__del__ waits and frees
    assert rafft_amd.fold("GGGGAAAACCCC")[0].str_struct == "((((....))))"


def test_gpu_queued_batches_with_equal_parameters_are_merged_and_split_back(bench_rows):
    """continuous batching: batches queued behind a running wave are folded as ONE wave when their parameters are identical;
    every batch still gets exactly its own result (rows live in pinned chunks shared between the merged batches), bad
    sequences stay per batch, and the statistics of the merged wave are counted once"""
    rng = np.random.default_rng(23)
    base = [r["seq"] for r in bench_rows[::4]]
    variants = [base, base[::-1], base[:200] + ["ACGT", ""] + base[200:], ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(40, 300, size=300)],
                base[100:400], base]
    kw = dict(nb_mode=100, max_stack=30, max_branch=1000, traj=False)
    want = [key(rafft_amd.fold_batch(v, raise_errors=False, **kw), False) if "ACGT" not in v else None for v in variants]
    single_structs = []
    for v in variants:
        rafft_amd.fold_batch(v, raise_errors=False, **kw)
        single_structs.append(rafft_amd.last_stats()["n_structs"])
    for rnd_ in range(3):
        pend = [rafft_amd.submit_batch(v, raise_errors=False, **kw) for v in variants]
        total = 0
        for i, pb in enumerate(pend):
            res = pb.result()
            total += rafft_amd.last_stats()["n_structs"]
            if want[i] is None:
                assert res[200] is None and res[201] is None and res.status(200) == _native.ERR_BAD_CHAR and res.status(201) == _native.ERR_EMPTY
                ok = [k for k in range(len(variants[i])) if k not in (200, 201)]
                assert [[(s.str_struct, s.dcal) for s in res[k]] for k in ok] == key(rafft_amd.fold_batch([variants[i][k] for k in ok], **kw), False)
            else:
                assert key(res, False) == want[i], (rnd_, i)
        assert total == sum(single_structs)          # every wave's work is attributed to exactly one of its batches


def test_gpu_steady_stream_of_equal_batches_allocates_nothing_after_its_first_waves(bench_rows):
    """Workspaces are sized at first use for the biggest wave the scheduler may merge from such batches (five of them, up to the
    merge cap) and are kept: whatever way the later rounds are merged, the library allocates no device buffer (a hipMalloc of
    gigabytes takes seconds now and then - tools/micro/malloc_busy.hip).  rafft_alloc_counters() is what bench.py reports."""
    import ctypes as C
    lib = _native.lib()

    def counters():
        a = (C.c_ulonglong * 5)()
        lib.rafft_alloc_counters(C.byref(a))
        return list(a)

    seqs = [r["seq"] for r in bench_rows if len(r["seq"]) <= 400][:600]
    kw = dict(nb_mode=100, max_stack=20, max_branch=1000, traj=False)
    want = key(rafft_amd.fold_batch(seqs, **kw), False)

    def round_(n, depth):
        pend = []
        for _ in range(n):
            pend.append(rafft_amd.submit_batch(seqs, **kw))
            if len(pend) >= depth:
                assert key(pend.pop(0).result(), False) == want
        for pb in pend:
            assert key(pb.result(), False) == want

    for _ in range(3):             # first waves: both workspaces for bulk waves meet a merged wave
        round_(8, 8)
    before = counters()
    assert before[0] > 0 and before[1] > 0 and before[3] > 0
    for n, depth in ((8, 8), (5, 2), (7, 4), (1, 1), (6, 6)):
        round_(n, depth)
    after = counters()
    assert after[0] == before[0] and after[1] == before[1], (before, after)
    assert after[2] >= before[2]


def test_gpu_merged_wave_overflow_regrows_and_stays_exact(bench_rows, monkeypatch):
    """a wave that serves several batches and overflows its (starved) HBM arenas is re-run with larger ones - for all of
    its batches - and every batch still gets exactly its result"""
    base = [r["seq"] for r in bench_rows[::6]]
    variants = [base, base[50:250], base[::-1], base[:100]]
    kw = dict(nb_mode=100, max_stack=20, max_branch=1000, traj=True)
    want = [key(rafft_amd.fold_batch(v, **kw), True) for v in variants]
    monkeypatch.setenv("RAFFT_EST", "0.05")
    regrows = 0
    for rnd_ in range(2):
        pend = [rafft_amd.submit_batch(v, **kw) for v in variants]
        for i, pb in enumerate(pend):
            assert key(pb.result(), True) == want[i], (rnd_, i)
            regrows += rafft_amd.last_stats()["n_regrows"]
    assert regrows > 0
    monkeypatch.delenv("RAFFT_EST")
    monkeypatch.setenv("RAFFT_TEST_OVF_AT", "5")       # pretend an overflow late in the first attempt of every job
    pend = [rafft_amd.submit_batch(v, **kw) for v in variants]
    for i, pb in enumerate(pend):
        assert key(pb.result(), True) == want[i]


def test_gpu_hard_failure_of_one_batch_leaves_the_others_whole(monkeypatch):
    """(a) a wave that folds several batches - merged by the scheduler because their parameters are equal - and hits a
    hard error: every member is folded again on its own and none of them pays for it; (b) a batch one of whose jobs
    failed: its other, still queued job is dropped, the batches queued behind it are folded completely (they used to come
    back as empty successes)"""
    rng = np.random.default_rng(29)
    rnd = lambda lens: ["".join(rng.choice(list("ACGU"), int(n))) for n in lens]
    A, B = rnd(rng.integers(30, 120, size=300)), rnd(rng.integers(30, 120, size=340))
    kw = dict(nb_mode=100, max_stack=10, max_branch=200)
    want_a, want_b = key(rafft_amd.fold_batch(A, **kw), False), key(rafft_amd.fold_batch(B, **kw), False)
    # (a) one wave at a time: while a blocker folds, A and B queue up and are merged into one wave of 640 sequences
    monkeypatch.setenv("RAFFT_MAX_WAVES", "1")
    monkeypatch.setenv("RAFFT_TEST_HARD_FAIL", "640")
    _native.lib().rafft_shutdown()          # the scheduler thread reads RAFFT_MAX_WAVES when it starts: stop it, the next submit starts a fresh one
    blocker = rafft_amd.submit_batch(rnd(rng.integers(200, 400, size=400)), nb_mode=100, max_stack=50, max_branch=1000)
    pa, pb = rafft_amd.submit_batch(A, **kw), rafft_amd.submit_batch(B, **kw)
    blocker.result()
    assert key(pa.result(), False) == want_a and key(pb.result(), False) == want_b
    # (b) A2 = A + two long sequences: cut in a long-tail job (2 sequences: fails by the hook) and a bulk job
    monkeypatch.setenv("RAFFT_TEST_HARD_FAIL", "2")
    A2 = A + rnd([1500, 1700])
    blocker = rafft_amd.submit_batch(rnd(rng.integers(200, 400, size=400)), nb_mode=100, max_stack=50, max_branch=1000)
    pa, pb = rafft_amd.submit_batch(A2, **kw), rafft_amd.submit_batch(B, **kw)
    blocker.result()
    with pytest.raises(Exception) as ei:
        pa.result()
    assert "hard failure" in str(ei.value)
    assert key(pb.result(), False) == want_b
    monkeypatch.delenv("RAFFT_TEST_HARD_FAIL")
    monkeypatch.delenv("RAFFT_MAX_WAVES")
    _native.lib().rafft_shutdown()
    assert key(rafft_amd.fold_batch(A2, **kw), False)[:300] == want_a
