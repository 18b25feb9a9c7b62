"""Oracle folds in worker processes (test infrastructure).

The oracle is single-threaded C; the full-size parity tests need it on thousands of
sequences.  Workers are plain child processes (`python -c ...`, tasks pickled over
stdin, results over stdout), never forks of the pytest process - that one has usually
initialised HIP by the time a pool is needed - and they only ever import the oracle:
they never touch the GPU."""
import os
import pickle
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def n_workers():
    # a GPU box job owns a 16-core share; the build container has 8 cores
    return max(1, min(len(os.sched_getaffinity(0)), 16))


def _fold(task):
    import oracle
    seq, nb_mode, ms, mb, traj = task
    r = oracle.fold(seq, nb_mode, ms, mb, traj=traj)
    if traj:
        return [[(x.str_struct, x.dcal) for x in st] for st in r[1]]
    return [(x.str_struct, x.dcal) for x in r]


def worker_main():
    tasks = pickle.load(sys.stdin.buffer)
    out = [_fold(t) for t in tasks]
    sys.stdout.buffer.write(pickle.dumps(out))
    sys.stdout.buffer.flush()


def fold_many(tasks, workers=None):
    """tasks: (seq, nb_mode, max_stack, max_branch, traj) tuples -> list of beams
    [(db, dcal), ...] or, with traj, lists of beams; results in task order."""
    if not tasks:
        return []
    import oracle
    oracle.oracle.build()                  # once, before the workers race to build it
    nw = min(workers or n_workers(), len(tasks))
    # greedy longest-first assignment (cost ~ L^2 * beam), so no worker is left with the long tail
    order = sorted(range(len(tasks)), key=lambda i: -(len(tasks[i][0]) ** 2) * tasks[i][2])
    load, parts = [0.0] * nw, [[] for _ in range(nw)]
    for i in order:
        w = load.index(min(load))
        parts[w].append(i)
        load[w] += (len(tasks[i][0]) ** 2 + 2000.0) * tasks[i][2]
    code = f"import sys; sys.path[:0] = [{ROOT!r}, {HERE!r}]; import _oracle_pool; _oracle_pool.worker_main()"
    procs = []
    for part in parts:
        p = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        p.stdin.write(pickle.dumps([tasks[i] for i in part]))
        p.stdin.close()
        procs.append(p)
    out = [None] * len(tasks)
    for part, p in zip(parts, procs):
        data = p.stdout.read()
        assert p.wait() == 0, "oracle worker failed"
        for i, r in zip(part, pickle.loads(data)):
            out[i] = r
    return out
