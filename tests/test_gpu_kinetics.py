"""rafft_kin on the GPU: the pair-set inclusion search + rate matrix as HIP kernels
(rafft_kin_rate_matrix) against the host mirror of the reference's get_transition_mat, and the dense solves on the
device against 60-digit arithmetic and the reference's own output."""
import os
import time

import numpy as np
import pytest

import oracle
import rafft_amd
from rafft_amd import rafft_kin, utils
from conftest import GOLD, load_json_gz

pytestmark = pytest.mark.gpu


def host_rate(fp):
    sl, index = rafft_kin.unique_structures(fp)
    sm = {st.str_struct: (index[st.str_struct], st.energy) for st in sl}
    return np.asarray(rafft_kin.get_transition_mat(fp, len(sl), sm), dtype=np.float64), sl


def check_rate_matrix(fp):
    want, sl = host_rate(fp)
    got, sl2, en = rafft_kin.rate_matrix_gpu(fp)
    got = got.cpu().numpy()
    assert [s.str_struct for s in sl] == [s.str_struct for s in sl2]
    assert np.array_equal(got != 0, want != 0)                     # the same connections (pair-set inclusion)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)      # the same Metropolis rates (device exp vs libm: ulps)
    return got


@pytest.mark.parametrize("name", ["example_rafft_20.out", "example_rafft.out"])
def test_gpu_rate_matrix_and_populations_on_reference_examples(name):
    fp, seq = utils.parse_rafft_output(os.path.join(GOLD, name))
    check_rate_matrix(fp)
    g, tr = load_json_gz("kinetics.json.gz")[name], load_json_gz("kinetics_truth.json.gz")[name]
    ks = tr["sample_index"]
    truth = np.array(tr["populations"])
    early = [i for i, k in enumerate(ks) if k <= 0.6 * tr["n_steps"]]
    for method in ("spectral", "implicit", "auto"):
        traj, times, sl, eq = rafft_kin.kinetics_gpu(fp, g["max_time"], g["n_steps"], method=method)
        assert [s.str_struct for s in sl] == g["struct_list"]
        np.testing.assert_allclose(np.array(times, dtype=float), np.array(g["times"]), rtol=1e-14)
        P = np.array(traj)[1:]
        # early and middle times: the reference's own output is accurate there, and so is the 60-digit truth
        assert np.abs(P[ks][early] - truth[early]).max() < 5e-6, method
        assert np.abs(P[: int(0.6 * g["n_steps"])] - np.array(g["trajectory"])[1:][: int(0.6 * g["n_steps"])]).max() < 5e-6, method
        assert P.min() > -1e-9 and np.allclose(P.sum(axis=1), 1.0)
        if method != "spectral":
            assert np.abs(P[ks] - truth).max() < 2e-2               # late times: the integrator stays close to the truth
            assert int(np.argmax(P[-1])) == int(np.argmax(truth[-1]))


def test_gpu_kinetics_cli_table(capsys):
    rafft_kin.main([os.path.join(GOLD, "example_rafft_20.out"), "-mt", "40", "--gpu"])
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 68
    top = lines[-1].split()
    assert abs(float(top[1]) - 0.531) < 0.01                       # README.org:146 prints 0.531 for the most populated structure


def test_gpu_cfg5_graph_rate_matrix_and_kinetics():
    """BASELINE configs[4]: one 400-nt sequence, beam 1000, its fast-folding graph through rafft_kin.
    The rate matrix of the ms=300 graph equals the host mirror entry by entry; on the ms=1000 graph (thousands of
    structures) it has the properties the master equation needs, and the populations come out of the device solver
    non-negative, normalised and heading for the Boltzmann distribution of the reachable structures."""
    rng = np.random.default_rng(400)
    s = "".join(rng.choice(list("ACGU"), 400))
    fin, traj = rafft_amd.fold(s, 100, 300, 1000, traj=True)
    check_rate_matrix(traj)
    t0 = time.time()
    fin, traj = rafft_amd.fold(s, 100, 1000, 1000, traj=True)
    t_fold = time.time() - t0
    t0 = time.time()
    rate, sl, en = rafft_kin.rate_matrix_gpu(traj)
    t_rate = time.time() - t0
    S = len(sl)
    assert S > 3000
    R = rate.cpu().numpy()
    off = R - np.diag(np.diag(R))
    assert off.min() >= 0 and off.max() <= 1.0 and np.abs(R.sum(axis=1)).max() < 1e-9       # generator matrix
    i, j = np.nonzero(off)
    lhs = off[i, j] * np.exp(-(en[i] - en.min()) / 0.61)                                        # detailed balance
    rhs = off[j, i] * np.exp(-(en[j] - en.min()) / 0.61)
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10, atol=1e-300)
    t0 = time.time()
    trj, times, sl2, eq = rafft_kin.kinetics_gpu(traj, 30, 40, substeps=8)
    t_kin = time.time() - t0
    P = np.array(trj)[1:]
    assert P.min() > -1e-9 and np.allclose(P.sum(axis=1), 1.0)
    assert P[0][0] < 1e-3 < P[0][1:1001].sum()                     # ~1000 downhill neighbours: the unfolded state empties at rate ~1000
    free = lambda p: float((p * (en + 0.61 * np.log(np.maximum(p, 1e-300)))).sum())
    f = [free(p) for p in P]
    assert all(b <= a + 1e-6 for a, b in zip(f, f[1:]))            # the free energy of the ensemble never rises
    # the sparse integrator (host, SuperLU on the non-zeros) and the dense one (device, rocSOLVER getrf) are the same scheme
    ta = np.array(rafft_kin.kinetics_gpu(traj, 30, 6, method="implicit", substeps=4)[0])
    tb = np.array(rafft_kin.kinetics_gpu(traj, 30, 6, method="implicit-dense", substeps=4)[0])
    assert np.abs(ta - tb).max() < 1e-7
    print(f"cfg5: fold {t_fold * 1e3:.1f} ms, {S} structures, rate matrix {t_rate * 1e3:.1f} ms, populations (40 times) {t_kin:.2f} s")
