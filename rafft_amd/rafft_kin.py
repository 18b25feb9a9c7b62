"""Kinetics on the fast-folding graph (a "next" row: post-processing of the fold's output).

`kinetics` is the host-side mirror of the reference's rafft/rafft_kin.py (`kinetics`, `get_transition_mat`): same
inputs, outputs, Metropolis rates, quirks and the same SciPy eig/inv solve; only the O(steps * ms^2 * L) Python
set-inclusion search of `get_connected_prev` (rafft_kin.py:48-56) is vectorised over pair tables - an exact equivalent.

`kinetics_gpu` is the MI355X path for big graphs (ms = 1000: ~10^4 structures): the inclusion search and the rate
matrix are one HIP kernel family (rafft_amd/csrc/rafft_kin.hip, C-ABI rafft_kin_rate_matrix) and the matrix stays in
device memory.  Where the solve runs depends on the solver: the spectral formula and "implicit-dense" run on the device
through rocSOLVER (torch.linalg as the binding); "implicit" on a sparse generator - every fast-folding graph - pulls the
non-zeros off the device and factorises them on the HOST with SciPy's SuperLU (40 sparse LUs of the configs[4] graph:
1.2 s against 11.8 s for 40 dense getrf of 6001^2 on the device); without SciPy, and for dense matrices, it is the dense
device path.  Two solvers:
  * "spectral" - the reference's formula p(t) = W exp(Vt) W^-1 p0, on the symmetrised matrix (the Metropolis rates
    satisfy detailed balance, so D^-1/2 A D^1/2 with D = diag(exp(-E/KT)) is symmetric: real eigenvalues, orthogonal
    eigenvectors, no inverse).  Like the reference's eig/inv it loses sqrt(pi_max/pi_min) * 1e-16 of accuracy, i.e. it is
    UNRELIABLE once the energies of the graph span more than a few dozen KT: on the reference's own example
    (example/rafft_20.out, 42 KT) its late-time populations are off by up to 0.48 against 60-digit arithmetic
    (tools/make_kinetics_truth.py; the README's 0.531 is the exact value, the current code prints 0.519), and on a 400-nt
    graph it returns negative populations (-0.82).
  * "implicit" - TR-BDF2 (second order, L-stable, positivity-friendly) from each output time to the next, one LU
    factorisation and a few dozen triangular solves per output interval.  No cancellation: valid for any energy span.
    The generator is sparse (a few non-zeros per row): sparse LU of the non-zeros pulled out of the device matrix
    ("implicit-dense" forces rocSOLVER's dense getrf, which is what a dense graph gets anyway).
    In float64 its late-time accuracy is limited by the conditioning of (I - chA) at huge h (~1e-2 at t = e^36 on the
    example, where the reference is off by 0.48).
`method="auto"` takes the spectral formula for energy spans up to 30 KT and the integrator beyond."""
from copy import deepcopy

import numpy as np
from numpy import array, diag, exp, zeros

from .utils import paired_positions

KT = 0.61


def _pair_table(db):
    pt = np.full(len(db), -1, dtype=np.int32)
    for i, j in paired_positions(db):
        pt[i] = j
        pt[j] = i
    return pt


def get_transition_mat(fast_paths, nb_struct, struct_map):
    """rafft_kin.py:68-91.  Note the reference's quirk: step 0 is compared with the LAST step
    (`fast_paths[step_i - 1]` with step_i == 0)."""
    transition_mat = zeros((nb_struct, nb_struct), dtype=np.longdouble)
    tables = [np.stack([_pair_table(s.str_struct) for s in step]) if step else None for step in fast_paths]
    for step_i, fold_step in enumerate(fast_paths):
        prev = fast_paths[step_i - 1]
        ptab = tables[step_i - 1]
        for ci, struct in enumerate(fold_step):
            cur = tables[step_i][ci]
            # previous structure connected  <=>  all of its pairs are pairs of `struct`
            connected = np.nonzero(((ptab == -1) | (ptab == cur[None, :])).all(axis=1))[0]
            map_cur, cur_nrj = struct_map[struct.str_struct]
            for si in connected:
                map_prev, prev_nrj = struct_map[prev[si].str_struct]
                delta_nrj = cur_nrj - prev_nrj
                if map_cur != map_prev:
                    transition_mat[map_prev, map_cur] = min(1.0, exp(-delta_nrj / KT))
                    transition_mat[map_cur, map_prev] = min(1.0, exp(delta_nrj / KT))
    for si in range(nb_struct):
        transition_mat[si, si] = -transition_mat[si, :].sum()
    return transition_mat


def unique_structures(fast_paths):
    """Structures of the graph in order of first appearance (rafft_kin.py:106-112)."""
    index, ordered = {}, []
    for step in fast_paths:
        for st in step:
            if st.str_struct not in index:
                index[st.str_struct] = len(ordered)
                ordered.append(st)
    return ordered, index


def kinetics(fast_paths, max_time, n_steps, initial_pop=None):
    """Master-equation populations on the fast-folding graph; same contract as the reference's
    `kinetics` (rafft_kin.py:94-150): returns (trajectory, times, struct_list, str_equi_pop) with
    trajectory[0] the initial population and n_steps rows at times exp(k * max_time / n_steps - 4)."""
    from scipy.linalg import eig, inv      # imported here: SciPy costs 0.3 s at start-up, the fold CLI never needs it
    struct_list, index = unique_structures(fast_paths)
    struct_map = {st.str_struct: (index[st.str_struct], st.energy) for st in struct_list}
    nb_struct = len(struct_list)
    rate = get_transition_mat(fast_paths, nb_struct, struct_map)

    p0 = zeros(nb_struct, dtype=np.longdouble)
    if initial_pop is None:
        p0[0] = 1.0                                  # everything starts unfolded
    else:
        for where, weight in initial_pop:
            p0[where] = weight

    # p(t) = W exp(diag(V) t) W^-1 p0   with (V, W) the eigen-decomposition of the transposed rate matrix
    V, W = eig(rate.T, check_finite=True)
    coef = inv(W) @ p0
    ks = np.arange(n_steps)
    sample_times = np.exp(ks * (max_time / n_steps) - 4)
    times = [exp(-4)] + [t for t in sample_times]
    trajectory = [deepcopy(p0)]
    for t in sample_times:
        pop = (W @ (np.exp(V * t) * coef)).real
        trajectory.append(pop / pop.sum())
    final = trajectory[-1]
    str_equi_pop = [(st.str_struct, st.energy, final[k].real, k) for k, st in enumerate(struct_list)]
    return trajectory, times, struct_list, str_equi_pop


def graph_arrays(fast_paths):
    """The graph as flat arrays for the C-ABI: step sizes, all dot-bracket rows back to back, row -> unique index,
    energy of every unique structure (of its first appearance, rafft_kin.py:115), the unique structures."""
    struct_list, index = unique_structures(fast_paths)
    sizes = np.array([len(step) for step in fast_paths], dtype=np.int32)
    rows = "".join(st.str_struct for step in fast_paths for st in step).encode("ascii")
    uid = np.array([index[st.str_struct] for step in fast_paths for st in step], dtype=np.int32)
    energy = np.array([float(st.energy) for st in struct_list], dtype=np.float64)
    return sizes, rows, uid, energy, struct_list


def rate_matrix_gpu(fast_paths, kt=KT):
    """get_transition_mat (rafft_kin.py:68-91) on the GPU -> (torch float64 CUDA tensor S x S, struct_list)"""
    import ctypes as C
    import torch
    from . import _native as N
    sizes, rows, uid, energy, struct_list = graph_arrays(fast_paths)
    S, L = len(struct_list), len(struct_list[0].str_struct)
    rate = torch.empty((S, S), dtype=torch.float64, device="cuda")
    N.check(N.lib().rafft_kin_rate_matrix(len(sizes), sizes.ctypes.data_as(C.POINTER(C.c_int)), L, rows,
                                          uid.ctypes.data_as(C.POINTER(C.c_int)), S,
                                          energy.ctypes.data_as(C.POINTER(C.c_double)), float(kt), C.c_void_p(rate.data_ptr())))
    return rate, struct_list, energy


SPECTRAL_MAX_SPAN_KT = 30.0     # sqrt(pi_max / pi_min) = e^15 = 3e6: the spectral formula keeps ~9 digits


def solve_master_equation(rate, energy, p0, sample_times, method="auto", substeps=32):
    """p(t) at `sample_times` for dp/dt = rate^T p (torch float64 tensors on any device; on the GPU the dense
    factorisations are rocSOLVER's).  Returns an (n_times, S) numpy array of populations normalised to 1."""
    import torch
    dev = rate.device
    S = rate.shape[0]
    A = rate.T.contiguous()                           # M_ij = k(i -> j), rafft_kin.py:83-85
    energy = np.asarray(energy, dtype=np.float64)
    span = float(energy.max() - energy.min()) / KT
    if method == "auto":
        method = "spectral" if span <= SPECTRAL_MAX_SPAN_KT else "implicit"
    if method == "spectral":
        e = torch.as_tensor(energy, device=dev)
        d = torch.exp(-0.5 * (e - e.min()) / KT)     # sqrt(pi), <= 1
        B = A * (d[None, :] / d[:, None])            # B[j,i] = A[j,i] * sqrt(pi_i / pi_j): symmetric by detailed balance
        B = 0.5 * (B + B.T)
        lam, Q = torch.linalg.eigh(B)                # rocSOLVER syevd on the GPU
        coef = Q.T @ (p0 / d)
        tt = torch.as_tensor(np.asarray(sample_times, dtype=np.float64), device=dev)
        P = d[:, None] * (Q @ (torch.exp(lam[:, None] * tt[None, :]) * coef[:, None]))      # all sample times at once
        P = P / P.sum(dim=0, keepdim=True)
        return P.T.cpu().numpy()
    if method not in ("implicit", "implicit-dense"):
        raise ValueError(f"unknown method {method!r}")
    # TR-BDF2 with gamma = 2 - sqrt(2): both stages solve with (I - c h A), c = 1 - 1/sqrt(2)
    g = 2.0 - 2.0 ** 0.5
    c = 1.0 - 0.5 * 2.0 ** 0.5
    # The generator has a handful of non-zeros per row (a structure is connected to the structures of the neighbouring
    # folding steps whose pair set it contains or is contained in): the factorisations are SPARSE LU (SuperLU; the 6001 x 6001
    # matrix of BASELINE configs[4] holds 0.05 % non-zeros - 40 dense getrf of it cost 11.8 s on the device, the sparse ones
    # 0.2 s).  The non-zeros are pulled out of the device matrix on the device; the dense path stays for dense graphs.
    nnz = int(torch.count_nonzero(A).item())
    sparse_ok = method == "implicit" and nnz <= 0.05 * S * S
    if sparse_ok:
        try:
            import scipy.sparse as sp
            from scipy.sparse.linalg import splu
        except ImportError:               # no SciPy: the dense rocSOLVER path below does the same integration
            sparse_ok = False
    if sparse_ok:
        idx = torch.nonzero(A)
        vals = A[idx[:, 0], idx[:, 1]].cpu().numpy()
        idx = idx.cpu().numpy()
        As = sp.csc_matrix((vals, (idx[:, 0], idx[:, 1])), shape=(S, S))
        eye_s = sp.identity(S, dtype=np.float64, format="csc")
        y = p0.cpu().numpy().astype(np.float64)
        t_now = 0.0
        out = []
        for t in sample_times:
            m = max(1, int(np.ceil(substeps * (np.log(float(t) / t_now) / 0.3 if t_now > 0 else 1.0))))
            h = (float(t) - t_now) / m
            # (minimum degree on the pattern of A + A^T: the graph is close to a forest of folding paths - 76 k non-zeros in
            #  L + U on the configs[4] graph against 1.0 M with the default COLAMD and 15 M unordered)
            lu = splu((eye_s - (c * h) * As).tocsc(), permc_spec="MMD_AT_PLUS_A")
            for _ in range(m):
                yg = lu.solve(y + (0.5 * g * h) * (As @ y))
                y = lu.solve(yg / (g * (2.0 - g)) - ((1.0 - g) ** 2 / (g * (2.0 - g))) * y)
            t_now = float(t)
            out.append(y / y.sum())
        return np.stack(out)
    eye = torch.eye(S, dtype=torch.float64, device=dev)
    y = p0.clone()
    t_now = 0.0
    out = []
    for t in sample_times:
        # `substeps` per factor e^0.3 of time (the reference's default spacing): the relative step stays constant
        m = max(1, int(np.ceil(substeps * (np.log(float(t) / t_now) / 0.3 if t_now > 0 else 1.0))))
        h = (float(t) - t_now) / m
        LU, piv = torch.linalg.lu_factor(eye - (c * h) * A)                  # rocSOLVER getrf, one per output interval
        for _ in range(m):
            rhs = y + (0.5 * g * h) * (A @ y)
            yg = torch.linalg.lu_solve(LU, piv, rhs[:, None])[:, 0]
            rhs = yg / (g * (2.0 - g)) - ((1.0 - g) ** 2 / (g * (2.0 - g))) * y
            y = torch.linalg.lu_solve(LU, piv, rhs[:, None])[:, 0]
        t_now = float(t)
        out.append((y / y.sum()).cpu().numpy().copy())
    return np.stack(out)


def kinetics_gpu(fast_paths, max_time, n_steps, initial_pop=None, method="auto", substeps=32):
    """Same contract as `kinetics` (rafft_kin.py:94-150).  The rate matrix (inclusion search + Metropolis rates) is computed on
    the MI355X; the master equation is solved on the device (spectral, implicit-dense) or - the sparse TR-BDF2 integrator that
    `implicit` and, beyond 30 KT of energy span, `auto` use - on the host from the non-zeros (see solve_master_equation).  Returns
    (trajectory, times, struct_list, str_equi_pop); trajectory rows are float64 numpy arrays."""
    import torch
    rate, struct_list, energy = rate_matrix_gpu(fast_paths)
    S = len(struct_list)
    p0 = torch.zeros(S, dtype=torch.float64, device=rate.device)
    if initial_pop is None:
        p0[0] = 1.0
    else:
        for where, weight in initial_pop:
            p0[where] = weight
    sample_times = np.exp(np.arange(n_steps) * (max_time / n_steps) - 4)
    times = [exp(-4)] + [t for t in sample_times]
    pops = solve_master_equation(rate, energy, p0, sample_times, method, substeps)
    trajectory = [p0.cpu().numpy().copy()] + [row for row in pops]
    final = trajectory[-1]
    str_equi_pop = [(st.str_struct, st.energy, float(final[k]), k) for k, st in enumerate(struct_list)]
    return trajectory, times, struct_list, str_equi_pop


def main(argv=None):
    """bin/rafft_kin (bin/rafft_kin:15-55) without the matplotlib plot."""
    import argparse
    from .utils import parse_rafft_output, read_sidecar
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('rafft_out', help="rafft_output (the --traj text, or the binary side-car with --sidecar)")
    parser.add_argument('--sidecar', action="store_true", help="rafft_out is the binary side-car written by `rafft --traj --sidecar`")
    parser.add_argument('--exact', action="store_true", help="with --sidecar: use the exact energies instead of the one decimal\n"
                                                             "of the text format (changes the populations: rates are exponential in the energy)")
    parser.add_argument('--n_steps', '-ns', help="integration steps", type=int, default=100)
    parser.add_argument('--init_pop', '-ip', help="initialization of the population <POS>:<WEI>", nargs="*")
    parser.add_argument('--max_time', '-mt', help="max time (exp scale)", type=float, default=30)
    parser.add_argument('--gpu', action="store_true", help="rate matrix on the MI355X (kinetics_gpu); the solve on the device or, for the sparse\n"
                                                           "integrator, on the host - see --method")
    parser.add_argument('--method', choices=["auto", "spectral", "implicit", "implicit-dense"], default="auto",
                        help="with --gpu: spectral = the reference's formula (device, rocSOLVER syevd; valid up to ~30 KT of energy span),\n"
                             "implicit = TR-BDF2 with sparse LU of the generator's non-zeros (host, SciPy SuperLU; dense device path\n"
                             "without SciPy), implicit-dense = the same with dense getrf on the device, auto = whichever is valid")
    args = parser.parse_args(argv)
    init_population = None
    if args.init_pop is not None:     # the reference crashes here (None += ...); we accept the documented syntax
        init_population = [(int(el.split(":")[0]), float(el.split(":")[1])) for el in args.init_pop]
    if args.sidecar:
        fast_paths, seq = read_sidecar(args.rafft_out, text_energies=not args.exact)
    else:
        fast_paths, seq = parse_rafft_output(args.rafft_out)
    if args.gpu:
        trajectory, times, struct_list, equi_pop = kinetics_gpu(fast_paths, args.max_time, args.n_steps, init_population, args.method)
    else:
        trajectory, times, struct_list, equi_pop = kinetics(fast_paths, args.max_time, args.n_steps, init_population)
    equi_pop.sort(key=lambda el: el[2])
    for st, nrj, fp, si in equi_pop:
        print("{} {:6.3f} {:5.1f} {:d}".format(st, fp, nrj, si))


if __name__ == "__main__":
    main()
