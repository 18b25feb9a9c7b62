"""Kinetics on the fast-folding graph: host-side mirror of the reference's rafft/rafft_kin.py
(`kinetics`, `get_transition_mat`; SURVEY.md 8f-2, a "next" row: post-processing of the fold's
output, not part of the GPU hot path).  Same inputs, outputs, Metropolis rates, quirks and the same
SciPy eig/inv solve; only the O(steps * ms^2 * L) Python set-inclusion search of
`get_connected_prev` (rafft_kin.py:48-56) is vectorised over pair tables - an exact equivalent."""
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
    args = parser.parse_args(argv)
    init_population = None
    if args.init_pop is not None:     # the reference crashes here (None += ...); we accept the documented syntax
        init_population = [(int(el.split(":")[0]), float(el.split(":")[1])) for el in args.init_pop]
    if args.sidecar:
        fast_paths, seq = read_sidecar(args.rafft_out, text_energies=not args.exact)
    else:
        fast_paths, seq = parse_rafft_output(args.rafft_out)
    trajectory, times, struct_list, equi_pop = kinetics(fast_paths, args.max_time, args.n_steps, init_population)
    equi_pop.sort(key=lambda el: el[2])
    for st, nrj, fp, si in equi_pop:
        print("{} {:6.3f} {:5.1f} {:d}".format(st, fp, nrj, si))


if __name__ == "__main__":
    main()
