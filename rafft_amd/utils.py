"""Host-side mirror of the reference's rafft/utils.py data types and I/O helpers
(the numeric helpers of that file - prep_sequence, seq_conv, auto_cor,
eval_one_struct, get_inner_loop/get_outer_loop - live in the HIP kernels)."""
import numpy as np


class Structure:
    """Beam entry as consumers of the reference see it (rafft/utils.py:32-39):
    `.str_struct`, `.energy` (kcal/mol, the float32 value ViennaRNA returns) and
    `.pair_list`.  `.dcal` is the exact integer energy."""
    __slots__ = ("str_struct", "energy", "dcal", "node_list")

    def __init__(self, str_struct="", dcal=0, energy=None):
        self.str_struct = str_struct
        self.dcal = int(dcal)
        # ViennaRNA: `(float)en / 100.` returned through a float
        self.energy = float(np.float32(np.float64(np.float32(dcal)) / 100.0)) if energy is None else energy
        self.node_list = ()

    @property
    def pair_list(self):
        return paired_positions(self.str_struct)

    def __repr__(self):
        return f"{self.str_struct} {self.energy:6.1f}"


def energies_from_dcal(dcal):
    """Vectorised `(float)en / 100.` (float32 result widened to Python floats)."""
    d = np.asarray(dcal)
    return (d.astype(np.float32).astype(np.float64) / 100.0).astype(np.float32).astype(np.float64)


def dot_bracket(pair_list, len_seq, SEQ=None):
    """rafft/utils.py:42-50"""
    s = ["."] * len_seq
    for pi, pj in pair_list:
        s[pi], s[pj] = "(", ")"
    return "".join(s)


def paired_positions(structure):
    """rafft/utils.py:53-67"""
    pile_reg, pile_pk, pairs = [], [], []
    for i, c in enumerate(structure):
        if c in "<(":
            pile_reg.append(i)
        elif c == "[":
            pile_pk.append(i)
        elif c in ">)":
            pairs.append((pile_reg.pop(), i))
        elif c == "]":
            pairs.append((pile_pk.pop(), i))
    return pairs


def read_fasta(infile):
    """rafft/utils.py:161-169"""
    results = {}
    name = None
    for line in open(infile):
        if line.startswith(">"):
            name = line.strip()[1:]
            results[name] = ""
        else:
            results[name] += line.strip()
    return results


def parse_rafft_output(infile):
    """Reader of the fast-folding-graph text (rafft/utils.py:172-185)."""
    results = []
    with open(infile) as fh:
        seq = fh.readline().strip()
        for line in fh:
            if line.startswith("# --"):
                results.append([])
            else:
                str_struct, nrj = line.strip().split()
                st = Structure(str_struct, 0)
                st.energy = float(nrj)
                st.dcal = int(round(float(nrj) * 100))
                results[-1].append(st)
    return results, seq


def format_trajectory(sequence, trajectory):
    """The `--traj` text of bin/rafft:73-79."""
    out = [sequence]
    for si, step in enumerate(trajectory):
        out.append("# {:-^20}".format(si))
        out.extend(f"{s.str_struct} {s.energy:6.1f}" for s in step)
    return "\n".join(out) + "\n"
