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


# ---- streaming writers: from the fold's flat result buffers to the reference's text formats, no
# Structure objects, no per-row Python - what `rafft --traj` needs at max_stack = 1000 and what the batch front-end
# needs for thousands of sequences.  `raw` is BatchResult.raw(i): (L, step sizes, rows uint8 (n, L), dcal int32 (n,)).

def _rows_with_energy(rows, dcal, prefix=b"", with_pairs=False):
    """bytes of the lines `<prefix><dot-bracket> <E:6.1f>[ <#pairs>]\n` for all rows at once (bin/rafft:64-68,77-78)"""
    n, L = rows.shape
    if n == 0:
        return b""
    en = np.char.mod("%6.1f", energies_from_dcal(dcal)).astype("S")
    cols = [en]
    if with_pairs:
        cols.append(np.char.add(b" ", np.char.mod("%d", (rows == ord("(")).sum(axis=1)).astype("S")))
    if any((np.char.str_len(c) != c.dtype.itemsize).any() for c in cols):
        # columns of ragged width (an energy below -999.9, pair counts of different digit counts): join row by row
        tail = cols[0] if len(cols) == 1 else np.char.add(cols[0], cols[1])
        return b"".join(prefix + rows[k].tobytes() + b" " + tail[k] + b"\n" for k in range(n))
    parts = []
    if prefix:
        parts.append(np.broadcast_to(np.frombuffer(prefix, dtype=np.uint8), (n, len(prefix))))
    parts += [rows, np.full((n, 1), 32, np.uint8)]
    parts += [np.frombuffer(c.tobytes(), dtype=np.uint8).reshape(n, c.dtype.itemsize) for c in cols]
    parts.append(np.full((n, 1), 10, np.uint8))
    return np.concatenate(parts, axis=1).tobytes()


def write_result_text(fh, sequence, raw, traj=False, bench=False):
    """One sequence in the reference's output formats (bin/rafft:59-79) straight from the flat result buffers:
    final (sequence line + rows), --bench (seq len db E #pairs per row, no header), --traj (sequence line, then per
    step `# ---------k----------` + rows).  `fh` is a binary file object."""
    L, sizes, rows, dcal = raw
    seqb = sequence.encode("ascii")
    if traj:
        fh.write(seqb + b"\n")
        o = 0
        for si, cnt in enumerate(sizes):
            fh.write("# {:-^20}\n".format(si).encode("ascii"))
            fh.write(_rows_with_energy(rows[o:o + cnt], dcal[o:o + cnt]))
            o += cnt
        return
    last = sizes[-1] if sizes else 0
    r, d = rows[len(rows) - last:], dcal[len(dcal) - last:]
    if bench:
        fh.write(_rows_with_energy(r, d, prefix=seqb + b" " + str(len(sequence)).encode() + b" ", with_pairs=True))
    else:
        fh.write(seqb + b"\n")
        fh.write(_rows_with_energy(r, d))


def write_sidecar_raw(path, sequence, raw):
    """write_sidecar from the flat result buffers of a --traj fold"""
    L, sizes, rows, dcal = raw
    with open(path, "wb") as fh:
        fh.write(_SIDECAR_MAGIC)
        fh.write(np.array([1, L, len(sizes), len(dcal)], dtype="<u4").tobytes())
        fh.write(sequence.encode("ascii"))
        fh.write(np.asarray(sizes, dtype="<i4").tobytes())
        fh.write(np.asarray(dcal, dtype="<i4").tobytes())
        fh.write(np.ascontiguousarray(rows).tobytes())


# ---- binary side-car of the fast-folding graph ---------------------
# The `--traj` text (bin/rafft:73-79) prints energies with one decimal, so a reader of the text loses the
# exact dcal values and has to re-parse every dot-bracket string.  The side-car keeps what the fold engine
# returns: the dot-bracket rows and their integer energies, step by step.
#   magic "RAFFTFFG" | u32 version=1 | u32 L | u32 n_steps | u32 n_structs
#   | sequence (L bytes) | step sizes (n_steps x i32) | dcal (n_structs x i32) | rows (n_structs x L bytes)
_SIDECAR_MAGIC = b"RAFFTFFG"


def write_sidecar(path, sequence, trajectory):
    import numpy as np
    L = len(sequence)
    sizes = np.array([len(step) for step in trajectory], dtype="<i4")
    structs = [s for step in trajectory for s in step]
    dcal = np.array([s.dcal for s in structs], dtype="<i4")
    with open(path, "wb") as fh:
        fh.write(_SIDECAR_MAGIC)
        fh.write(np.array([1, L, len(sizes), len(structs)], dtype="<u4").tobytes())
        fh.write(sequence.encode("ascii"))
        fh.write(sizes.tobytes())
        fh.write(dcal.tobytes())
        for s in structs:
            row = s.str_struct.encode("ascii")
            assert len(row) == L
            fh.write(row)


def read_sidecar(path, text_energies=False):
    """-> (fast_paths, sequence), the same shape parse_rafft_output returns, with exact dcal energies.
    text_energies=True rounds `.energy` to the one decimal the `--traj` text carries (bin/rafft:77-78), which is
    all the reference's rafft_kin ever sees; `.dcal` stays exact."""
    import numpy as np
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:8] != _SIDECAR_MAGIC:
        raise ValueError(f"{path}: not a RAFFT fast-folding-graph side-car")
    version, L, n_steps, n_structs = (int(x) for x in np.frombuffer(buf, dtype="<u4", count=4, offset=8))
    if version != 1:
        raise ValueError(f"{path}: unsupported side-car version {version}")
    o = 24
    sequence = buf[o:o + L].decode("ascii"); o += L
    sizes = np.frombuffer(buf, dtype="<i4", count=n_steps, offset=o); o += 4 * n_steps
    dcal = np.frombuffer(buf, dtype="<i4", count=n_structs, offset=o); o += 4 * n_structs
    if int(sizes.sum()) != n_structs or len(buf) != o + n_structs * L:
        raise ValueError(f"{path}: truncated or inconsistent side-car")
    fast_paths, k = [], 0
    for n in sizes:
        step = []
        for _ in range(int(n)):
            st = Structure(buf[o + k * L:o + (k + 1) * L].decode("ascii"), int(dcal[k]))
            if text_energies:
                st.energy = float(f"{st.energy:6.1f}")
            step.append(st)
            k += 1
        fast_paths.append(step)
    return fast_paths, sequence
