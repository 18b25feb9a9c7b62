"""`rafft` command line - same flags, defaults and output formats as the reference's
bin/rafft (bin/rafft:7-31,34-80), running the fold on the MI355X.

Differences, all additive: `-sf` may hold several FASTA records (they are folded as one
GPU batch and printed one after the other); `--nono` (the reference's to-be-removed
test implementation, bin/rafft:29) is not provided.

Energies: the reference evaluates with ViennaRNA's loaded parameter set (rafft/utils.py:17-21).  Here the set is read ONCE, up
front: from RAFFT_PARAMS=<ViennaRNA 2.x parameter file>, else through an importable `RNA` module (RNA.params_save),
else the built-in Turner-2004 37 C tables are used.  The built-in tables reproduce all 11 505 energies the reference
publishes, but entries no published energy exercises are rule-derived: out of sample about 1 structure in 20 gets
another energy than ViennaRNA's - supply ViennaRNA's own tables when that matters.  Which set is
active: `python -c "import rafft_amd; print(rafft_amd.params_info())"`."""
import argparse
import sys


# (flags, keyword arguments) - the option surface of the reference's CLI, bin/rafft:11-30.  `--pad`, `--min_bp`
# and `--bp_only` are accepted and ignored exactly as there (they are parsed but never forwarded, bin/rafft:50-52).
_OPTIONS = [
    (("--sequence", "-s"), dict(help="sequence")),
    (("--seq_file", "-sf"), dict(help="sequence file (FASTA or plain)")),
    (("--n_mode", "-n"), dict(type=int, default=100, help="number of positional lags searched for stems")),
    (("--max_stack", "-ms"), dict(type=int, default=1, help="number of structures kept per folding step")),
    (("--min_nrj", "-mn"), dict(type=float, default=0, help="a stem must change the energy by less than this")),
    (("--min_bp", "-mb"), dict(type=int, default=1, help="accepted, unused")),
    (("--min_hp", "-mh"), dict(type=int, default=3, help="minimum unpaired positions in a hairpin")),
    (("--pad", "-p"), dict(type=float, default=1.0, help="accepted, unused")),
    (("--max_branch",), dict(type=int, default=1000, help="maximum number of new structures per folding step")),
    (("--bp_only",), dict(action="store_true", help="accepted, unused")),
    (("--bench",), dict(action="store_true", help="one line per structure: seq len structure energy #pairs")),
    (("-tr", "--traj"), dict(action="store_true", help="print the whole fast-folding graph")),
    (("--temp",), dict(type=float, default=37.0, help="temperature in C (default 37).  Other temperatures need a ViennaRNA parameter file with\nenthalpies (RAFFT_PARAMS=<file>, or an importable RNA module): the built-in tables are 37 C only")),
    (("-gc", "--gc_wei"), dict(type=float, default=3.0, help="GC weight")),
    (("-au", "--au_wei"), dict(type=float, default=2.0, help="AU weight")),
    (("-gu", "--gu_wei"), dict(type=float, default=1.0, help="GU weight")),
    (("--batch",), dict(action="store_true", help="every record of -sf is its own sequence, all folded as one GPU batch: FASTA records,\n"
                                                  "one sequence per line, or a CSV with a header (column --csv_column; the reference's\n"
                                                  "benchmark_cleaned_all_length.csv has `seq`).  With --bench this replaces the process pool of\n"
                                                  "benchmark_results/bench_fft.py")),
    (("--csv_column",), dict(default="seq", help="sequence column of a CSV given to -sf --batch (default: seq)")),
    (("--output", "-o"), dict(help="write the result there instead of stdout")),
    (("--sidecar",), dict(help="with --traj: also write the fast-folding graph as a binary side-car (exact dcal energies,\n"
                               "no strings to re-parse) that `rafft_kin --sidecar` reads; with several sequences the\n"
                               "files are <SIDECAR>.0, <SIDECAR>.1, ...")),
]


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    for flags, kw in _OPTIONS:
        parser.add_argument(*flags, **kw)
    return parser.parse_args(argv)


def read_sequences(args):
    assert args.sequence is not None or args.seq_file is not None, "error, the sequence is missing!"
    if args.sequence is not None:
        return [args.sequence]
    lines = [l.strip() for l in open(args.seq_file)]
    if not args.batch:   # reference behaviour: all non-header lines joined (bin/rafft:42)
        return ["".join(l for l in lines if not l.startswith(">")).replace("T", "U")]
    seqs, cur = [], []
    if lines and "," in lines[0] and not lines[0].startswith(">"):     # CSV with a header line
        import csv
        with open(args.seq_file, newline="") as fh:
            rd = csv.DictReader(fh)
            if args.csv_column not in (rd.fieldnames or []):
                raise SystemExit(f"{args.seq_file}: no column {args.csv_column!r} (columns: {rd.fieldnames})")
            return [r[args.csv_column].strip().replace("T", "U") for r in rd if r[args.csv_column].strip()]
    fasta = any(l.startswith(">") for l in lines)
    for l in lines:
        if fasta:
            if l.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            elif l:
                cur.append(l)
        elif l:
            seqs.append(l)
    if cur:
        seqs.append("".join(cur))
    return [s.replace("T", "U") for s in seqs]


def format_result(sequence, result, args):
    out = []
    if args.traj:
        final_struct, trajectory = result
        out.append(f"{sequence}")
        for si, fold_step in enumerate(trajectory):
            out.append("# {:-^20}".format(si))
            for struct in fold_step:
                out.append(f"{struct.str_struct} {struct.energy:6.1f}")
    else:
        if not args.bench:
            out.append(f"{sequence}")
        for struct in result:
            if args.bench:
                out.append(f"{sequence} {len(sequence)} {struct.str_struct} {struct.energy:6.1f} {struct.str_struct.count('(')}")
            else:
                out.append(f"{struct.str_struct} {struct.energy:6.1f}")
    return "\n".join(out)


def _table_note():
    """one line on stderr when the fold used rule / model values of the built-in tables (never with ViennaRNA's own tables loaded)"""
    try:
        from .rafft import last_stats
        st = last_stats()
    except Exception:
        return
    import os
    if st.get("n_kept_guessed", 0) > 0 and not os.environ.get("RAFFT_QUIET"):
        sys.stderr.write(f"rafft: built-in energy tables - {st['n_dE_guessed']} of {st['n_dE_evals']} stem energies ({st['n_kept_guessed']} kept "
                         "candidates) read an interior-loop entry that no published energy pins; set RAFFT_PARAMS=<ViennaRNA parameter file> "
                         "for ViennaRNA's own values (RAFFT_QUIET=1 silences this)\n")


def main(argv=None, fold_batch=None):
    args = parse_arguments(argv)
    seqs = read_sequences(args)
    if fold_batch is None:
        from .rafft import fold_batch
    results = fold_batch(seqs, args.n_mode, args.max_stack, args.max_branch, args.min_hp, args.min_nrj, args.traj,
                         args.temp, args.gc_wei, args.au_wei, args.gu_wei)
    out = open(args.output, "wb") if args.output else sys.stdout.buffer if hasattr(sys.stdout, "buffer") else None
    try:
        for k, s in enumerate(seqs):
            side = (args.sidecar if len(seqs) == 1 else f"{args.sidecar}.{k}") if (args.sidecar and args.traj) else None
            raw = results.raw(k) if hasattr(results, "raw") and out is not None else None
            if raw is not None:       # streaming: flat result buffers -> text, no Structure objects
                from .utils import write_result_text, write_sidecar_raw
                write_result_text(out, s, raw, traj=args.traj, bench=args.bench)
                if side:
                    write_sidecar_raw(side, s, raw)
            else:                     # an injected fold function (tests) or a captured text stdout
                txt = format_result(s, results[k], args) + "\n"
                if out is not None:
                    out.write(txt.encode("ascii"))
                else:
                    sys.stdout.write(txt)
                if side:
                    from .utils import write_sidecar
                    write_sidecar(side, s, results[k][1])
        if out is not None:
            out.flush()
        _table_note()
    finally:
        if args.output:
            out.close()


if __name__ == '__main__':
    main()
