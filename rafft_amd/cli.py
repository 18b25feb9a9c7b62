"""`rafft` command line - same flags, defaults and output formats as the reference's
bin/rafft (bin/rafft:7-31,34-80), running the fold on the MI355X.

Differences, all additive: `-sf` may hold several FASTA records (they are folded as one
GPU batch and printed one after the other); `--nono` (the reference's to-be-removed
test implementation, bin/rafft:29) is not provided."""
import argparse
import sys


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('--sequence', '-s', help="sequence")
    parser.add_argument('--seq_file', '-sf', help="sequence file")
    parser.add_argument('--n_mode', '-n', help="Number of positional lags to search for stems", type=int, default=100)
    parser.add_argument('--max_stack', '-ms', help="number of stored structures (default=1)", type=int, default=1)
    parser.add_argument('--min_nrj', '-mn', help="minimum loop energy to be formed", type=float, default=0)
    parser.add_argument('--min_bp', '-mb', help="minimum bp number to be detectable (parsed, unused - as in the reference)", type=int, default=1)
    parser.add_argument('--min_hp', '-mh', help="minimum unpaired positions in hairpins", type=int, default=3)
    parser.add_argument('--pad', '-p', help="padding (parsed, unused - as in the reference)", type=float, default=1.0)
    parser.add_argument('--max_branch', help="maximum branches to explor", type=int, default=1000)
    parser.add_argument('--bp_only', action="store_true", help="(parsed, unused - as in the reference)")
    parser.add_argument('--bench', action="store_true", help="output for benchmarks")
    parser.add_argument('-tr', '--traj', action="store_true", help="output full trajectories")
    parser.add_argument('--temp', type=float, help="temperature (only 37.0)", default=37.0)
    parser.add_argument('-gc', '--gc_wei', type=float, help="GC weight", default=3.00)
    parser.add_argument('-au', '--au_wei', type=float, help="AU weight", default=2.00)
    parser.add_argument('-gu', '--gu_wei', type=float, help="GU weight", default=1.00)
    parser.add_argument('--batch', action="store_true",
                        help="treat every FASTA record / line of -sf as its own sequence (one GPU batch)")
    return parser.parse_args(argv)


def read_sequences(args):
    assert args.sequence is not None or args.seq_file is not None, "error, the sequence is missing!"
    if args.sequence is not None:
        return [args.sequence]
    lines = [l.strip() for l in open(args.seq_file)]
    if not args.batch:   # reference behaviour: all non-header lines joined (bin/rafft:42)
        return ["".join(l for l in lines if not l.startswith(">")).replace("T", "U")]
    seqs, cur = [], []
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


def main(argv=None, fold_batch=None):
    args = parse_arguments(argv)
    seqs = read_sequences(args)
    if fold_batch is None:
        from .rafft import fold_batch
    results = fold_batch(seqs, args.n_mode, args.max_stack, args.max_branch, args.min_hp, args.min_nrj, args.traj,
                         args.temp, args.gc_wei, args.au_wei, args.gu_wei)
    for s, r in zip(seqs, results):
        print(format_result(s, r, args))


if __name__ == '__main__':
    main()
