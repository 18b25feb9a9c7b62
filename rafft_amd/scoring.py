"""Prediction accuracy against a known structure: PPV and sensitivity as the reference obtains them
from RNAstructure's `scorer` (benchmark_results/scoring.py:76-94; SURVEY.md 8f-4, a "next" row).
`scorer`'s default rule: a pair (i,j) counts as found when the other structure holds (i,j), (i+-1,j)
or (i,j+-1).  Reproduces the pvv/sens columns of the reference's *_scores.csv (checked in tests)."""
from .utils import paired_positions


def _found(pair, others):
    i, j = pair
    return (i, j) in others or (i - 1, j) in others or (i + 1, j) in others or (i, j - 1) in others or (i, j + 1) in others


def score(predicted, known):
    """(ppv, sensitivity) in percent; the reference maps an undefined value (no pairs) to 0."""
    P, K = set(paired_positions(predicted)), set(paired_positions(known))
    ppv = 100.0 * sum(1 for p in P if _found(p, K)) / len(P) if P else 0.0
    sens = 100.0 * sum(1 for k in K if _found(k, P)) / len(K) if K else 0.0
    return ppv, sens


def best_of(structures, known):
    """The reference's selection in test_one_seq (scoring.py:83-94): last structure reaching the
    highest PPV (`>=`)."""
    best = (0.0, 0.0, None)
    for st in structures:
        db = st if isinstance(st, str) else st.str_struct
        p, s = score(db, known)
        if p >= best[0]:
            best = (p, s, db)
    return best
