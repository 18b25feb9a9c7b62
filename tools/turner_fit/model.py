"""Feature-level restatement of ViennaRNA's `eval_structure` (dangles=2,
special hairpins on, 37 C) used only for fitting/validating the parameter
tables against the reference's known-answer triples.  energy = const + sum
count[key] * theta[key].  The production evaluators (oracle/ C and the HIP
device code) are separate implementations checked against the same triples.
"""
import math
from collections import Counter
from . import prior as P

BASE = {"N": 0, "A": 1, "C": 2, "G": 3, "U": 4}
PAIR = [[0] * 5 for _ in range(5)]
PAIR[2][3] = 1  # CG
PAIR[3][2] = 2  # GC
PAIR[3][4] = 3  # GU
PAIR[4][3] = 4  # UG
PAIR[1][4] = 5  # AU
PAIR[4][1] = 6  # UA
RTYPE = [0, 2, 1, 4, 3, 6, 5, 7]


def ptype(a, b):
    t = PAIR[a][b]
    return t if t else 7


def pair_table(struct):
    pt = [0] * (len(struct) + 2)
    st = []
    for i, c in enumerate(struct, 1):
        if c == "(":
            st.append(i)
        elif c == ")":
            j = st.pop()
            pt[i] = j
            pt[j] = i
    assert not st
    return pt


def canon_stack(t1, t2):
    return ("stack",) + min((t1, t2), (t2, t1))


def canon_int11(t1, t2, a, b):
    return ("int11",) + min((t1, t2, a, b), (t2, t1, b, a))


def canon_int22(t1, t2, a, b, c, d):
    return ("int22",) + min((t1, t2, a, b, c, d), (t2, t1, c, d, a, b))


def prior_value(key):
    k = key[0]
    if k == "stack":
        return P.STACK[key[1] - 1][key[2] - 1]
    if k == "hp":
        return P.HAIRPIN[key[1]]
    if k == "bulge":
        return P.BULGE[key[1]]
    if k == "int":
        return P.INTERIOR[key[1]]
    if k == "mmH":
        return P.MM_HAIRPIN[key[1]][key[2]][key[3]]
    if k == "mmI":
        return P.MM_INTERIOR[key[1]][key[2]][key[3]]
    if k == "mm1n":
        return P.MM_INTERIOR_1N[key[1]][key[2]][key[3]]
    if k == "mm23":
        return P.MM_INTERIOR_23[key[1]][key[2]][key[3]]
    if k == "mmM":
        return P.MM_MULTI[key[1]][key[2]][key[3]]
    if k == "mmE":
        return P.MM_EXT[key[1]][key[2]][key[3]]
    if k == "d5":
        return P.DANGLE5[key[1]][key[2]]
    if k == "d3":
        return P.DANGLE3[key[1]][key[2]]
    if k == "termAU":
        return P.TERM_AU
    if k == "MLclosing":
        return P.ML_CLOSING
    if k == "MLintern":
        return P.ML_INTERN
    if k == "MLbase":
        return P.ML_BASE
    if k == "tri":
        return P.TRILOOPS[key[1]]
    if k == "tetra":
        return P.TETRALOOPS[key[1]]
    if k == "hexa":
        return P.HEXALOOPS[key[1]]
    if k == "int11":
        return P.int11_prior(*key[1:])
    if k == "int21":
        return P.int21_prior(*key[1:])
    if k == "int22":
        return P.int22_prior(*key[1:])
    raise KeyError(key)


class Feat:
    def __init__(self):
        self.c = Counter()
        self.const = 0

    def add(self, key, n=1):
        self.c[key] += n


def _type7_guard(t):
    # NS pairs index table row 7; priors only cover 1..6 for mismatch tables
    return t


def hairpin(f, size, t, si1, sj1, s6):
    if size <= 30:
        f.add(("hp", size))
    else:
        f.add(("hp", 30))
        f.const += int(P.LXC * math.log(size / 30.0))
    if size < 3:
        return
    if size == 4 and s6[:6] in P.TETRALOOPS:
        f.c.subtract({("hp", 4): 1})
        f.add(("tetra", s6[:6]))
        return
    if size == 6 and s6[:8] in P.HEXALOOPS:
        f.c.subtract({("hp", 6): 1})
        f.add(("hexa", s6[:8]))
        return
    if size == 3:
        if s6[:5] in P.TRILOOPS:
            f.c.subtract({("hp", 3): 1})
            f.add(("tri", s6[:5]))
            return
        if t > 2:
            f.add(("termAU",))
        return
    f.add(("mmH", t, si1, sj1))


def intloop(f, n1, n2, t, t2, si1, sj1, sp1, sq1):
    nl, ns = (n1, n2) if n1 > n2 else (n2, n1)
    if nl == 0:
        f.add(canon_stack(t, t2))
        return
    if ns == 0:
        if nl <= 30:
            f.add(("bulge", nl))
        else:
            f.add(("bulge", 30))
            f.const += int(P.LXC * math.log(nl / 30.0))
        if nl == 1:
            f.add(canon_stack(t, t2))
        else:
            if t > 2:
                f.add(("termAU",))
            if t2 > 2:
                f.add(("termAU",))
        return
    if ns == 1:
        if nl == 1:
            f.add(canon_int11(t, t2, si1, sj1))
            return
        if nl == 2:
            if n1 == 1:
                f.add(("int21", t, t2, si1, sq1, sj1))
            else:
                f.add(("int21", t2, t, sq1, si1, sp1))
            return
        u = nl + 1
        if u <= 30:
            f.add(("int", u))
        else:
            f.add(("int", 30))
            f.const += int(P.LXC * math.log(u / 30.0))
        f.const += min(P.MAX_NINIO, (nl - ns) * P.NINIO)
        f.add(("mm1n", t, si1, sj1))
        f.add(("mm1n", t2, sq1, sp1))
        return
    if ns == 2:
        if nl == 2:
            f.add(canon_int22(t, t2, si1, sp1, sq1, sj1))
            return
        if nl == 3:
            f.add(("int", 5))
            f.const += P.NINIO
            f.add(("mm23", t, si1, sj1))
            f.add(("mm23", t2, sq1, sp1))
            return
    u = nl + ns
    if u <= 30:
        f.add(("int", u))
    else:
        f.add(("int", 30))
        f.const += int(P.LXC * math.log(u / 30.0))
    f.const += min(P.MAX_NINIO, (nl - ns) * P.NINIO)
    f.add(("mmI", t, si1, sj1))
    f.add(("mmI", t2, sq1, sp1))


def ml_stem(f, t, si1, sj1, ext):
    if si1 >= 0 and sj1 >= 0:
        f.add(("mmE" if ext else "mmM", t, si1, sj1))
    elif si1 >= 0:
        f.add(("d5", t, si1))
    elif sj1 >= 0:
        f.add(("d3", t, sj1))
    if t > 2:
        f.add(("termAU",))
    if not ext:
        f.add(("MLintern",))


def features(seq, struct):
    n = len(seq)
    S = [0] + [BASE[c] for c in seq] + [0]
    pt = pair_table(struct)
    f = Feat()
    # exterior loop
    i = 1
    while i <= n:
        if pt[i] == 0:
            i += 1
            continue
        j = pt[i]
        t = ptype(S[i], S[j])
        ml_stem(f, t, S[i - 1] if i > 1 else -1, S[j + 1] if j < n else -1, True)
        i = j + 1
    # all pairs
    for i in range(1, n + 1):
        j = pt[i]
        if j <= i:
            continue
        t = ptype(S[i], S[j])
        # find inner branches
        br = []
        p = i + 1
        while p < j:
            if pt[p] == 0:
                p += 1
            else:
                br.append((p, pt[p]))
                p = pt[p] + 1
        if not br:
            hairpin(f, j - i - 1, t, S[i + 1], S[j - 1], seq[i - 1:i + 7])
        elif len(br) == 1:
            p, q = br[0]
            t2 = RTYPE[ptype(S[p], S[q])]
            intloop(f, p - i - 1, j - q - 1, t, t2, S[i + 1], S[j - 1], S[p - 1], S[q + 1])
        else:
            u = j - i - 1
            for p, q in br:
                u -= q - p + 1
                ml_stem(f, ptype(S[p], S[q]), S[p - 1], S[q + 1], False)
            ml_stem(f, ptype(S[j], S[i]), S[j - 1], S[i + 1], False)
            f.add(("MLclosing",))
            if u:
                f.add(("MLbase",), u)
    return f


def energy(seq, struct, theta=None):
    f = features(seq, struct)
    e = f.const
    for k, c in f.c.items():
        if c:
            e += c * (theta[k] if theta is not None and k in theta else prior_value(k))
    return e
