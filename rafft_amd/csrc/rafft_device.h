// rafft_device.h - device-side Turner-2004 loop energies for gfx950.
//
// Replaces the ViennaRNA call inside the reference's inner loop
// (rafft/utils.py:135-138 eval_one_struct -> fold_compound.eval_structure):
// instead of re-evaluating the whole structure in O(L) for every candidate stem
// (rafft/rafft.py:98) the fold kernels evaluate only the loops a stem changes
// (nearest-neighbour energies are a sum over loops), in integer dcal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RAFFT_MAX_LEN 32768

// Small, hot tables (3.4 KB): copied into LDS by the expand kernel of the smallest size class.
struct SmallT {
    int16_t stack[7][7];
    int16_t mmH[7][5][5], mmI[7][5][5], mm1n[7][5][5], mm23[7][5][5], mmM[7][5][5], mmE[7][5][5];
    int16_t d5[7][5], d3[7][5];
    int32_t hairpin[31], bulge[31], interior[31];
    int32_t ml_base, ml_closing, ml_intern, ninio, max_ninio, term_au;
    // 1 with the built-in tables: bit 0 of an entry of the interior-loop tables (int11, int21, int22, mmI, mm1n, mm23, bulge,
    // interior) then says "rule / model value, no reference-held energy row exercises it" (the values themselves are multiples
    // of 10); 0 with a loaded parameter file, whose every entry is ViennaRNA's (rafft_params.h: scaled_tables)
    int32_t lsb, pad_lsb_[3];
    // special hairpins (tri-, tetra-, hexaloops) in one open-addressing hash table: key = 3 bits per base of
    // the loop with its closing pair, tagged with the loop size in bits 28..; 0 = empty slot
    uint32_t sp_key[128]; int32_t sp_e[128];
};
static_assert(sizeof(SmallT) % 16 == 0, "SmallT is copied to LDS in 16-byte pieces");
__host__ __device__ inline uint32_t sp_tag(int size) { return (uint32_t)(size == 3 ? 1 : size == 4 ? 2 : 3) << 28; }
__host__ __device__ inline uint32_t sp_slot(uint32_t tagged) { return (tagged * 2654435761u) >> 25; }
// Big, rarely hit tables stay in HBM/L2.
struct BigT {
    int16_t int11[7][7][5][5];
    int16_t int21[7][7][5][5][5];
    int16_t int22[7][7][5][5][5][5];
    int32_t logext[RAFFT_MAX_LEN + 2];   // (int)(lxc*log(size/30.)), host libm, size > 30
};
struct EnergyTables {
    SmallT s;
    BigT b;
};

// Pair type of two base codes (ViennaRNA numbering: CG=1 GC=2 GU=3 UG=4 AU=5 UA=6, 0 = no pair) and the
// type of the reversed pair, as register arithmetic on packed 3-bit tables: a per-lane table load from
// memory here would sit on the dependency chain of every energy lookup.
//   row a (1..4) occupies bits 15(a-1) .. 15(a-1)+14, entry b at 3b inside the row
#define RAFFT_PT_BITS 0x1060c2001005000ULL
#define RAFFT_RT_BITS 0x173850u
__device__ __host__ __forceinline__ int pair_type(int a, int b)
{
    return a ? (int)((RAFFT_PT_BITS >> (15 * (a - 1) + 3 * b)) & 7ULL) : 0;
}
__device__ __host__ __forceinline__ int rtype(int t) { return (int)((RAFFT_RT_BITS >> (3 * t)) & 7u); }

// special-loop key: 3 bits per base, first base in the low bits
__device__ __forceinline__ uint32_t loop_key(const uint8_t *S, int i, int m)
{
    uint32_t k = 0;
    for (int t = 0; t < m; t++) k |= (uint32_t)S[i + t] << (3 * t);
    return k;
}

__device__ inline int e_hairpin(const SmallT *T, const BigT *B, int size, int type, const uint8_t *S, int ci, int cj)
{
    int e = (size <= 30) ? T->hairpin[size] : T->hairpin[30] + B->logext[size];
    if (size < 3) return e;
    if (size == 3 || size == 4 || size == 6) {
        const uint32_t k = loop_key(S, ci, size + 2) | sp_tag(size);
        for (uint32_t sl = sp_slot(k);; sl = (sl + 1) & 127u) {
            const uint32_t kk = T->sp_key[sl];
            if (kk == k) return T->sp_e[sl];
            if (kk == 0) break;
        }
        if (size == 3) return e + (type > 2 ? T->term_au : 0);
    }
    return e + T->mmH[type][S[ci + 1]][S[cj - 1]];
}

// `g`: set when the value read is a rule / model value of the built-in tables (SmallT::lsb)
__device__ inline int e_intloop(const SmallT *T, const BigT *B, int n1, int n2, int type, int type2,
                                int si1, int sj1, int sp1, int sq1, int &g)
{
    const int lsb = T->lsb;
#define RAFFT_TV(x) ([&](int v_) { g |= v_ & lsb; return v_ & ~lsb; }((int)(x)))
    int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1, e, u;
    if (nl == 0) return T->stack[type][type2];
    if (ns == 0) {
        e = (nl <= 30) ? RAFFT_TV(T->bulge[nl]) : RAFFT_TV(T->bulge[30]) + B->logext[nl];
        if (nl == 1) e += T->stack[type][type2];
        else {
            if (type > 2) e += T->term_au;
            if (type2 > 2) e += T->term_au;
        }
        return e;
    }
    if (ns == 1) {
        if (nl == 1) return RAFFT_TV(B->int11[type][type2][si1][sj1]);
        if (nl == 2) {
            if (n1 == 1) return RAFFT_TV(B->int21[type][type2][si1][sq1][sj1]);
            return RAFFT_TV(B->int21[type2][type][sq1][si1][sp1]);
        }
        u = nl + 1;
        e = (u <= 30) ? RAFFT_TV(T->interior[u]) : RAFFT_TV(T->interior[30]) + B->logext[u];
        e += min(T->max_ninio, (nl - ns) * T->ninio);
        e += RAFFT_TV(T->mm1n[type][si1][sj1]) + RAFFT_TV(T->mm1n[type2][sq1][sp1]);
        return e;
    }
    if (ns == 2) {
        if (nl == 2) return RAFFT_TV(B->int22[type][type2][si1][sp1][sq1][sj1]);
        if (nl == 3) {
            e = RAFFT_TV(T->interior[5]) + T->ninio;
            e += RAFFT_TV(T->mm23[type][si1][sj1]) + RAFFT_TV(T->mm23[type2][sq1][sp1]);
            return e;
        }
    }
    u = nl + ns;
    e = (u <= 30) ? RAFFT_TV(T->interior[u]) : RAFFT_TV(T->interior[30]) + B->logext[u];
    e += min(T->max_ninio, (nl - ns) * T->ninio);
    e += RAFFT_TV(T->mmI[type][si1][sj1]) + RAFFT_TV(T->mmI[type2][sq1][sp1]);
    return e;
#undef RAFFT_TV
}

// si1/sj1 < 0: neighbour does not exist (sequence end)
__device__ inline int e_stem(const SmallT *T, int type, int si1, int sj1, bool ext)
{
    int e = 0;
    if (si1 >= 0 && sj1 >= 0) e += ext ? T->mmE[type][si1][sj1] : T->mmM[type][si1][sj1];
    else if (si1 >= 0) e += T->d5[type][si1];
    else if (sj1 >= 0) e += T->d3[type][sj1];
    if (type > 2) e += T->term_au;
    if (!ext) e += T->ml_intern;
    return e;
}

// Energy of the single loop closed by (ci,cj) in 0-based root coordinates; ci < 0 means
// the exterior loop (cj == L).  `pv(x)` returns the partner of x or -1.  S is indexed
// in the same coordinates (S may be a pointer shifted by a window base).
// `bad` is set when a non-canonical pair is met (cannot happen inside the fold).
template <class PV>
__device__ inline int loop_energy(const SmallT *T, const BigT *B, const uint8_t *S, int L, const PV &pv, int ci, int cj, int *bad)
{
    if (ci < 0) {
        int e = 0;
        for (int p = 0; p < L;) {
            int q = pv(p);
            if (q < 0) { p++; continue; }
            int tt = pair_type(S[p], S[q]);
            if (!tt) { *bad |= 1; return 0; }
            e += e_stem(T, tt, p > 0 ? (int)S[p - 1] : -1, q < L - 1 ? (int)S[q + 1] : -1, true);
            p = q + 1;
        }
        return e;
    }
    int type = pair_type(S[ci], S[cj]);
    if (!type) { *bad |= 1; return 0; }
    int nbr = 0, p1 = 0, q1 = 0;
    for (int p = ci + 1; p < cj;) {
        int q = pv(p);
        if (q < 0) { p++; continue; }
        if (!nbr) { p1 = p; q1 = q; }
        nbr++;
        p = q + 1;
    }
    if (nbr == 0) return e_hairpin(T, B, cj - ci - 1, type, S, ci, cj);
    if (nbr == 1) {
        int t2 = pair_type(S[p1], S[q1]);
        if (!t2) { *bad |= 1; return 0; }
        int g = 0;
        const int e1 = e_intloop(T, B, p1 - ci - 1, cj - q1 - 1, type, rtype(t2), S[ci + 1], S[cj - 1], S[p1 - 1], S[q1 + 1], g);
        if (g) *bad |= 2;                 // (bit 1: not an error - the loop's energy involves a rule / model value of the built-in tables)
        return e1;
    }
    int e = 0, u = cj - ci - 1;
    for (int p = ci + 1; p < cj;) {
        int q = pv(p);
        if (q < 0) { p++; continue; }
        int tt = pair_type(S[p], S[q]);
        if (!tt) { *bad |= 1; return 0; }
        e += e_stem(T, tt, S[p - 1], S[q + 1], false);
        u -= q - p + 1;
        p = q + 1;
    }
    e += e_stem(T, rtype(type), S[cj - 1], S[ci + 1], false);
    e += T->ml_closing + u * T->ml_base;
    return e;
}

// ViennaRNA returns `(float)en / 100.` through a float (rafft compares these floats)
__device__ __host__ inline double dcal_to_energy(int dcal)
{
    return (double)(float)((double)(float)dcal / 100.);
}

// commutative 128-bit hash of a pair set: sum over pairs of two independent mixes
__device__ __host__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __host__ inline void pair_hash(int i, int j, uint64_t *h1, uint64_t *h2)
{
    uint64_t k = ((uint64_t)(uint32_t)i << 20) | (uint32_t)j;
    *h1 = mix64(k);
    *h2 = mix64(k ^ 0xa5a5a5a5deadbeefULL);
}

// ---- loop energy from an explicit (virtual) branch list -------------------
// A loop of the structure = closing pair (ci,cj) + the ordered outermost pairs of the
// helices hanging in it.  Candidate stems change loops only by splicing branch lists,
// so the fold kernels never walk a pair table: br[a0..a1) ++ [mid] ++ br[b0..b1).
struct BrList {
    const uint32_t *br;
    int a0, a1, b0, b1, has_mid, mp, mq;
    __device__ __forceinline__ int count() const { return (a1 - a0) + has_mid + (b1 - b0); }
    __device__ __forceinline__ void get(int i, int &p, int &q) const
    {
        const int na = a1 - a0;
        if (has_mid && i == na) { p = mp; q = mq; return; }
        uint32_t u = i < na ? br[a0 + i] : br[b0 + (i - na - has_mid)];
        p = (int)(u & 0xffffu); q = (int)(u >> 16);
    }
};

__device__ inline int loop_energy_br(const SmallT *T, const BigT *B, const uint8_t *S, int L, int ci, int cj, const BrList &bl, int &g)
{
    const int k = bl.count();
    if (ci < 0) {
        int e = 0;
        for (int i = 0; i < k; i++) {
            int p, q;
            bl.get(i, p, q);
            e += e_stem(T, pair_type(S[p], S[q]), p > 0 ? (int)S[p - 1] : -1, q < L - 1 ? (int)S[q + 1] : -1, true);
        }
        return e;
    }
    const int type = pair_type(S[ci], S[cj]);
    if (k == 0) return e_hairpin(T, B, cj - ci - 1, type, S, ci, cj);
    if (k == 1) {
        int p, q;
        bl.get(0, p, q);
        return e_intloop(T, B, p - ci - 1, cj - q - 1, type, rtype(pair_type(S[p], S[q])), S[ci + 1], S[cj - 1], S[p - 1], S[q + 1], g);
    }
    int e = 0, u = cj - ci - 1;
    for (int i = 0; i < k; i++) {
        int p, q;
        bl.get(i, p, q);
        e += e_stem(T, pair_type(S[p], S[q]), S[p - 1], S[q + 1], false);
        u -= q - p + 1;
    }
    e += e_stem(T, rtype(type), S[cj - 1], S[ci + 1], false);
    return e + T->ml_closing + u * T->ml_base;
}

// The same loop energy from prefix sums over the region's branches: pe_ext[i] / pe_ml[i] = sum of the stem
// terms of branches < i as exterior-loop / multiloop branches, psp[i] = sum of their spans.  O(1) per loop
// whatever the number of branches; integer sums, so identical to loop_energy_br.
struct BrPrefix { const int *pe_ext, *pe_ml; const uint16_t *psp; };
__device__ inline int loop_energy_pre(const SmallT *T, const BigT *B, const uint8_t *S, int L, int ci, int cj, const BrList &bl, const BrPrefix &pf, int &g)
{
    const int k = bl.count();
    if (ci < 0) {
        int e = pf.pe_ext[bl.a1] - pf.pe_ext[bl.a0] + pf.pe_ext[bl.b1] - pf.pe_ext[bl.b0];
        if (bl.has_mid)
            e += e_stem(T, pair_type(S[bl.mp], S[bl.mq]), bl.mp > 0 ? (int)S[bl.mp - 1] : -1, bl.mq < L - 1 ? (int)S[bl.mq + 1] : -1, true);
        return e;
    }
    const int type = pair_type(S[ci], S[cj]);
    if (k == 0) return e_hairpin(T, B, cj - ci - 1, type, S, ci, cj);
    if (k == 1) {
        int p, q;
        bl.get(0, p, q);
        return e_intloop(T, B, p - ci - 1, cj - q - 1, type, rtype(pair_type(S[p], S[q])), S[ci + 1], S[cj - 1], S[p - 1], S[q + 1], g);
    }
    int e = pf.pe_ml[bl.a1] - pf.pe_ml[bl.a0] + pf.pe_ml[bl.b1] - pf.pe_ml[bl.b0];
    int u = cj - ci - 1 - ((int)pf.psp[bl.a1] - (int)pf.psp[bl.a0] + (int)pf.psp[bl.b1] - (int)pf.psp[bl.b0]);
    if (bl.has_mid) {
        e += e_stem(T, pair_type(S[bl.mp], S[bl.mq]), S[bl.mp - 1], S[bl.mq + 1], false);
        u -= bl.mq - bl.mp + 1;
    }
    e += e_stem(T, rtype(type), S[cj - 1], S[ci + 1], false);
    return e + T->ml_closing + u * T->ml_base;
}

// number of branches whose 5' end lies before x (branches sorted by p)
__device__ __forceinline__ int br_lower(const uint32_t *br, int nbr, int x)
{
    int lo = 0, hi = nbr;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((int)(br[mid] & 0xffffu) < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The same for four positions at once, in lockstep: the trip count depends on nbr alone (uniform over the wavefront), every step
// loads the four probes together - four dependent chains of LDS round trips become one.  (a = number of branches whose 5' end lies
// before x: all elements below index a are smaller; steps run from the largest power of two <= nbr down to 1.)
__device__ __forceinline__ void br_lower4(const uint32_t *br, int nbr, int x0, int x1, int x2, int x3, int &r0, int &r1, int &r2, int &r3)
{
    int a = 0, b = 0, c = 0, e = 0;
    int step = nbr > 0 ? 1 << (31 - __clz(nbr)) : 0;
    for (; step > 0; step >>= 1) {
        const int ia = a + step, ib = b + step, ic = c + step, ie = e + step;
        const int va = ia <= nbr ? (int)(br[ia - 1] & 0xffffu) : 0x7fffffff, vb = ib <= nbr ? (int)(br[ib - 1] & 0xffffu) : 0x7fffffff;
        const int vc = ic <= nbr ? (int)(br[ic - 1] & 0xffffu) : 0x7fffffff, ve = ie <= nbr ? (int)(br[ie - 1] & 0xffffu) : 0x7fffffff;
        if (va < x0) a = ia;
        if (vb < x1) b = ib;
        if (vc < x2) c = ic;
        if (ve < x3) e = ie;
    }
    r0 = a; r1 = b; r2 = c; r3 = e;
}
