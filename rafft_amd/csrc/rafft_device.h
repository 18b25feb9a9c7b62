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
    // (round 5) 1024-bit filter in front of that table: bit (size class << 8 | 2-bit codes of the closing pair's 5' base, the first
    // and the last base of the loop and the closing pair's 3' base) is set when some special loop of that size class has them; nine
    // hairpins in ten are told "no special loop" by one look-up instead of building the key of up to eight bases and probing
    uint32_t sp_filter[32];
    // (round 5) stacking energy of a stem's pair t on its pair t-1 by the 2-bit codes (A C G U = 0..3) of the four bases:
    // index = x5(t) | x5(t-1) << 2 | x3(t-1) << 4 | x3(t) << 6 - the order in which the bases of a contiguous stem come out of
    // the packed strands (stem_stack_packed): stack[type(x5(t), x3(t))][rtype(type(x5(t-1), x3(t-1)))], 0 when either is no pair
    int16_t stk4[256];
};
static_assert(sizeof(SmallT) % 16 == 0, "SmallT is copied to LDS in 16-byte pieces");
__host__ __device__ inline uint32_t sp_tag(int size) { return (uint32_t)(size == 3 ? 1 : size == 4 ? 2 : 3) << 28; }
__host__ __device__ inline uint32_t sp_slot(uint32_t tagged) { return (tagged * 2654435761u) >> 25; }
// index into SmallT::sp_filter: size class (1: tri-, 2: tetra-, 3: hexaloops) and the base codes (N=0 A=1 .. U=4; 2 bits each, N
// aliases U - a filter may say "maybe" too often, never "no" wrongly) of the closing pair (c5, c3) and of the loop's ends (l5, l3)
__host__ __device__ inline uint32_t sp_filter_index(int size, int c5, int l5, int l3, int c3)
{
    return ((sp_tag(size) >> 28) << 8) | (uint32_t)((c5 + 3) & 3) | (uint32_t)((l5 + 3) & 3) << 2 | (uint32_t)((l3 + 3) & 3) << 4 | (uint32_t)((c3 + 3) & 3) << 6;
}
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

// (round 5: NOT inlined.  Every expansion of a region calls it from three or four places - the loop as it is, the loop inside a
//  stem, a loop between two pairs of a stem with a gap - and each inlined copy kept its operands alive across the kernel's densest
//  stretch: out of line the one-wavefront kernel spills 6 VGPRs instead of 12, the 256-thread class 17 instead of 35, the small-region
//  kernels 6 instead of 24; same rate, less scratch traffic - tools/kernel_regs.py, tools/ab_many.sh.  e_intloop out of line as well
//  was slower.)
__device__ __noinline__ int e_hairpin(const SmallT *T, const BigT *B, int size, int type, const uint8_t *S, int ci, int cj)
{
    int e = (size <= 30) ? T->hairpin[size] : T->hairpin[30] + B->logext[size];
    if (size < 3) return e;
    const int l5 = S[ci + 1], l3 = S[cj - 1];
    if (size == 3 || size == 4 || size == 6) {
        const uint32_t fi = sp_filter_index(size, S[ci], l5, l3, S[cj]);
        if ((T->sp_filter[fi >> 5] >> (fi & 31u)) & 1u) {      // (SmallT::sp_filter: some special loop of this size has these four bases)
            const uint32_t k = loop_key(S, ci, size + 2) | sp_tag(size);
            for (uint32_t sl = sp_slot(k);; sl = (sl + 1) & 127u) {
                const uint32_t kk = T->sp_key[sl];
                if (kk == k) return T->sp_e[sl];
                if (kk == 0) break;
            }
        }
        if (size == 3) return e + (type > 2 ? T->term_au : 0);
    }
    return e + T->mmH[type][l5][l3];
}

// `g`: set when the value read is a rule / model value of the built-in tables (SmallT::lsb)
// (round 5) Two paths instead of a decision tree with a body per kind of loop.  The lanes of a wavefront evaluate loops of every
// kind at once, and a tree runs each of its bodies in turn for the lanes that took it: stack, bulge, 1 x n, 2 x 3 and the generic
// interior loop are now ONE straight line (every table read unconditionally, the kind chooses by selection), the three tabulated
// small loops (1 x 1, 2 x 1, 2 x 2; HBM / L2) one load whose address is selected.  Same values read, same sums.
__device__ inline int e_intloop(const SmallT *T, const BigT *B, int n1, int n2, int type, int type2,
                                int si1, int sj1, int sp1, int sq1, int &g)
{
    const int lsb = T->lsb;
    const int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1, u = nl + ns;
    if (ns >= 1 && nl <= 2) {
        // int11[type][type2][si1][sj1] | int21[type][type2][si1][sq1][sj1] (n1 == 1), int21[type2][type][sq1][si1][sp1] (n2 == 1) |
        // int22[type][type2][si1][sp1][sq1][sj1]
        const bool c11 = nl == 1, c22 = ns == 2, sw = !c11 && !c22 && n1 != 1;
        const int ta = sw ? type2 : type, tb = sw ? type : type2;
        const int ia = sw ? sq1 : si1;
        const int ib = c11 ? sj1 : c22 ? sp1 : (sw ? si1 : sq1);
        const int ic = c22 ? sq1 : (sw ? sp1 : sj1);
        const int i2 = ((ta * 7 + tb) * 5 + ia) * 5 + ib, i3 = i2 * 5 + ic, i4 = i3 * 5 + sj1;
        static_assert(offsetof(BigT, int11) == 0 && offsetof(BigT, int21) == 2 * 1225 && offsetof(BigT, int22) == 2 * (1225 + 6125), "BigT layout");
        const int off = c11 ? i2 : c22 ? 1225 + 6125 + i4 : 1225 + i3;
        const int v = ((const int16_t *)B)[off];
        g |= v & lsb;
        return v & ~lsb;
    }
    static_assert(offsetof(SmallT, bulge) == offsetof(SmallT, hairpin) + 31 * 4 && offsetof(SmallT, interior) == offsetof(SmallT, hairpin) + 62 * 4, "SmallT layout");
    static_assert(offsetof(SmallT, mmI) == offsetof(SmallT, mmH) + 350 && offsetof(SmallT, mm1n) == offsetof(SmallT, mmH) + 700 && offsetof(SmallT, mm23) == offsetof(SmallT, mmH) + 1050, "SmallT layout");
    const int stk = T->stack[type][type2];
    const bool bul = ns == 0, c23 = ns == 2 && nl == 3;
    const int sz = ((const int32_t *)((const char *)T + offsetof(SmallT, hairpin)))[(bul ? 31 : 62) + (u < 30 ? u : 30)];   // bulge[min(u, 30)] | interior[min(u, 30)]
    const int16_t *mm = (const int16_t *)((const char *)T + offsetof(SmallT, mmH)) + (ns == 1 ? 350 : c23 ? 525 : 175);    // mm1n | mm23 | mmI
    const int m1 = mm[(type * 5 + si1) * 5 + sj1], m2 = mm[(type2 * 5 + sq1) * 5 + sp1];
    const int nin0 = T->ninio, ninx = T->max_ninio, tau = T->term_au;
    int e = sz & ~lsb;
    if (u > 30) e += B->logext[u];
    const int dn = (nl - ns) * nin0;
    const int nin = c23 ? nin0 : (ninx < dn ? ninx : dn);
    const int au = (type > 2 ? tau : 0) + (type2 > 2 ? tau : 0);
    const int tail = bul ? (nl == 1 ? stk : au) : nin + (m1 & ~lsb) + (m2 & ~lsb);
    const int gt = bul ? (sz & lsb) : ((sz | m1 | m2) & lsb);
    g |= nl == 0 ? 0 : gt;
    return nl == 0 ? stk : e + tail;
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
__device__ __host__ inline void pair_mix(int i, int j, uint64_t *f1, uint64_t *f2)
{
    uint64_t k = ((uint64_t)(uint32_t)i << 20) | (uint32_t)j;
    *f1 = mix64(k);
    *f2 = mix64(k ^ 0xa5a5a5a5deadbeefULL);
}
// (round 5) The hash of ONE pair is F(i, j) - F(i - 1, j + 1), F = the two mixes above.  Along a run of stacked pairs
// (i - t, j + t) the terms telescope: a contiguous stem of any length costs two evaluations of F - at its innermost pair and one
// place beyond its outermost pair - where rounds 1-4 mixed every pair of it.  The sum over a pair SET is still a function of the set
// alone (whatever stems it was assembled from: the partial sums telescope exactly, modulo 2^64), and two different sets differ in the
// ends of their maximal diagonal runs, i.e. in which values of F enter with +1 and -1 - no weaker than the plain sum of mixes.
__device__ __host__ inline void pair_hash(int i, int j, uint64_t *h1, uint64_t *h2)
{
    uint64_t a1, a2, b1, b2;
    pair_mix(i, j, &a1, &a2);
    pair_mix(i - 1, j + 1, &b1, &b2);
    *h1 = a1 - b1; *h2 = a2 - b2;
}
// ... of the contiguous stem with innermost pair (a0, b0) and outermost pair (ao, bo) = (a0 - nb + 1, b0 + nb - 1)
__device__ __host__ inline void stem_hash(int a0, int b0, int ao, int bo, uint64_t *h1, uint64_t *h2)
{
    uint64_t a1, a2, b1, b2;
    pair_mix(a0, b0, &a1, &a2);
    pair_mix(ao - 1, bo + 1, &b1, &b2);
    *h1 = a1 - b1; *h2 = a2 - b2;
}

// ---- stacking energies of a contiguous stem from packed strands (round 5) ----
// `P2`: the region's bases, 2 bits per position (A C G U = 0..3; N never pairs), 16 positions per word, one word of slack behind
// the last.  The 5' strand of a stem whose innermost pair sits at region positions (mi, mj) is positions mi - nb + 1 .. mi, its
// 3' strand mj .. mj + nb - 1: two 32-bit windows hold both for nb <= 16, and the stacking term of pair t on pair t - 1 is ONE
// look-up in SmallT::stk4 by four 2-bit codes - where the loop used to read two positions, two bases and the stack table per pair.
__device__ __forceinline__ uint32_t strand_window(const uint32_t *P2, int start)
{
    const int q = start >> 4;
    return __builtin_amdgcn_alignbit(P2[q + 1], P2[q], (uint32_t)(start & 15) * 2u);
}
__device__ __forceinline__ int stem_stack_windows(const SmallT *T, uint32_t w5, uint32_t w3, int nb)
{
    int e = 0;
    for (int t = 1; t < nb; t++) {
        const uint32_t i5 = (w5 >> (2 * (nb - 1 - t))) & 15u;     // x5(t) | x5(t-1) << 2
        const uint32_t i3 = (w3 >> (2 * (t - 1))) & 15u;          // x3(t-1) | x3(t) << 2
        e += T->stk4[i5 | (i3 << 4)];
    }
    return e;
}
// inclusive prefix sum over the 64 lanes of a wavefront (all of them active) in six DPP additions - row_shr 1, 2, 4, 8 inside the
// rows of 16, then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into the upper half (row_bcast:31) - where six
// rounds of __shfl_up are six trips through the LDS crossbar
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}
// the same over the rows of 16 lanes (a team of materialize_team_kernel is one row): four row_shr additions
__device__ __forceinline__ int row16_incl_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    return x;
}
// one step of the OR-reduction over a row of 16 lanes (DPP row_shr): lane 15 of every row ends up with the OR of the row
__device__ __forceinline__ uint32_t row16_or(uint32_t x)
{
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    return x;
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
