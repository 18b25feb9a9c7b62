// Host-side check of the round-5 device helpers that have host-callable forms (rafft_amd/csrc/rafft_device.h), built host-only with
// hipcc by tests/test_host.py: the telescoping pair hash and the packed-strand stacking table.  Test infrastructure, not product code.
//   * stem_hash(a0, b0, ao, bo) == sum of pair_hash over the stem's pairs, for any length;
//   * a pair set hashes the same whatever stems it is assembled from (a stem of 9 = a stem of 4 + a stem of 5 on the same diagonal);
//   * sets that differ in one pair, or by the Prouhet-Tarry-Escott pattern {0,4,7,11} / {1,2,9,10} on one diagonal (equal sums of the
//     first three powers: what a polynomial hash in the position would confuse), hash differently;
//   * stk4 as scaled_tables fills it == stack[type(pair t)][rtype(type(pair t-1))] for every quadruple of bases, and
//     stem_stack over windows built like the kernels' equals the pair-by-pair sum for random stems of up to 16 pairs.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include "../../rafft_amd/csrc/rafft_device.h"
#include "../../rafft_amd/csrc/rafft_params.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fails++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

int main()
{
    std::mt19937_64 rng(5);
    // ---- telescoping hash
    for (int it = 0; it < 20000; it++) {
        const int nb = 1 + (int)(rng() % 40), a0 = nb + (int)(rng() % 30000), b0 = a0 + 4 + (int)(rng() % 2000);
        const int ao = a0 - nb + 1, bo = b0 + nb - 1;
        uint64_t s1 = 0, s2 = 0, t1, t2;
        for (int t = 0; t < nb; t++) { uint64_t x, y; pair_hash(a0 - t, b0 + t, &x, &y); s1 += x; s2 += y; }
        stem_hash(a0, b0, ao, bo, &t1, &t2);
        CHECK(s1 == t1 && s2 == t2, "stem_hash != sum of pair_hash (nb %d)", nb);
        if (nb >= 2) {      // the same pairs as two stems
            const int k = 1 + (int)(rng() % (nb - 1));
            uint64_t u1, u2, v1, v2;
            stem_hash(a0, b0, a0 - k + 1, b0 + k - 1, &u1, &u2);
            stem_hash(a0 - k, b0 + k, ao, bo, &v1, &v2);
            CHECK(u1 + v1 == t1 && u2 + v2 == t2, "two stems != one stem");
        }
        uint64_t d1, d2;      // one pair less
        stem_hash(a0, b0, ao + 1, bo - 1, &d1, &d2);
        CHECK(nb == 1 || d1 != t1, "a pair less, same hash");
    }
    {
        const int P[4] = {0, 4, 7, 11}, Q[4] = {1, 2, 9, 10};
        uint64_t p1 = 0, p2 = 0, q1 = 0, q2 = 0, x, y;
        for (int k = 0; k < 4; k++) { pair_hash(100 + P[k], 400 - P[k], &x, &y); p1 += x; p2 += y; pair_hash(100 + Q[k], 400 - Q[k], &x, &y); q1 += x; q2 += y; }
        CHECK(p1 != q1 && p2 != q2, "Prouhet-Tarry-Escott sets collide");
    }
    // ---- stk4 and the window form of the stacking sum
    rafft_par::ParamSet P;
    rafft_par::builtin(P);
    EnergyTables *h = new EnergyTables();
    std::string err;
    CHECK(rafft_par::scaled_tables(P, 37.0, h, err), "scaled_tables: %s", err.c_str());
    for (int i = 0; i < 256; i++) {
        const int x5t = (i & 3) + 1, x5p = ((i >> 2) & 3) + 1, x3p = ((i >> 4) & 3) + 1, x3t = ((i >> 6) & 3) + 1;
        const int ty = pair_type(x5t, x3t), ti = pair_type(x5p, x3p);
        CHECK(h->s.stk4[i] == ((ty && ti) ? h->s.stack[ty][rtype(ti)] : 0), "stk4[%d]", i);
    }
    const char pr[6][2] = {{2, 3}, {3, 2}, {3, 4}, {4, 3}, {1, 4}, {4, 1}};      // CG GC GU UG AU UA as base codes
    for (int it = 0; it < 20000; it++) {
        const int nb = 1 + (int)(rng() % 16);
        int c5[16], c3[16];                    // pair t: (c5[t], c3[t]), t = 0 innermost
        for (int t = 0; t < nb; t++) { const int k = (int)(rng() % 6); c5[t] = pr[k][0]; c3[t] = pr[k][1]; }
        int want = 0;
        for (int t = 1; t < nb; t++) want += h->s.stack[pair_type(c5[t], c3[t])][rtype(pair_type(c5[t - 1], c3[t - 1]))];
        // windows as strand_window returns them: w5 holds the 5' strand from its OUTERMOST base upwards, w3 the 3' strand from the innermost
        uint32_t w5 = 0, w3 = 0;
        for (int t = 0; t < nb; t++) { w5 |= (uint32_t)((c5[t] + 3) & 3) << (2 * (nb - 1 - t)); w3 |= (uint32_t)((c3[t] + 3) & 3) << (2 * t); }
        int got = 0;
        for (int t = 1; t < nb; t++) got += h->s.stk4[((w5 >> (2 * (nb - 1 - t))) & 15u) | (((w3 >> (2 * (t - 1))) & 15u) << 4)];
        CHECK(got == want, "stacking sum from windows %d != %d (nb %d)", got, want, nb);
    }
    // ---- the special-hairpin filter never says "no" to a listed loop
    int listed = 0;
    for (int sl = 0; sl < 128; sl++) {
        const uint32_t k = h->s.sp_key[sl];
        if (!k) continue;
        listed++;
        const int size = (k >> 28) == 1 ? 3 : (k >> 28) == 2 ? 4 : 6;
        auto base = [&](int t) { return (int)((k >> (3 * t)) & 7u); };
        const uint32_t fi = sp_filter_index(size, base(0), base(1), base(size), base(size + 1));
        CHECK((h->s.sp_filter[fi >> 5] >> (fi & 31u)) & 1u, "filter misses a special loop of size %d", size);
    }
    int bits = 0;
    for (int w = 0; w < 32; w++) bits += __builtin_popcount(h->s.sp_filter[w]);
    CHECK(listed > 20 && bits > 0 && bits <= listed, "filter: %d bits for %d loops", bits, listed);
    printf("device helpers: %d failures; %d special loops behind %d filter bits\n", fails, listed, bits);
    return fails ? 1 : 0;
}
