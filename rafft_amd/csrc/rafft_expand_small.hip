// rafft_expand_small.hip - expand kernel for SMALL unpaired regions (gfx950), size classes 4 and 5.
//
// On the benchmark set 63 % of the regions have at most 32 positions and a wavefront per region leaves most lanes of most
// instructions idle (27 of 64 active on average in the general kernel).  Here a TEAM of TL = 16 or 32 lanes expands one
// region, 4 or 2 regions per wavefront, and everything that made the general kernel general is gone:
//   * 2n-1 <= nb_mode, so every lag is searched (rafft/rafft.py:92 takes min(nb_mode, 2n-1) lags): nothing is ranked or
//     selected; the lag value (rafft/utils.py:125-132) only breaks dE ties between kept candidates and is computed for those;
//   * n <= 32: the strand, its reverse and the contiguity of positions are 32-bit masks in registers (ballots over the team);
//     correlation cells of a lag = base masks AND shifted reversed base masks (the direct form scipy itself uses for short
//     inputs, rafft/utils.py:121);
//   * a lane owns lags tl and tl + TL: window_slide (rafft/rafft.py:36-83) and the stem's dE (rafft/rafft.py:97-98) stay in
//     its registers; only kept candidates meet in LDS for the stable dE order (rafft/rafft.py:108);
//   * the bases the energy model looks at all lie ON the loop (closing pair, unpaired positions, branch ends and their
//     neighbours along the loop): they arrive with the packed position / branch entries (Dev::pos_packed) and are scattered
//     into a window of the loop's span in LDS - no copy of the span, no dependent gather.
// Same integer counts, same fp64 recurrence on the same cells in the same order, same tie rules as expand_kernel: the two
// are interchangeable region by region (tests force either).
#pragma once

template <int TL> struct SmLds {
    static constexpr int NL = 2 * TL;                                   // lag slots of a team (2n-1 <= 2 TL - 1)
    static constexpr int off_S = 0;                                     // window of the loop's span, bytes [0, SM_SPAN)
    static constexpr int off_pos = SM_SPAN + 8;                         // uint16 [TL]
    static constexpr int off_br = off_pos + 2 * TL;                     // uint32 [TL]
    static constexpr int off_pe = off_br + 4 * TL;                      // int pe_ext[TL + 1], pe_ml[TL + 1]
    static constexpr int off_psp = off_pe + 8 * (TL + 1);               // uint16 psp[TL + 1]
    static constexpr int off_ck = (off_psp + 2 * (TL + 1) + 7) & ~7;    // sort keys of the kept candidates [NL]
    static constexpr int off_val = off_ck + 8 * NL;                     // their lag values (fp64) [NL]
    static constexpr int per_team = (off_val + 8 * NL + 15) & ~15;
};
// the per-team areas start behind a 4 KiB guard (the energy tables live in it): the window of bases is addressed by SEQUENCE
// position through a pointer shifted back by the span's start (< 4096), which must stay inside the LDS
#define SM_GUARD 4608
static_assert(sizeof(SmallT) <= SM_GUARD, "energy tables must fit the guard area");
#define SM_WG_WAVES 4
template <int TL> constexpr int small_lds_bytes() { return SM_GUARD + SM_WG_WAVES * (64 / TL) * SmLds<TL>::per_team; }

#ifndef RAFFT_SMALL_WAVES
#define RAFFT_SMALL_WAVES 4
#endif

__device__ __forceinline__ uint32_t sm_shift(uint32_t x, int s)          // x >> s for s >= 0, x << -s otherwise; |s| may reach 32
{
    return s >= 0 ? (s < 32 ? x >> s : 0u) : (s > -32 ? x << -s : 0u);
}

// PROD: the production build (no seam, no phase stamps, no diagnostic exits compiled in): fewer live registers, fewer spills
template <int TL, bool PROD = false>
__global__ __launch_bounds__(64 * SM_WG_WAVES, RAFFT_SMALL_WAVES) void expand_small_kernel(Dev d, int cls_arg)
{
    const int cls = cls_arg & 0xFF, diag = PROD ? 0 : cls_arg >> 8;       // diag: diagnostic early exits (RAFFT_SMALL_DIAG)
    if (diag == 1) return;
    using LY = SmLds<TL>;
    constexpr int TPW = 64 / TL;                  // teams per wavefront
    extern __shared__ __align__(16) unsigned char lds_all[];
    {   // hot energy tables: one copy per workgroup
        int *dst = (int *)lds_all;
        const int *src = (const int *)&d.T->s;
        for (int i = threadIdx.x; i < (int)(sizeof(SmallT) / 4); i += 64 * SM_WG_WAVES) dst[i] = src[i];
        __syncthreads();                          // the only workgroup-wide barrier: from here on every wavefront is on its own
    }
    if (diag == 2) return;
    const SmallT *T = (const SmallT *)lds_all;
    const BigT *B = &d.T->b;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tl = lane & (TL - 1), tq = lane / TL, tbase = tq * TL;
    const unsigned long long tmask = ((1ULL << TL) - 1ULL) << tbase;     // my team's lanes
    const unsigned long long lt_lane = (1ULL << lane) - 1ULL, lt_team = (1ULL << tbase) - 1ULL;
    const uint32_t TM = (uint32_t)((1ULL << TL) - 1ULL);
    unsigned char *lds = lds_all + SM_GUARD + (wv * TPW + tq) * LY::per_team;
    uint8_t *Sw = lds + LY::off_S;
    uint16_t *pos = (uint16_t *)(lds + LY::off_pos);
    uint32_t *brl = (uint32_t *)(lds + LY::off_br);
    int *pe_ext = (int *)(lds + LY::off_pe), *pe_ml = pe_ext + (TL + 1);
    uint16_t *psp = (uint16_t *)(lds + LY::off_psp);
    unsigned long long *ck = (unsigned long long *)(lds + LY::off_ck);
    double *val = (double *)(lds + LY::off_val);

    if (d.c->overflow) return;                    // (an arena overflowed earlier in this wave: see expand_kernel)
    const unsigned n_items = d.c->n_work[cls].v;
    if (diag == 3) return;
    const unsigned gw = blockIdx.x * SM_WG_WAVES + wv, n_waves = gridDim.x * SM_WG_WAVES;
    if (gw == 0 && lane == 0) d.c->n_mat = 0;                // the beam step that follows counts its new structures here
    const int shard = gw & (NSHARD - 1);
    unsigned long long st_items = 0, st_n = 0, st_lags = 0, st_nbr = 0;      // statistics (lane 0 of every team)
    unsigned int st_eval = 0, st_guess = 0, st_kguess = 0;                   // stem energies evaluated / involving a rule or model value / kept ones that do (every lane its own)
    const unsigned FETCH = n_items > 4u * TPW * n_waves ? (unsigned)d.fetch_bulk : 1u;            // groups of TPW regions claimed per atomic
    unsigned fetch_base = 0, fetch_left = 0;                                  // uniform across the wavefront
    const FetchPlan fplan = fetch_plan(d, n_items, FETCH * TPW, TPW);
    int fshard = (int)(gw & (NSHARD - 1));
    unsigned long long ffailed = 0;
    unsigned long long slab_base = 0; unsigned slab_left = 0;                 // lane 0 only: reserved candidate slots
    const bool dbg = !PROD && d.dbg.lag != nullptr;                                    // kernel-level seam (one region, team 0)
    const double par_none = 0.0; (void)par_none;

    if (diag == 4) return;
    const bool eprof = !PROD && d.prof_e != nullptr && lane == 0;      // diagnostic phase stamps (RAFFT_TRACE=3)
    unsigned long long eacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, et = eprof ? clock64() : 0, n_rounds = 0;
#define SSTAMP(k) do { if (eprof) { const unsigned long long tn_ = clock64(); eacc[k] += tn_ - et; et = tn_; } } while (0)
    for (;;) {
        wave_sync();                                   // the previous regions' LDS use is over
        if (fetch_left == 0) {
            unsigned fcount = TPW;
            fetch_base = fetch_chunk(d, cls, fplan, fshard, ffailed, fcount);
            if (fetch_base == ~0u) break;
            fetch_left = fcount / TPW;
        }
        const unsigned item0 = fetch_base;
        fetch_base += TPW; fetch_left--;
        if (item0 >= n_items) { fetch_left = 0; continue; }      // (tail of the list's last chunk; other shards may still hold chunks)
        SSTAMP(0);   // fetch
        n_rounds++;
        const bool act = item0 + (unsigned)tq < n_items;           // a team without a region idles through this round
        const int nid = d.work[cls][act ? item0 + tq : item0];
        const NodeRec *nr = &d.nd[nid];
        const int n = nr->n, ci = nr->ci, cj = nr->cj, nbr = nr->nbr, L = nr->L, par_dcal = nr->pdcal;
        const uint16_t *posg = d.pos + nr->pos;
        const uint32_t *brg = d.br + nr->br;
        const uint8_t *codes = d.codes + nr->soff;
        const int sx0 = ci < 0 ? 0 : ci;
        const uint8_t *Sl = (const uint8_t *)Sw - sx0;             // bases by sequence position (only positions ON the loop are filled)
        const int m = 2 * n - 1;

        // ---- the loop: unpaired positions, branch helices, closing pair (rafft/utils.py:24-29 Node); the base codes ride
        // in the packed entries
        int myp = -1, c = 0;
        if (act && tl < n) {
            const int p = posg[tl];
            myp = p & 0x0FFF; c = p >> 12;
            pos[tl] = (uint16_t)myp;
            Sw[myp - sx0] = (uint8_t)c;
        }
        if (act && tl < nbr) {
            const uint32_t u = brg[tl];
            brl[tl] = u & 0x0FFF0FFFu;
            Sw[(int)(u & 0x0FFFu) - sx0] = (uint8_t)((u >> 12) & 0xFu);
            Sw[(int)((u >> 16) & 0x0FFFu) - sx0] = (uint8_t)(u >> 28);
        }
        if (act && ci >= 0 && tl < 2) { const int x = tl ? cj : ci; Sw[x - sx0] = codes[x]; }
        // base masks of the strand (bit t = position t holds A / C / G / U), contiguity with the previous position
        // the strand again, 2 bits per position, in a register pair of every lane (stem_stack_windows; n <= 32: one 64-bit word)
        const uint32_t px_ = row16_or((uint32_t)((c + 3) & 3) << (2 * (tl & 15)));
        const unsigned long long p2 = (unsigned long long)(uint32_t)__shfl((int)px_, tbase + 15, 64) |
                                      (TL == 32 ? (unsigned long long)(uint32_t)__shfl((int)px_, tbase + 31, 64) << 32 : 0ULL);
        const unsigned long long bA = __ballot(c == 1), bC = __ballot(c == 2), bG = __ballot(c == 3), bU = __ballot(c == 4);
        const int pprev = __shfl_up(myp, 1, TL);
        const unsigned long long bg = __ballot(act && tl >= 1 && tl < n && myp - pprev == 1);
        const uint32_t mA = (uint32_t)(bA >> tbase) & TM, mC = (uint32_t)(bC >> tbase) & TM, mG = (uint32_t)(bG >> tbase) & TM,
                       mU = (uint32_t)(bU >> tbase) & TM, mg = (uint32_t)(bg >> tbase) & TM;
        const int rs = 32 - n;                                      // strand reversed: bit j of r? = bit n-1-j of m?
        const uint32_t rA = __brev(mA) >> rs, rC = __brev(mC) >> rs, rG = __brev(mG) >> rs, rU = __brev(mU) >> rs, rg = __brev(mg) >> rs;
        wave_sync();
        SSTAMP(1);   // header + loop fill + masks
        double wgc = d.gc, wau = d.au, wgu = d.gu;      // (the pair weights in vector registers of their own: expand_kernel's cell loop)
        asm volatile("" : "+v"(wgc), "+v"(wau), "+v"(wgu));

        // ---- window_slide of every lag (rafft/rafft.py:36-83); a lane owns lags tl and tl + TL
        int w_nb[2], w_mi[2], w_mj[2];
        double w_sc[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int k = tl + s * TL;
            int mx_nb = 0, mx_i = 0, mx_j = 0;
            double mx_s = 0.0;
            if (act && k < m) {
                const int len = k < n ? k + 1 : 2 * n - k - 1;
                const int len2 = (len >> 1) + (len & 1);
                const int ip0 = k < n ? 0 : k - n + 1, jp0 = k < n ? k : n - 1;
                // eligible cells (pos[jp]-pos[ip] > min_hp) form a prefix; those with jp - ip > min_hp are eligible without looking (expand_kernel)
                const int csure = len - 1 - d.min_hp;
                int lo = csure > 0 ? min((csure + 1) >> 1, len2) : 0, hi = len2;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)pos[jp0 - mid] - (int)pos[ip0 + mid] > d.min_hp) lo = mid + 1; else hi = mid;
                }
                const int lim = lo;
                if (lim > 0) {
                    const int sft = n - 1 - k;              // bit ip of x? = base at position k - ip
                    const uint32_t xA = sm_shift(rA, sft), xC = sm_shift(rC, sft), xG = sm_shift(rG, sft), xU = sm_shift(rU, sft);
                    const uint32_t pGC = d.gc != 0.0 ? ((mG & xC) | (mC & xG)) : 0u;
                    const uint32_t pAU = d.au != 0.0 ? ((mA & xU) | (mU & xA)) : 0u;
                    const uint32_t pGU = d.gu != 0.0 ? ((mG & xU) | (mU & xG)) : 0u;
                    const uint32_t cm = mg & sm_shift(rg, sft - 1) & ~(1u << ip0);      // contiguous with the previous cell (never the first)
                    const uint32_t range = (lim >= 32 ? ~0u : ((1u << lim) - 1u)) << ip0;
                    uint32_t any = (pGC | pAU | pGU) & range;
                    double prev = 0.0;
                    int last_ip = -2, runlen = 0;
                    bool found = false;
                    while (any) {                           // pairing cells only: a zero cell resets the run and can never win
                        const int ip = __ffs((int)any) - 1;
                        any &= any - 1;
                        const uint32_t bit = 1u << ip;
                        const double w8 = (pGC & bit) ? wgc : (pAU & bit) ? wau : wgu;
                        if (ip != last_ip + 1) { prev = 0.0; runlen = 0; }
                        double t = w8;
                        if (cm & bit) t = (prev + w8) * w8;
                        runlen++;
                        if (t >= mx_s) { mx_s = t; mx_nb = runlen; mx_i = ip; mx_j = k - ip; found = true; }
                        prev = t; last_ip = ip;
                    }
                    if (!found) { mx_i = ip0 + lim - 1; mx_j = k - mx_i; }      // last eligible (zero) cell, nb = 0
                }
            }
            w_nb[s] = mx_nb; w_mi[s] = mx_i; w_mj[s] = mx_j; w_sc[s] = mx_s;
        }

        // (round 5) the lags that gave a stem, compacted over the team: half of a small region's lags do, so a lane's second slot
        // is usually empty afterwards - and the second copy of the loop-energy code below (the dE of slot 1: every lane on a path of
        // its own) is skipped by the whole wavefront when no team has more stems than lanes.  The seam reports every lag: as it was.
        int w_k[2] = {tl, tl + TL};                   // the lag a slot holds
        if (!dbg) {
            const unsigned long long sb0 = __ballot(w_nb[0] > 0) & tmask, sb1 = __ballot(w_nb[1] > 0) & tmask;
            const int n0 = __popcll(sb0), nst = n0 + __popcll(sb1);
            uint32_t *sl = (uint32_t *)ck;            // (the sort keys take this place once dE is done)
            if (w_nb[0] > 0) sl[__popcll(sb0 & lt_lane)] = (uint32_t)w_nb[0] | ((uint32_t)w_mi[0] << 8) | ((uint32_t)tl << 16);
            if (w_nb[1] > 0) sl[n0 + __popcll(sb1 & lt_lane)] = (uint32_t)w_nb[1] | ((uint32_t)w_mi[1] << 8) | ((uint32_t)(tl + TL) << 16);
            wave_sync();
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int idx = tl + s * TL;
                const uint32_t e = idx < nst ? sl[idx] : 0u;
                w_nb[s] = (int)(e & 255u); w_mi[s] = (int)((e >> 8) & 255u); w_k[s] = (int)(e >> 16); w_mj[s] = w_k[s] - w_mi[s];
            }
            wave_sync();
        }
        SSTAMP(2);   // window_slide
        // ---- dE of every candidate stem: only the loops it changes (rafft/rafft.py:97-98 evaluates the whole structure)
        // prefix sums of the branches' stem terms: every loop below costs O(1) whatever its number of branches
        {
            int ve = 0, vm = 0, vs = 0;
            if (act && tl < nbr) {
                const uint32_t u = brl[tl];
                const int p = (int)(u & 0xffffu), q = (int)(u >> 16);
                const int tt = pair_type(Sl[p], Sl[q]);
                if (ci < 0) ve = e_stem(T, tt, p > 0 ? (int)Sl[p - 1] : -1, q < L - 1 ? (int)Sl[q + 1] : -1, true);
                vm = e_stem(T, tt, p > 0 ? (int)Sl[p - 1] : 0, q < L - 1 ? (int)Sl[q + 1] : 0, false);
                vs = q - p + 1;
            }
            int xe = ve, xm = vm, xs = vs;
#pragma unroll
            for (int o = 1; o < TL; o <<= 1) {
                const int ye = __shfl_up(xe, o, TL), ym = __shfl_up(xm, o, TL), ys = __shfl_up(xs, o, TL);
                if (tl >= o) { xe += ye; xm += ym; xs += ys; }
            }
            if (act && tl < nbr) { pe_ext[tl] = xe - ve; pe_ml[tl] = xm - vm; psp[tl] = (uint16_t)(xs - vs); }
            if (act && tl == (nbr > 0 ? nbr - 1 : 0)) {
                if (nbr > 0) { pe_ext[nbr] = xe; pe_ml[nbr] = xm; psp[nbr] = (uint16_t)xs; }
                else { pe_ext[0] = 0; pe_ml[0] = 0; psp[0] = 0; }
            }
        }
        wave_sync();
        SSTAMP(3);   // branch prefix sums
        const double par_e = dcal_to_energy(par_dcal);
        const BrPrefix pf{pe_ext, pe_ml, psp};
        int w_dd[2];
        bool w_keep[2];
        {
            const bool anystem = act && (w_nb[0] > 0 || w_nb[1] > 0);
            int e_old = 0, g_old = 0;         // (g: the energy involves a rule / model value of the built-in tables - SmallT::lsb)
            if (anystem) {
                const BrList all_br{brl, 0, nbr, 0, 0, 0, 0, 0};
                e_old = loop_energy_pre(T, B, Sl, L, ci, cj, all_br, pf, g_old);      // the loop as it is (same for every stem)
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                w_dd[s] = 0; w_keep[s] = false;
                const int nb = w_nb[s];
                if (act && nb > 0) {
                    int g = g_old;
                    const int mi = w_mi[s], mj = w_mj[s];
                    const int a0 = pos[mi], b0 = pos[mj], ao = pos[mi - nb + 1], bo = pos[mj + nb - 1];
                    int lo, hi, lo_o, hi_o;
                    br_lower4(brl, nbr, a0, b0, ao, bo, lo, hi, lo_o, hi_o);
                    BrList outer{brl, 0, lo_o, hi_o, nbr, 1, ao, bo};
                    int e_new = loop_energy_pre(T, B, Sl, L, ci, cj, outer, pf, g);
                    BrList inner{brl, lo, hi, 0, 0, 0, 0, 0};
                    e_new += loop_energy_pre(T, B, Sl, L, a0, b0, inner, pf, g);
                    // the stem itself: a contiguous one takes its stacking energies from the packed strand (expand_kernel)
                    if (nb <= 16 && a0 - ao == nb - 1 && bo - b0 == nb - 1)
                        e_new += stem_stack_windows(T, (uint32_t)(p2 >> (2 * (mi - nb + 1))), (uint32_t)(p2 >> (2 * mj)), nb);
                    else {
                    int pa = a0, pb = b0, ty_in = pair_type(Sl[a0], Sl[b0]);
                    for (int t = 1; t < nb; t++) {
                        const int a = pos[mi - t], b = pos[mj + t];
                        const int ty = pair_type(Sl[a], Sl[b]);
                        if (pa == a + 1 && pb == b - 1)
                            e_new += T->stack[ty][rtype(ty_in)];
                        else {
                            const int lo2 = br_lower(brl, nbr, a), hi2 = br_lower(brl, nbr, b);
                            BrList mid{brl, lo2, lo, hi, hi2, 1, pa, pb};
                            e_new += loop_energy_pre(T, B, Sl, L, a, b, mid, pf, g);
                            lo = lo2; hi = hi2;
                        }
                        pa = a; pb = b; ty_in = ty;
                    }
                    }
                    const int ddc = e_new - e_old;
                    w_dd[s] = ddc;
                    const double dE = dcal_to_energy(par_dcal + ddc) - par_e;
                    w_keep[s] = dE < d.min_nrj;                  // rafft/rafft.py:102
                    st_eval++; st_guess += g ? 1 : 0; st_kguess += (g && w_keep[s]) ? 1 : 0;
                }
            }
        }
        SSTAMP(4);   // dE
        // lag values (rafft/utils.py:125-132: exact pair counts, fp64 divide) - of the kept candidates only, they break dE ties
        double w_val[2] = {0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int k = w_k[s];
            if ((w_keep[s] || dbg) && act && k < m) {
                const int sft = n - 1 - k;
                const uint32_t xU = sm_shift(rU, sft), xC = sm_shift(rC, sft);
                const double nAU = 2.0 * (double)__popc(mA & xU), nGC = 2.0 * (double)__popc(mG & xC), nGU = 2.0 * (double)__popc(mG & xU);
                const double raw = nAU * d.au + nGC * d.gc + nGU * d.gu;
                const int nk = k < m - 1 - k ? k : m - 1 - k;
                w_val[s] = raw / ((double)nk + 1.0);
            }
        }
        int w_rank[2] = {0, 0};                       // seam only: rank of the lag by (value desc, lag desc) (rafft/rafft.py:117-118,92)
        if (dbg) {
#pragma unroll
            for (int s = 0; s < 2; s++) { const int k = tl + s * TL; if (act && k < m) val[k] = w_val[s]; }
            wave_sync();
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int k = tl + s * TL;
                if (act && k < m && tq == 0) {
                    int r = 0;
                    for (int y = 0; y < m; y++) { const double v = val[y]; r += (v > w_val[s] || (v == w_val[s] && y > k)) ? 1 : 0; }
                    w_rank[s] = r;
                    d.dbg.lag[r] = k; d.dbg.corval[r] = w_val[s];
                    d.dbg.nb[r] = w_nb[s]; d.dbg.mi[r] = w_mi[s]; d.dbg.mj[r] = w_mj[s]; d.dbg.score[r] = w_sc[s];
                    if (d.dbg.ddcal) d.dbg.ddcal[r] = w_nb[s] > 0 ? w_dd[s] : INT_MIN;
                }
            }
            if (lane == 0 && d.dbg.n_ranked) *d.dbg.n_ranked = m;
            wave_sync();
        }

        // ---- kept candidates: slots in the candidate arena (one reservation per wavefront), stable dE order, emit
        const unsigned long long kb0 = __ballot(w_keep[0]), kb1 = __ballot(w_keep[1]);
        const int nk0 = __popcll(kb0 & tmask), nkept = nk0 + __popcll(kb1 & tmask);
        const int toff = __popcll(kb0 & lt_team) + __popcll(kb1 & lt_team);          // kept by the teams before mine
        const int tot = __popcll(kb0) + __popcll(kb1);
        const int slot[2] = {(int)__popcll(kb0 & tmask & lt_lane), nk0 + (int)__popcll(kb1 & tmask & lt_lane)};
        unsigned long long cbase = 0;
        int ovf = 0;
        if (lane == 0 && tot) {
            if ((unsigned)tot > slab_left) {        // a new slab of candidate slots (the rest of the old one is dropped)
                const unsigned slab = d.cand_shard_cap >= 64u * (unsigned)d.cand_slab ? (unsigned)d.cand_slab : 16u;
                const unsigned want = (unsigned)tot > slab ? (unsigned)tot : slab;
                const unsigned long long b0 = atomicAdd(&d.c->cand[shard].v, (unsigned long long)want);
                if (b0 + want > d.cand_shard_cap) { atomicOr(&d.c->overflow, OVF_CAND); ovf = 1; slab_left = 0; }
                else { slab_base = (unsigned long long)shard * d.cand_shard_cap + b0; slab_left = want; }
            }
            if (!ovf) { cbase = slab_base; slab_base += tot; slab_left -= tot; }
        }
        cbase = __shfl(cbase, 0, 64);
        ovf = __shfl(ovf, 0, 64);
#pragma unroll
        for (int s = 0; s < 2; s++)
            if (w_keep[s]) {
                ck[slot[s]] = ((unsigned long long)((unsigned)w_dd[s] ^ 0x80000000u) << 32) | (unsigned)w_k[s];
                val[slot[s]] = w_val[s];
            }
        wave_sync();
        SSTAMP(5);   // lag values, compaction, slot reservation
        if (!ovf) {
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (w_keep[s]) {
                    const int k = w_k[s];
                    const unsigned long long kx = ck[slot[s]];
                    int rank = 0;
                    for (int y = 0; y < nkept; y++) {
                        const unsigned long long ky = ck[y];
                        if ((ky >> 32) == (kx >> 32)) {          // dE tie: lag rank order = (value desc, lag desc)
                            const int lagq = (int)(ky & 0xFFFFFFFFu);
                            const double qv = val[y];
                            rank += (lagq != k && (qv > w_val[s] || (qv == w_val[s] && lagq > k))) ? 1 : 0;
                        } else
                            rank += ky < kx ? 1 : 0;
                    }
                    const int mi = w_mi[s], mj = w_mj[s], nb = w_nb[s];
                    const int a0 = pos[mi], b0 = pos[mj], ao = pos[mi - nb + 1], bo = pos[mj + nb - 1];
                    uint64_t h1 = 0, h2 = 0;
                    if (a0 - ao == nb - 1 && bo - b0 == nb - 1) stem_hash(a0, b0, ao, bo, &h1, &h2);      // contiguous: the pair hashes telescope
                    else
                        for (int t = 0; t < nb; t++) {
                            uint64_t a, b;
                            pair_hash(pos[mi - t], pos[mj + t], &a, &b);
                            h1 += a; h2 += b;
                        }
                    Cand cd;
                    cd.ddcal = w_dd[s]; cd.mi = (uint16_t)mi; cd.mj = (uint16_t)mj; cd.nb = (uint16_t)nb;
                    { int c0, c1, c2, c3; br_lower4(brl, nbr, a0, b0, ao, bo, c0, c1, c2, c3); cd.set_cuts(c0, c1, c2, c3); }
                    cd.h1 = h1; cd.h2 = h2;
                    d.cand[cbase + toff + rank] = cd;
                    d.cslot[cbase + toff + rank] = 0ULL;      // (both child slots: nobody has asked yet)
                    if (dbg && d.dbg.kept) d.dbg.kept[rank] = w_rank[s];
                }
        }
        if (act && tl == 0) {
            NodeRec *nw = &d.nd[nid];
            nw->cand = cbase + toff;
            nw->ncand = ovf ? 0 : nkept;
            if (dbg && d.dbg.n_ranked) d.dbg.n_ranked[1] = nkept;
            st_items++; st_n += n; st_lags += m; st_nbr += nbr;
        }
        SSTAMP(6);   // order + emit
    }
    if (eprof) {
        for (int k = 0; k < 7; k++) atomicAdd(&d.prof_e[cls * PROF_E + k], eacc[k]);
        atomicAdd(&d.prof_e[cls * PROF_E + 8], n_rounds);
        atomicAdd(&d.prof_e[cls * PROF_E + 9], 1ULL);
    }
#undef SSTAMP
    // statistics: one atomic per wavefront and counter
    for (int o = 32; o > 0; o >>= 1) {
        st_items += __shfl_xor(st_items, o, 64); st_n += __shfl_xor(st_n, o, 64);
        st_lags += __shfl_xor(st_lags, o, 64); st_nbr += __shfl_xor(st_nbr, o, 64);
        st_eval += __shfl_xor(st_eval, o, 64); st_guess += __shfl_xor(st_guess, o, 64); st_kguess += __shfl_xor(st_kguess, o, 64);
    }
    if (lane == 0 && st_items) {
        Counters::StatLine *sl = &d.c->xstat[cls][gw & (NSHARD - 1)];
        atomicAdd(&sl->items, st_items);
        atomicAdd(&sl->n, st_n);
        atomicAdd(&sl->lags, st_lags);
        atomicAdd(&sl->nbr, st_nbr);
        if (T->lsb && st_eval) {          // (built-in tables: stem energies evaluated / involving a rule or model value / kept ones that do)
            atomicAdd(&sl->evals, (unsigned long long)st_eval);
            if (st_guess) atomicAdd(&sl->guessed, (unsigned long long)st_guess);
            if (st_kguess) atomicAdd(&sl->kept_guessed, (unsigned long long)st_kguess);
        }
    }
}
