// rafft_kernels.hip - the HIP kernels of the fold hot path (gfx950 / MI355X only).
//
//   expand_kernel       persistent workgroups (one wavefront for the common size class)
//                       fetch unpaired regions from a work list: correlation of the region
//                       with itself (rafft/utils.py:115-132; popcounts on bit masks for short
//                       regions, LDS-resident packed complex FFTs otherwise), selection of the
//                       nb_mode best lags (rafft/rafft.py:117-118,92), window_slide
//                       (rafft/rafft.py:36-83), local Turner dE of every candidate stem from
//                       prefix sums over the loop's branch list + filter/sort
//                       (rafft/rafft.py:86-109)
//                       Template switch PROD: production builds with the debug seam, the phase
//                       stamps and - for the classes whose regions all take the popcount
//                       correlation - the FFT compiled out (no register spills).
//   expand_small_kernel  (rafft_expand_small.hip) the same for regions of up to 16 / 32 positions:
//                       teams of 16 / 32 lanes, four or two regions per wavefront
//   beam_step_kernel    one workgroup per sequence: helix combination in product order, flat
//                       over all parents of the beam, with `seen` dedupe and the max_branch
//                       rule, stable energy sort and beam cut (rafft/rafft.py:176-214)
//   materialize_kernel  one wavefront per new beam member: dot-bracket row + child
//                       regions (rafft/rafft.py:127-152, rafft/utils.py:141-152)
//   dedupe_kernel       identical loops reached through different structures share one
//                       expansion (no counterpart in the reference, which recomputes)
//   output_kernel       gathers dot-bracket rows (rafft/utils.py:42-50)
//   eval_kernel         whole-structure energy (rafft/utils.py:135-138), C-ABI hook
//
// This is bandwidth/latency-bound small-FFT + integer table work: no MFMA.
#include "rafft_kernels.h"

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) // a * conj(b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// "this value is needed HERE": keeps the compiler from sinking a load below a branch that may not need it.  Loads issued back to
// back are one round trip; a load sunk to its first use, behind an early exit, is a dependent round trip of its own.
__device__ __forceinline__ void pin(unsigned int &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(int &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(unsigned long long &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(ulonglong2 &x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); }
__device__ __forceinline__ void pin(uint4 &x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }

// exclusive prefix sum of one int per thread over the workgroup; returns total in *tot.
// `scratch` needs (NT/64 + 1) ints of LDS.
template <int NT>
__device__ inline int block_exscan(int v, int *scratch, int *tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = wave_incl_scan(v);           // (round 5: six DPP additions - every caller has the whole wavefront active - instead of six __shfl_up)
    if (NT == 64) {
        *tot = __builtin_amdgcn_readlane(x, 63);
        return x - v;
    }
    __syncthreads();
    if (lane == 63) scratch[wv] = x;
    __syncthreads();
    int base = 0, t = 0;
    for (int i = 0; i < NT / 64; i++) {
        int s = scratch[i];
        if (i < wv) base += s;
        t += s;
    }
    *tot = t;
    return base + x - v;
}

// The same for a 0/1 flag: a ballot and two bit counts instead of six shuffle steps per wavefront.
template <int NT>
__device__ inline int block_exscan_flag(int f, int *scratch, int *tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(f != 0);
    const int pre = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));   // set bits below my lane
    const int wtot = __popcll(bal);
    (void)lane;
    if (NT == 64) {
        *tot = wtot;
        return pre;
    }
    __syncthreads();
    if (lane == 0) scratch[wv] = wtot;
    __syncthreads();
    int base = 0, t = 0;
    for (int i = 0; i < NT / 64; i++) {
        int s = scratch[i];
        if (i < wv) base += s;
        t += s;
    }
    *tot = t;
    return base + pre;
}

// Exact top-K selection for a workgroup: moves the K smallest of the N DISTINCT 64-bit keys in LDS to
// keys[0..K) (unordered).  Byte-wise radix select from the most significant byte: 8 passes over the
// keys with a 256-bin LDS histogram instead of sorting all N.  `hist` needs 256 ints, `sh` 32 ints.
template <int NT>
__device__ inline void select_smallest_inplace(unsigned long long *keys, int N, int K, int *hist, int *sh)
{
    const int tid = threadIdx.x;
    if (K >= N) return;
    // bytes in which no two keys differ need no pass (the keys of the beam step are (energy + bias) << 32 | generation order: of
    // their eight bytes three or four vary) - one OR-reduction of key ^ keys[0] finds them
    unsigned long long diff = 0;
    const unsigned long long key0 = keys[0];
    for (int i = tid; i < N; i += NT) diff |= keys[i] ^ key0;
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (tid < 2) sh[26 + tid] = 0;
    __syncthreads();
    if ((tid & 63) == 0) { atomicOr((unsigned int *)&sh[26], (unsigned int)diff); atomicOr((unsigned int *)&sh[27], (unsigned int)(diff >> 32)); }
    __syncthreads();
    diff = ((unsigned long long)(unsigned int)sh[27] << 32) | (unsigned int)sh[26];
    unsigned long long prefix = 0;
    int kk = K;
    bool take_le = false;              // every key of the threshold bin is wanted: nothing below that byte needs looking at
    for (int pass = 7; pass >= 0 && !take_le; pass--) {
        if (((diff >> (8 * pass)) & 255ULL) == 0) { prefix |= key0 & (255ULL << (8 * pass)); continue; }
        for (int i = tid; i < 256; i += NT) hist[i] = 0;
        __syncthreads();
        const int sh_hi = 8 * (pass + 1);
        for (int i = tid; i < N; i += NT) {
            const unsigned long long key = keys[i];
            if (pass == 7 || (key >> sh_hi) == (prefix >> sh_hi)) atomicAdd(&hist[(int)((key >> (8 * pass)) & 255ULL)], 1);
        }
        __syncthreads();
        // bucket holding the kk-th smallest key of the current group
        int run = 0;
        for (int base = 0; base < 256; base += NT) {
            const int b = base + tid;
            const int h = b < 256 ? hist[b] : 0;
            int tot, ex = block_exscan<NT>(h, sh, &tot);
            if (b < 256 && run + ex < kk && kk <= run + ex + h) { sh[28] = b; sh[29] = kk - (run + ex); sh[30] = h; }
            run += tot;
            __syncthreads();
        }
        prefix |= (unsigned long long)(unsigned)sh[28] << (8 * pass);
        kk = sh[29];
        take_le = kk == sh[30];
        __syncthreads();
        if (take_le) prefix |= (pass > 0) ? ((1ULL << (8 * pass)) - 1ULL) : 0ULL;      // (the largest key the bin can hold)
    }
    // `prefix` is now the K-th smallest key (or, stopped early, an upper bound of the wanted bin that no unwanted key reaches):
    // keep every key <= prefix - exactly K, keys are distinct
    int outn = 0;
    for (int base = 0; base < N; base += NT) {
        const int i = base + tid;
        unsigned long long key = 0;
        int f = 0;
        if (i < N) { key = keys[i]; f = key <= prefix ? 1 : 0; }
        int tot, ex = block_exscan_flag<NT>(f, sh, &tot);     // (barriers inside: all reads of this slab are done)
        if (f) keys[outn + ex] = key;                     // outn + ex <= i: never clobbers an unread key
        outn += tot;
        __syncthreads();
    }
}

struct PlainView {
    const int16_t *pt;
    __device__ __forceinline__ int operator()(int x) const { return pt[x]; }
};

// ------------------------------------------------------------ expand kernel

// 64 bits of the bit string X (W words) starting at bit `start` (may be negative / past the end -> zeros)
__device__ __forceinline__ unsigned long long mask_window(const unsigned long long *X, int W, int start)
{
    if (start >= 64 * W || start <= -64) return 0ULL;
    const int q = start >> 6, bsh = start & 63;           // arithmetic shift: floor division
    const unsigned long long lo = (q >= 0 && q < W) ? X[q] : 0ULL;
    const unsigned long long hi = (q + 1 >= 0 && q + 1 < W) ? X[q + 1] : 0ULL;
    return bsh ? (lo >> bsh) | (hi << (64 - bsh)) : lo;
}

// Bit masks of a region: forward masks F[0..3] = positions holding A, C, G, U, F[4] = contiguity with the previous
// position; R[0..3] = the base strings reversed (bit j of R = bit n-1-j of F), R[4] = "contiguous with the NEXT position" reversed
// (bit j of R[4] = bit n-j of F[4]) - so that for the cell (ip, jp = lag - ip) of a diagonal all five reversed strings are read at
// the same bit n - 1 - lag + ip.  W words each, F first, then R.  One synchronisation inside (the caller adds the one behind).
// (`code_at(t)`: the base code of the region's position t - an LDS array, or the sequence's codes read through `pos` for the class
//  whose regions are too big for an LDS copy)
template <int NT, class CodeAt>
__device__ inline void build_masks(unsigned long long *F, unsigned long long *R, int W, int n, const CodeAt &code_at, const uint16_t *pos, int tid)
{
    for (int wq = tid >> 6; wq < W; wq += NT / 64) {       // each wavefront ballots whole 64-bit words
        const int t = wq * 64 + (tid & 63);
        const int c0 = t < n ? code_at(t) : 0;
        const unsigned long long bA = __ballot(c0 == 1), bC = __ballot(c0 == 2), bG = __ballot(c0 == 3), bU = __ballot(c0 == 4);
        const unsigned long long bg = __ballot(t >= 1 && t < n && (int)pos[t] - (int)pos[t > 0 ? t - 1 : 0] == 1);
        if ((tid & 63) == 0) { F[0 * W + wq] = bA; F[1 * W + wq] = bC; F[2 * W + wq] = bG; F[3 * W + wq] = bU; F[4 * W + wq] = bg; }
    }
    if (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads();
    // reverse the whole 64 W-bit string (word order and bit order), then shift the n live bits down:
    // R bit j = T bit (j + 64 W - n) with T[w] = brev(F[W-1-w]); bits of F past n are zero.  The contiguity string is shifted one
    // bit less (bit j = F[4] bit n - j = T bit j + 64 W - n - 1; its bit 0 is F[4] bit n: zero)
    for (int idx = tid; idx < 5 * W; idx += NT) {
        const int which = idx / W, w = idx - which * W;
        const int s0 = 64 * w + 64 * W - n - (which == 4 ? 1 : 0), q = s0 >> 6, bsh = s0 & 63;      // (arithmetic shift: q = -1 for s0 = -1)
        const unsigned long long lo = q >= 0 && q < W ? __brevll(F[which * W + W - 1 - q]) : 0ULL;
        const unsigned long long hi = q + 1 < W ? __brevll(F[which * W + W - 2 - q]) : 0ULL;
        R[which * W + w] = bsh ? (lo >> bsh) | (hi << (64 - bsh)) : lo;
    }
}

// Next chunk of CH work items of class `cls` for this wavefront (all 64 lanes call it; `shard` is the wavefront's current
// shard, `failed` the shards it has found empty - both kept between calls): returns the first item of the chunk, or ~0u
// when every chunk of the list has been handed out.  Chunk c belongs to shard c % NSHARD; the fast path is ONE returning
// atomic on the wavefront's own shard (64 cursors, 64 bytes apart: 0.56 ns per atomic chip-wide against 11.4 ns on a
// single cursor, tools/micro/atomic_spacing.hip).  Whoever claims the last chunk of a shard sets its bit in wdone[cls];
// a wavefront that finds its shard empty reads that ONE word and moves to a shard that still has chunks.
// Chunks taper: the first Dev::taper_pct percent of a list go out CH items at a time, the rest CT at a time (`count` says which) -
// the wavefronts that finish a launch are then a fraction of a big chunk apart, not a whole one.
struct FetchPlan { unsigned cA, nA, chunks_total, CH, CT; unsigned long long exist; };      // (what fetch_chunk needs of a list: computed once per launch)
__device__ inline FetchPlan fetch_plan(const Dev &d, unsigned n_items, unsigned CH, unsigned CT)
{
    FetchPlan f;
    f.CH = CH; f.CT = CT;
    f.cA = CH > CT ? (unsigned)(((unsigned long long)n_items * (unsigned long long)d.taper_pct / 100ULL) / CH) : n_items / CH;      // big chunks
    f.nA = f.cA * CH;                                                                                                            // items in them
    f.chunks_total = f.cA + (n_items - f.nA + CT - 1) / CT;
    f.exist = f.chunks_total >= NSHARD ? ~0ULL : ((1ULL << f.chunks_total) - 1ULL);      // shards that hold any chunk
    return f;
}
__device__ inline unsigned fetch_chunk(const Dev &d, int cls, const FetchPlan &f, int &shard, unsigned long long &failed, unsigned &count)
{
    const int lane = threadIdx.x & 63;
    const unsigned cA = f.cA, nA = f.nA, chunks_total = f.chunks_total, CH = f.CH, CT = f.CT;
    const unsigned long long exist = f.exist;
    for (;;) {
        if ((exist >> shard) & ~(failed >> shard) & 1ULL) {
            const unsigned cnt = (chunks_total - (unsigned)shard + (NSHARD - 1)) / NSHARD;                     // chunks of this shard
            unsigned k = 0;
            if (lane == 0) k = (unsigned)atomicAdd(&d.c->wcur[cls][shard].v, 1ULL);
            k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
            if (k < cnt) {
                if (k == cnt - 1 && lane == 0) atomicOr(&d.c->wdone[cls], 1ULL << shard);
                const unsigned c = (unsigned)shard + NSHARD * k;
                if (c < cA) { count = CH; return c * CH; }
                count = CT;
                return nA + (c - cA) * CT;
            }
            failed |= 1ULL << shard;
        }
        unsigned long long done = 0;
        if (lane == 0) done = __hip_atomic_load(&d.c->wdone[cls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        done = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(done >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)done);
        const unsigned long long live = exist & ~(done | failed);
        if (!live) return ~0u;
        const unsigned long long rot = shard ? ((live >> shard) | (live << (64 - shard))) : live;
        shard = (shard + __ffsll((long long)rot) - 1) & (NSHARD - 1);
    }
}
static_assert(NSHARD == 64, "fetch_chunk reads one work cursor per lane");

#ifndef RAFFT_EXPAND256_PROD_WAVES
#define RAFFT_EXPAND256_PROD_WAVES 4
#endif
#ifndef RAFFT_EXPAND64_WAVES
#define RAFFT_EXPAND64_WAVES 3        // <= 168 VGPRs (12 B/lane of scratch): its LDS allows three wavefronts per SIMD anyway; a cap of 128 spilled 152 B/lane
#endif
// Synchronisation inside one region's work.  The one-wavefront class needs no s_barrier: the LDS operations of a
// wavefront execute in program order, so a compiler fence at wavefront scope is all it takes - and, unlike
// __syncthreads(), it does not wait for the global loads in flight.  That also lets WPB wavefronts share a workgroup
// (each with its own slice of LDS and its own regions, never waiting for each other) and with it ONE copy of the hot
// energy tables in LDS.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#define ESYNC() do { if (NT == 64) wave_sync(); else __syncthreads(); } while (0)

// LONGSEQ: 0 - the usual case: the bases of the loop are staged in LDS.
//          1 - sequences longer than 4096 nt: the bases are read from HBM/L2 (no room for them beside the FFT buffers).
//          2 - regions of more than 4096 positions (FFT size > 8192, whose two complex buffers exceed the LDS): the
//              correlation is the exact direct form on multi-word bit masks - popcount(base mask AND shifted reversed base
//              mask), the analogue of scipy's own direct branch (rafft/utils.py:121) - and the lag values live in a
//              per-workgroup scratch in HBM instead of LDS.  Same integer pair counts, same fp64 values, same ranking.
// PROD: the production build of a class without FFT buffers (no seam, no forced FFT, no negative weights, no diagnostics): the
// debug-seam stores, the phase stamps, the FFT and the cell-by-cell window_slide are compiled out - fewer live registers, fewer spills.
// PROD 2: the same for a class that keeps its FFT (regions beyond Dev::direct_n positions): only the diagnostics go.
// PROD 3 (with LONGSEQ 2): the same kernel as the class for regions beyond 4096 positions, compiled for FOUR wavefronts per SIMD: it
//         serves the regions of 1025-4096 positions of ordinary sequences when the host routes them here (RAFFT_C3_DIRECT, class_cfg) -
//         ~50 KiB of LDS instead of the 150 KiB of the FFT plan, two or three workgroups per CU instead of one.
template <int NT, bool TAB_LDS, int WPB, int LONGSEQ = 0, int PROD = 0>
__global__ __launch_bounds__(NT * WPB, (NT == 64 ? (WPB == 16 ? 4 : RAFFT_EXPAND64_WAVES) : NT == 256 ? (PROD == 1 ? RAFFT_EXPAND256_PROD_WAVES : 3) : (PROD == 3 ? 4 : 2))) void expand_kernel(Dev d, int cls_arg, int Pmax, int Lmax, int nmax, int brmax, int Kmax)
{
    const int cls = cls_arg & 0xFF;
    const DebugOut dbg = PROD ? DebugOut{} : d.dbg;
    const int rep = PROD ? 0 : d.rep, force_fft = PROD ? 0 : d.force_fft;
    unsigned long long *const prof_e = PROD ? nullptr : d.prof_e;
    const bool dry = !PROD && (cls_arg & 0x100) != 0;      // diagnostic (RAFFT_TWICE=2): everything but the result stores
    const int skip_lvl = PROD ? 0 : (cls_arg >> 9) & 15;      // diagnostic (RAFFT_TWICE=3..7): a region stops after window_slide (1), ranking (2), lag values (3), FFTs (4), LDS fill (5)

    static_assert(WPB == 1 || NT == 64, "only the one-wavefront class packs several wavefronts into a workgroup");
    static_assert(LONGSEQ == 0 || NT > 64, "long sequences never reach the one-wavefront class");
    extern __shared__ __align__(16) unsigned char lds_all[];
    const bool nofft = PROD == 1 || (cls_arg & 0x2000) != 0;   // the host promises: no seam, no forced FFT, no negative weights, every region within Dev::direct_n
    // (the class for regions beyond 4096 positions - 512 threads, lag values in HBM - keeps no LDS copy of the base codes: 32 768
    //  positions x 2 bytes are 64 KiB of its plan already; the FFT-free class for 1025-4096 positions, 256 threads, does)
    constexpr bool CODE_LDS = !(LONGSEQ == 2 && NT == 512);
    const ExpandLds lay = expand_lds(Pmax, Lmax, nmax, brmax, Kmax, TAB_LDS, WPB, nofft, NT, CODE_LDS);
    const int tid = threadIdx.x % NT;                 // position inside this region's team (a wavefront / the workgroup)
    const int team = threadIdx.x / NT;                // wavefront of the workgroup (0 when the workgroup is the team)
    const unsigned gteam = blockIdx.x * WPB + team, n_teams = gridDim.x * WPB;
    unsigned char *const lds = lds_all + team * lay.per_team;
    const SmallT *T = &d.T->s;
    const BigT *B = &d.T->b;
    const float2 *tw = d.tw;
    int twN = MAX_P;          // the twiddle table holds exp(-2 pi i m / twN), m < twN/2
    if (TAB_LDS) {            // persistent workgroup: hot energy tables and twiddles live in LDS, one copy per workgroup
        unsigned char *shared = lds_all + WPB * lay.per_team;
        int *dst = (int *)(shared + lay.off_tab);
        const int *src = (const int *)&d.T->s;
        for (int i = threadIdx.x; i < (int)(sizeof(SmallT) / 4); i += NT * WPB) dst[i] = src[i];
        T = (const SmallT *)dst;
        {   // (always, so that the compiler knows `tw` for an LDS pointer: a pointer that may be either makes every twiddle
            //  read a FLAT load, and a flat load waits for every global load in flight; the host keeps Pmax <= CLS2_P here)
            float2 *twl = (float2 *)(shared + lay.off_tw);
            if (!nofft) for (int m = threadIdx.x; m < Pmax / 2; m += NT * WPB) twl[m] = d.tw[m * (MAX_P / Pmax)];
            tw = twl;
            twN = Pmax;
        }
        __syncthreads();      // the only workgroup-wide barrier of the packed form
    }
    uint16_t *pos = (uint16_t *)(lds + lay.off_pos);
    uint8_t *code = lds + lay.off_code;
    uint32_t *P2 = (uint32_t *)(lds + lay.off_p2);      // the bases again, 2 bits per position (stem_stack_windows); only with CODE_LDS
    uint8_t *Sl_lds = lds + lay.off_S;
    uint32_t *brl = (uint32_t *)(lds + lay.off_br);
    uint16_t *rk = (uint16_t *)(lds + lay.off_rk);
    uint16_t *wnb = (uint16_t *)(lds + lay.off_nb);
    uint16_t *wmi = (uint16_t *)(lds + lay.off_mi);
    uint16_t *widx = (uint16_t *)(lds + lay.off_mj);    // the lags that gave a stem, compacted (the stem's mj is lag - mi: not stored)
    int *dd = (int *)(lds + lay.off_dd);
    uint16_t *keep = (uint16_t *)(lds + lay.off_keep);
    double *wtab = (double *)(lds + lay.off_w);
    int *misc = (int *)(lds + lay.off_misc);

    if (!nofft && tid < 25) {      // (a class without FFT buffers has no such table: expand_lds)
        int a = tid / 5, b = tid % 5;
        int tp = pair_type(a, b);
        wtab[tid] = (tp == 5 || tp == 6) ? d.au : (tp == 1 || tp == 2) ? d.gc : (tp == 3 || tp == 4) ? d.gu : 0.0;
    }
    // an arena overflowed in an earlier kernel of this wave: the host regrows and folds the wave again, whatever is queued behind
    // that kernel (the host issues a step ahead of its read-backs) finds records that were never written - and does nothing
    if (d.c->overflow) return;
    const unsigned n_items = d.c->n_work[cls].v;
    if (gteam == 0 && tid == 0) d.c->n_mat = 0;                // the beam step that follows counts its new structures here
    // (class 3 is served by two kernels on one work list: the list's length says which of them works - launch_expand_cls)
    if (((cls_arg & 0x4000) && n_items > (unsigned)d.c3_switch) || ((cls_arg & 0x8000) && n_items <= (unsigned)d.c3_switch)) return;
    const int shard = gteam & (NSHARD - 1);
    unsigned st_items = 0, st_n = 0, st_lags = 0, st_nbr = 0;   // per-team statistics (uniform over the team: scalar registers; a team's share of one launch fits 32 bits)
    if (tid < 3) misc[24 + tid] = 0;         // per team: stem energies evaluated / involving a rule or model value / kept ones that do

    // Work items are fetched FETCH at a time and candidate slots are reserved in slabs, so that the
    // two atomics with a returned value (a full L2 round trip each) are paid once per several regions.
    // (only when there is plenty of work: with fewer regions than workgroups every region gets its own)
    const bool eprof = prof_e != nullptr && tid == 0;      // diagnostic phase stamps (RAFFT_TRACE=3)
    unsigned long long eacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, et = eprof ? clock64() : 0, ft1 = 0;
#define ESTAMP(k) do { if (eprof) { const unsigned long long tn_ = clock64(); eacc[k] += tn_ - et; et = tn_; ft1 = tn_; } } while (0)
#define FSTAMP(k) do { if (eprof) { const unsigned long long tn_ = clock64(); eacc[k] += tn_ - ft1; ft1 = tn_; } } while (0)   // inside dE (10, 11) and emit (12-15)
    const unsigned FETCH = (NT == 64 && n_items > 4u * n_teams) ? (unsigned)d.fetch_bulk : 1u;
    const FetchPlan fplan = fetch_plan(d, n_items, NT == 64 ? FETCH : 1u, 1u);
    unsigned fetch_base = 0, fetch_left = 0;                 // uniform across the workgroup
    int fshard = (int)(gteam & (NSHARD - 1));                // work-cursor shard this team claims from next (fetch_chunk)
    unsigned long long ffailed = 0;
    unsigned long long slab_base = 0; unsigned slab_left = 0;   // reserved candidate slots (NT == 64: uniform over the wavefront; wider teams: thread 0 only)

    for (;;) {
        ESYNC();                       // previous region's LDS use is over
        if (prof_e != nullptr && (rep & 256)) {      // diagnostic: how long the previous region's stores take to drain
            const unsigned long long t0_ = clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) atomicAdd(&prof_e[cls * PROF_E + 32], (unsigned long long)(clock64() - t0_));
        }
        unsigned long long ft0 = eprof ? clock64() : 0;
        if (fetch_left == 0) {
            unsigned fcount = 1;
            if (NT == 64) fetch_base = fetch_chunk(d, cls, fplan, fshard, ffailed, fcount);
            else {
                if (tid < 64) { const unsigned b_ = fetch_chunk(d, cls, fplan, fshard, ffailed, fcount); if (tid == 0) misc[8] = (int)b_; }
                ESYNC();
                fetch_base = (unsigned)misc[8];
            }
            if (fetch_base == ~0u) break;
            fetch_left = fcount;
        }
        // (the work item is the same for the whole team: saying so - readfirstlane - turns the header loads below into scalar
        //  loads, off the vector memory queue and out of the vector registers; only for the one-wavefront class, where a
        //  team IS a wavefront)
        const unsigned item = NT == 64 ? (unsigned)__builtin_amdgcn_readfirstlane((int)fetch_base) : fetch_base;
        fetch_base++; fetch_left--;
        if (item >= n_items) { fetch_left = 0; continue; }       // (tail of the list's last chunk; other shards may still hold chunks)
        if (eprof) { const unsigned long long t_ = clock64(); eacc[8] += t_ - ft0; ft0 = t_; }
        const int nid = NT == 64 ? __builtin_amdgcn_readfirstlane(d.work[cls][item]) : d.work[cls][item];
        if (eprof) { const unsigned long long t_ = clock64(); eacc[9] += t_ - ft0 + (unsigned long long)(nid & 0); ft0 = t_; }
        const int L = d.nd[nid].L;                 // (the record carries its sequence's length and offset: no look-up keyed on `seq`)
        const int n = d.nd[nid].n, ci = d.nd[nid].ci, cj = d.nd[nid].cj, nbr = d.nd[nid].nbr;
        const int par_dcal = d.nd[nid].pdcal;
        const uint16_t *posg = d.pos + d.nd[nid].pos;
        const uint32_t *brg = d.br + d.nd[nid].br;
        const uint8_t *codes = d.codes + d.nd[nid].soff;
        auto code_at = [&](int t) -> int { return CODE_LDS ? (int)code[t] : (int)codes[pos[t]]; };
        // (LDS copy of the bases: only the loop's span [sx0, sx1) is staged, at Sl_lds[x - sx0]; the pointer is shifted so
        //  that it still takes sequence positions - sx0 < 4096 never exceeds the offset of that area, the shifted pointer stays
        //  inside the LDS.  The address space is known at compile time either way.)
        const int sx0 = ci < 0 ? 0 : ci, sx1 = ci < 0 ? L : cj + 1;
        // (the copy goes four bases at a time, whole aligned words of the sequence: the LDS copy starts `spad` bytes in, so that it
        //  is aligned like its source; the 8 bytes of slack in front of and behind the area hold the up to three bases too many)
        const int spad = LONGSEQ ? 0 : (int)((uintptr_t)(codes + sx0) & 3u);
        const uint8_t *Sl = LONGSEQ ? codes : (const uint8_t *)Sl_lds + spad - sx0;
        const int m = 2 * n - 1;
        const int P = next_pow2_ge(m);
        const int logP = 31 - __clz(P);
        const int Pk = LONGSEQ == 2 ? 0 : P;       // the lag values occupy 8 P bytes of region A - unless they live in HBM
        const int size_bk = n <= 8 ? 0 : n <= 16 ? 1 : n <= 32 ? 2 : n <= 64 ? 3 : n <= 128 ? 4 : 5;      // (diagnostic: regions and cycles by size)
        ESTAMP(0);   // fetch + header
        if (skip_lvl >= 7) continue;
        if (skip_lvl >= 6) { if (tid == 0 && n + ci + cj + nbr + L + par_dcal == -12345) d.c->overflow = 1; continue; }     // (header values consumed)
        const unsigned long long t_region0 = eprof ? clock64() : 0;
        const int Kp = d.K < m ? (d.K > 0 ? d.K : 0) : m;

        for (int rep_ = 0; rep_ < 1 + ((rep >> 4) & 1); rep_++) {
        // (every lane of the team walks the loop, so that the rows of 16 lanes that pack the bases - 2 bits each, SmallT::stk4 - are whole)
        for (int t0 = 0; t0 < n; t0 += NT) {
            const int t = t0 + tid;
            int c = 0;
            if (t < n) {
                const int p = posg[t];
                if (d.pos_packed) { pos[t] = (uint16_t)(p & 0x0FFF); c = p >> 12; code[t] = (uint8_t)c; }   // (Dev::pos_packed: no sequence beyond 4096 nt in this wave)
                else { pos[t] = (uint16_t)p; if (CODE_LDS) { c = codes[p]; code[t] = (uint8_t)c; } }
            }
            if (CODE_LDS) {
                const uint32_t x = row16_or((uint32_t)((c + 3) & 3) << (2 * (t & 15)));
                if ((t & 15) == 15 && t - 15 < n) P2[t >> 4] = x;
            }
        }
        if (CODE_LDS && tid == 0) P2[(n + 15) >> 4] = 0u;       // (the word of slack behind the last: strand_window reads two)
        if (LONGSEQ == 0) {   // bases: only the span of this loop is ever looked at (closing pair, its neighbours inside, branches)
            const uint32_t *src4 = (const uint32_t *)(codes + sx0 - spad);
            const int nw4 = (sx1 - sx0 + spad + 3) >> 2;
            for (int x = tid; x < nw4; x += NT) ((uint32_t *)Sl_lds)[x] = src4[x];
        }
        for (int t = tid; t < nbr; t += NT) brl[t] = d.pos_packed ? (brg[t] & 0x0FFF0FFFu) : brg[t];   // (Dev::pos_packed: the codes ride along)
        ESYNC();
        }

        ESTAMP(1);   // LDS fill
        if (skip_lvl >= 5) continue;
        // ---- correlation: conv(A,U), conv(G,C), conv(G,U).
        // Regions of <= 64 positions (one wavefront holds the whole strand in 64-bit masks) use the exact
        // direct form: popcount(mask & shifted reversed mask) per lag - the analogue of scipy's own
        // method="auto" picking direct convolution for short inputs (rafft/utils.py:121).  Longer regions
        // go through two packed complex FFTs in LDS.  Both give the same exact integer pair counts.
        const bool direct = (NT == 64) && n <= 64 && !force_fft;
        float2 *z1 = (float2 *)(lds + lay.offA);
        float2 *z2 = z1 + P;
        // The wide classes correlate regions of up to Dev::direct_n positions by the exact direct form on multi-word bit masks -
        // what the class for regions beyond 4096 positions always does - and longer ones by the FFT (rafft/utils.py:115-122:
        // scipy's convolve makes the same kind of choice); same integer pair counts either way.
        const bool mw = !direct && LONGSEQ != 2 && (nofft || (n <= d.direct_n && P >= 128 && dbg.lag == nullptr && !force_fft &&
                        d.gc >= 0.0 && d.au >= 0.0 && d.gu >= 0.0));
        if (!direct && !mw && LONGSEQ != 2)
        for (int rep_ = 0; rep_ < 1 + (rep & 1); rep_++) {   // rep: profiling-only phase doubling
            for (int t = tid; t < P; t += NT) {
                int c = t < n ? code[t] : 0;
                z1[t] = make_float2(c == 1 ? 1.f : 0.f, c == 3 ? 1.f : 0.f); // A + iG
                z2[t] = make_float2(c == 4 ? 1.f : 0.f, c == 2 ? 1.f : 0.f); // U + iC
            }
            ESYNC();
            // DIF, natural in -> bit-reversed out.  Two radix-2 stages (spans s and s/2) are done per pass on four
            // elements held in registers: the same operations in the same order as stage by stage (bit-identical
            // results), half the LDS traffic and barriers.
            auto add2 = [](float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); };
            auto sub2 = [](float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); };
            int s = P >> 1;
            for (; s >= 2; s >>= 2) {
                const int h = s >> 1, tws = (twN / 2) / s;
                for (int b = tid; b < (P >> 2); b += NT) {
                    const int off = b & (h - 1);
                    const int j = ((b - off) << 2) + off;          // j mod 2s < s/2
                    const float2 w1a = tw[off * tws], w1b = tw[(off + h) * tws], w2 = tw[off * 2 * tws];
                    {
                        const float2 x0 = z1[j], x1 = z1[j + h], x2 = z1[j + s], x3 = z1[j + s + h];
                        const float2 a0 = add2(x0, x2), a2 = cmul(sub2(x0, x2), w1a), a1 = add2(x1, x3), a3 = cmul(sub2(x1, x3), w1b);
                        z1[j] = add2(a0, a1); z1[j + h] = cmul(sub2(a0, a1), w2);
                        z1[j + s] = add2(a2, a3); z1[j + s + h] = cmul(sub2(a2, a3), w2);
                    }
                    {
                        const float2 x0 = z2[j], x1 = z2[j + h], x2 = z2[j + s], x3 = z2[j + s + h];
                        const float2 a0 = add2(x0, x2), a2 = cmul(sub2(x0, x2), w1a), a1 = add2(x1, x3), a3 = cmul(sub2(x1, x3), w1b);
                        z2[j] = add2(a0, a1); z2[j + h] = cmul(sub2(a0, a1), w2);
                        z2[j + s] = add2(a2, a3); z2[j + s + h] = cmul(sub2(a2, a3), w2);
                    }
                }
                ESYNC();
            }
            if (s == 1) {                                           // odd number of stages: the last one alone
                for (int b = tid; b < (P >> 1); b += NT) {
                    const int j = b << 1;
                    float2 a = z1[j], bb = z1[j + 1];
                    z1[j] = add2(a, bb); z1[j + 1] = cmul(sub2(a, bb), tw[0]);
                    a = z2[j]; bb = z2[j + 1];
                    z2[j] = add2(a, bb); z2[j + 1] = cmul(sub2(a, bb), tw[0]);
                }
                ESYNC();
            }
            // separate the packed real spectra, multiply.  The spectra sit in bit-reversed order: walking k = 0, 1, 2 ... would
            // send the 64 lanes of a wavefront to addresses P/2, P/4 ... apart - one LDS bank for all of them.  So the walk is
            // over the POSITIONS: the even ones hold exactly the k < P/2 (top bit of k = lowest bit of the position), position 1
            // holds k = P/2; neighbours in the walk are neighbours in LDS, and the mirror position of -k runs the other way.
            for (int t = tid; t <= (P >> 1); t += NT) {
                const int jk = t == (P >> 1) ? 1 : 2 * t;
                const int k = (int)(__brev((unsigned)jk) >> (32 - logP));
                const int km = (P - k) & (P - 1);
                const int jm = (int)(__brev((unsigned)km) >> (32 - logP));
                float2 A1 = z1[jk], B1 = z1[jm], A2 = z2[jk], B2 = z2[jm];
                float2 Fa = make_float2(0.5f * (A1.x + B1.x), 0.5f * (A1.y - B1.y));
                float2 Fg = make_float2(0.5f * (A1.y + B1.y), -0.5f * (A1.x - B1.x));
                float2 Fu = make_float2(0.5f * (A2.x + B2.x), 0.5f * (A2.y - B2.y));
                float2 Fc = make_float2(0.5f * (A2.y + B2.y), -0.5f * (A2.x - B2.x));
                float2 X = cmul(Fa, Fu), Y = cmul(Fg, Fc), Z = cmul(Fg, Fu);
                z1[jk] = make_float2(X.x - Y.y, X.y + Y.x);
                z2[jk] = Z;
                if (jm != jk) {
                    z1[jm] = make_float2(X.x + Y.y, Y.x - X.y);
                    z2[jm] = make_float2(Z.x, -Z.y);
                }
            }
            ESYNC();
            // DIT inverse, bit-reversed in -> natural out; again two stages (spans s and 2s) per pass
            int si = 1;
            if (logP & 1) {                                         // odd number of stages: the first one alone
                for (int b = tid; b < (P >> 1); b += NT) {
                    const int j = b << 1;
                    float2 a = z1[j], bb = cmulc(z1[j + 1], tw[0]);
                    z1[j] = add2(a, bb); z1[j + 1] = sub2(a, bb);
                    a = z2[j]; bb = cmulc(z2[j + 1], tw[0]);
                    z2[j] = add2(a, bb); z2[j + 1] = sub2(a, bb);
                }
                ESYNC();
                si = 2;
            }
            for (; si < P; si <<= 2) {
                const int s1 = si, s2 = si << 1, tws = (twN / 2) / s1;
                for (int b = tid; b < (P >> 2); b += NT) {
                    const int off = b & (s1 - 1);
                    const int j = ((b - off) << 2) + off;          // j mod 4 s1 < s1
                    const float2 w1 = tw[off * tws], w2a = tw[off * (tws >> 1)], w2b = tw[(off + s1) * (tws >> 1)];
                    {
                        const float2 x0 = z1[j], x2 = z1[j + s2];
                        const float2 t1 = cmulc(z1[j + s1], w1), t3 = cmulc(z1[j + s2 + s1], w1);
                        const float2 y0 = add2(x0, t1), y1 = sub2(x0, t1), y2 = add2(x2, t3), y3 = sub2(x2, t3);
                        const float2 u2 = cmulc(y2, w2a), u3 = cmulc(y3, w2b);
                        z1[j] = add2(y0, u2); z1[j + s2] = sub2(y0, u2); z1[j + s1] = add2(y1, u3); z1[j + s2 + s1] = sub2(y1, u3);
                    }
                    {
                        const float2 x0 = z2[j], x2 = z2[j + s2];
                        const float2 t1 = cmulc(z2[j + s1], w1), t3 = cmulc(z2[j + s2 + s1], w1);
                        const float2 y0 = add2(x0, t1), y1 = sub2(x0, t1), y2 = add2(x2, t3), y3 = sub2(x2, t3);
                        const float2 u2 = cmulc(y2, w2a), u3 = cmulc(y3, w2b);
                        z2[j] = add2(y0, u2); z2[j + s2] = sub2(y0, u2); z2[j + s1] = add2(y1, u3); z2[j + s2 + s1] = sub2(y1, u3);
                    }
                }
                ESYNC();
            }
        }

        ESTAMP(2);   // FFTs
        if (skip_lvl >= 4) continue;
        // ---- lag values (exact integer pair counts, IEEE fp64 divide) and ranking
        // Which lags are searched (rafft/rafft.py:117-118 takes the nb_mode best by (value desc, lag desc)):
        //  - all of them when 2n-1 <= nb_mode: nothing to rank;
        //  - otherwise the best nb_mode are SELECTED exactly (byte-wise radix select on the order-preserving bit
        //    pattern of the fp64 value, ties: larger lag first) - their order is not needed, because the only
        //    place it shows is the stable dE sort of the candidates, and that breaks ties from (value, lag) itself;
        //  - tiny FFT sizes (P <= 128) and the debug seam, which reports the ranking, sort all keys in place.
        const bool dbgrank = dbg.lag != nullptr;
        const bool ranked = m > Kp;
        const bool selected = ranked && P >= 128;
        const bool inplace = (ranked && !selected) || (dbgrank && !selected);    // keys sorted in place, rk[] in rank order
        double *keyv = LONGSEQ == 2 ? d.big_keyv + (size_t)gteam * d.big_stride : (double *)(lds + lay.offA);
        uint16_t *lagk = LONGSEQ == 2 ? (uint16_t *)(keyv + P) : (uint16_t *)(lds + lay.offA + 8 * P);
        // (round 5, production builds - weights >= 0) The top byte of the order-preserving key of a lag value - sign and the upper seven
        // bits of the exponent - only says whether the value is 0, below 2 or at least 2: counted here with three ballots per 64 lags
        // (wavefront-uniform counters: scalar registers), which is the radix select's first pass without a pass over the keys - for the
        // class whose lag values live in HBM one read of them less.  Values outside [2^-15, 2^17) (user weights of another scale) or a
        // negative one: `c_odd`, and the select starts at the top byte as before.
        int c_hi = 0, c_lo = 0, c_odd = 0;
        auto tally = [&](double v_, bool valid) {
            if (PROD && selected) {
                c_hi += __popcll(__ballot(valid && v_ >= 2.0));
                c_lo += __popcll(__ballot(valid && v_ > 0.0 && v_ < 2.0));
                c_odd |= __ballot(valid && (v_ >= 131072.0 || v_ < 0.0 || (v_ > 0.0 && v_ < 0x1p-15))) != 0ULL ? 1 : 0;
            }
        };
        if (LONGSEQ == 2 || mw) {
            // base masks of the region (the same arrays window_slide uses below, built once here) ...
            const int W = (n + 63) >> 6;
            unsigned long long *F = (unsigned long long *)(lds + lay.offA + 8 * Pk);
            unsigned long long *R = F + MASK_F_WORDS * W;
            build_masks<NT>(F, R, W, n, code_at, pos, tid);
            ESYNC();
            // ... and the three pair counts of every lag: bit ip of window(R_x, sft + 64 w) = base x at position k - ip
            for (int k = tid; k < P; k += NT) {
                double v = -INFINITY;
                if (k < m) {
                    // Only the words that hold cells of this diagonal - positions ip with 0 <= k - ip < n - are visited (half of them on
                    // average: the lags near either end have short diagonals), and the 64-bit window of the reversed masks slides: every
                    // step loads ONE new word per mask and reuses the high word of the step before (mask_window would load two and
                    // range-check both).  Same bits, same counts.
                    const int sft = n - 1 - k;
                    const int ip_lo = k > n - 1 ? k - (n - 1) : 0, ip_hi = k < n - 1 ? k : n - 1;
                    const int w0 = ip_lo >> 6, w1 = ip_hi >> 6;
                    const int start = (w0 << 6) + sft;                 // first bit of the window of word w0 (negative: bits before the string are zeros)
                    int q = start >> 6;                                 // (arithmetic shift: floor)
                    const int bsh = start & 63;
                    const unsigned long long *RU = R + 3 * W, *RC = R + 1 * W;
                    unsigned long long loU = (q >= 0 && q < W) ? RU[q] : 0ULL, loC = (q >= 0 && q < W) ? RC[q] : 0ULL;
                    int cAU = 0, cGC = 0, cGU = 0;
                    for (int w = w0; w <= w1; w++, q++) {
                        const bool in = q + 1 >= 0 && q + 1 < W;
                        const unsigned long long hiU = in ? RU[q + 1] : 0ULL, hiC = in ? RC[q + 1] : 0ULL;
                        const unsigned long long xU = bsh ? (loU >> bsh) | (hiU << (64 - bsh)) : loU, xC = bsh ? (loC >> bsh) | (hiC << (64 - bsh)) : loC;
                        const unsigned long long fA = F[0 * W + w], fG = F[2 * W + w];
                        cAU += __popcll(fA & xU); cGC += __popcll(fG & xC); cGU += __popcll(fG & xU);
                        loU = hiU; loC = hiC;
                    }
                    const double raw = (2.0 * (double)cAU) * d.au + (2.0 * (double)cGC) * d.gc + (2.0 * (double)cGU) * d.gu;
                    const int nk = k < m - 1 - k ? k : m - 1 - k;
                    v = raw / ((double)nk + 1.0);
                }
                keyv[k] = v;
                tally(v, k < m);
            }
            ESYNC();
            if (inplace) {                                  // lag column of the in-place sort (tiny regions in a class without FFT buffers;
                for (int k = tid; k < P; k += NT) lagk[k] = (uint16_t)k;      //  it takes the place of the masks, rebuilt for window_slide)
                ESYNC();
            }
        } else if (direct) {
            const int c = tid < n ? code[tid] : 0;
            const unsigned long long mA = __ballot(c == 1), mC = __ballot(c == 2), mG = __ballot(c == 3), mU = __ballot(c == 4);
            const unsigned long long rU = __brevll(mU) >> (64 - n), rC = __brevll(mC) >> (64 - n);   // strand reversed
            for (int rep_ = 0; rep_ < 1 + (rep & 1) + ((rep >> 5) & 1); rep_++)
            for (int k = tid; k < P; k += NT) {
                double v = -INFINITY;
                if (k < m) {
                    const int sft = n - 1 - k;                     // bit i of x* = base at position k - i
                    const unsigned long long xU = sft >= 0 ? (rU >> sft) : (rU << -sft);
                    const unsigned long long xC = sft >= 0 ? (rC >> sft) : (rC << -sft);
                    double nAU = 2.0 * (double)__popcll(mA & xU);
                    double nGC = 2.0 * (double)__popcll(mG & xC);
                    double nGU = 2.0 * (double)__popcll(mG & xU);
                    double raw = nAU * d.au + nGC * d.gc + nGU * d.gu;
                    int nk = k < m - 1 - k ? k : m - 1 - k;
                    v = raw / ((double)nk + 1.0);
                }
                keyv[k] = v;
                tally(v, k < m);
            }
            ESYNC();
            if (inplace) {                                  // lag column of the in-place sort
                for (int k = tid; k < P; k += NT) lagk[k] = (uint16_t)k;
                ESYNC();
            }
        } else {
            // keyv[k] aliases z1[k] byte for byte and is written by the thread that read it;
            // lagk aliases the head of z2, so it is filled only after every read of z2.
            const float invP = 1.0f / (float)P;
            for (int k = tid; k < P; k += NT) {
                double v = -INFINITY;
                if (k < m) {
                    double nAU = 2.0 * (double)rintf(z1[k].x * invP);
                    double nGC = 2.0 * (double)rintf(z1[k].y * invP);
                    double nGU = 2.0 * (double)rintf(z2[k].x * invP);
                    double raw = nAU * d.au + nGC * d.gc + nGU * d.gu;
                    int nk = k < m - 1 - k ? k : m - 1 - k;
                    v = raw / ((double)nk + 1.0);
                }
                keyv[k] = v;
                tally(v, k < m);
            }
            ESYNC();
            if (inplace) {                                  // lag column of the in-place sort
                for (int k = tid; k < P; k += NT) lagk[k] = (uint16_t)k;
                ESYNC();
            }
        }
        ESTAMP(3);   // lag values
        if (skip_lvl >= 3) continue;
        if (selected) {
            int *hist = (int *)(lds + lay.offA + (LONGSEQ == 2 ? lay.szA - 2048 : nofft ? 8 * P + 8 * MASK_WORDS * ((nmax + 63) >> 6) : 9 * P));      // 256 bins behind the lag values and the bit masks (8 P + 0.69 P at most); at region A's end when the masks of the biggest regions are already there
            int *shs = hist + 256;                                   // scan scratch [32]
            for (int rep_ = 0; rep_ < 1 + ((rep >> 1) & 1); rep_++) {
            auto ukey = [&](int i) -> unsigned long long {
                unsigned long long u = (unsigned long long)__double_as_longlong(keyv[i]);
                return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
            };
            unsigned long long prefix = 0;
            int kk = Kp;
            bool take_ge = false;          // every key >= prefix is selected (the threshold fell between two values)
            int pass0 = 7;
            if (PROD) {
                if (NT > 64) {             // (the counters are per wavefront: summed over the team)
                    if (tid < 3) shs[20 + tid] = 0;
                    ESYNC();
                    if ((tid & 63) == 0) { atomicAdd(&shs[20], c_hi); atomicAdd(&shs[21], c_lo); atomicOr(&shs[22], c_odd); }
                    ESYNC();
                    c_hi = shs[20]; c_lo = shs[21]; c_odd = shs[22];
                }
                if (!c_odd) {              // byte 7 of the keys: 0xC0 for [2, 2^17), 0xBF for [2^-15, 2), 0x80 for 0
                    const int c_zero = m - c_hi - c_lo;
                    int binc;
                    if (kk <= c_hi) { prefix = 0xC0ULL << 56; binc = c_hi; }
                    else if (kk <= c_hi + c_lo) { prefix = 0xBFULL << 56; kk -= c_hi; binc = c_lo; }
                    else { prefix = 0x80ULL << 56; kk -= c_hi + c_lo; binc = c_zero; }
                    pass0 = kk == binc ? -1 : 6;       // (the whole bin is wanted: nothing below that byte needs looking at)
                    take_ge = kk == binc;
                }
            }
            for (int pass = pass0; pass >= 0; pass--) {
                for (int i = tid; i < 256; i += NT) hist[i] = 0;
                ESYNC();
                const int sh_hi = 8 * (pass + 1);
                for (int i = tid; i < m; i += NT) {
                    const unsigned long long u = ukey(i);
                    if (pass == 7 || (u >> sh_hi) == (prefix >> sh_hi)) atomicAdd(&hist[(int)((u >> (8 * pass)) & 255ULL)], 1);
                }
                ESYNC();
                // largest byte b with count(bytes > b) < kk <= count(bytes >= b): suffix scan over the bins
                {
                    constexpr int BPT = NT >= 256 ? 1 : 256 / NT;      // bins per thread, from the top bin down
                    int hs[BPT], mine = 0;
#pragma unroll
                    for (int j = 0; j < BPT; j++) { const int bi = tid * BPT + j; hs[j] = bi < 256 ? hist[255 - bi] : 0; mine += hs[j]; }
                    int tot, ex = block_exscan<NT>(mine, shs, &tot);
#pragma unroll
                    for (int j = 0; j < BPT; j++) {
                        if (ex < kk && kk <= ex + hs[j] && hs[j] > 0) { shs[28] = 255 - (tid * BPT + j); shs[29] = kk - ex; shs[30] = hs[j]; }
                        ex += hs[j];
                    }
                    ESYNC();
                }
                prefix |= (unsigned long long)(unsigned)shs[28] << (8 * pass);
                kk = shs[29];
                const bool whole_bin = kk == shs[30];      // all keys of the threshold bin are wanted: no need to look
                ESYNC();                           // at the lower bytes (the usual case after two or three passes)
                if (whole_bin) { take_ge = true; break; }
            }
            // take every lag with key > prefix and the kk largest lags among key == prefix (sweep from the top)
            int outn = 0, tie_run = 0;
            for (int base = 0; base < P; base += NT) {
                const int i = P - 1 - (base + tid);
                unsigned long long u = 0;
                int tie = 0;
                if (i >= 0 && i < m) { u = ukey(i); tie = (u == prefix) ? 1 : 0; }
                int ttot = 0, tex = 0;
                if (!take_ge) tex = block_exscan_flag<NT>(tie, shs, &ttot);     // (the order among ties only matters when the cut falls inside them)
                const int g = (i >= 0 && i < m) && (take_ge ? u >= prefix : (u > prefix || (tie && tie_run + tex < kk))) ? 1 : 0;
                int gtot, gex = block_exscan_flag<NT>(g, shs, &gtot);
                if (g) rk[outn + gex] = (uint16_t)i;
                outn += gtot; tie_run += ttot;
                ESYNC();
            }
            // the debug seam reports the ranking: sort the selected lags by (value desc, lag desc)
            if (dbgrank) {
            int M2 = 2; while (M2 < Kp) M2 <<= 1;
            for (int i = Kp + tid; i < M2; i += NT) rk[i] = 0xFFFF;
            ESYNC();
            for (int k2 = 2; k2 <= M2; k2 <<= 1)
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < M2; i += NT) {
                        int ixj = i ^ j;
                        if (ixj > i) {
                            const uint16_t la = rk[i], lb = rk[ixj];
                            bool a_first;
                            if (la == 0xFFFF) a_first = false;
                            else if (lb == 0xFFFF) a_first = true;
                            else { const double va = keyv[la], vb = keyv[lb]; a_first = (va > vb) || (va == vb && la > lb); }
                            const bool up = (i & k2) == 0;
                            if (up ? !a_first : a_first) { rk[i] = lb; rk[ixj] = la; }
                        }
                    }
                    ESYNC();
                }
            }
            }
            for (int r = tid; r < Kp; r += NT)
                if (dbg.lag) { dbg.lag[r] = rk[r]; dbg.corval[r] = keyv[rk[r]]; }
            if (tid == 0 && dbg.n_ranked) *dbg.n_ranked = Kp;
            ESYNC();
        } else {
        if (inplace)
        for (int rep_ = 0; rep_ < 1 + ((rep >> 1) & 1); rep_++)
            for (int k2 = 2; k2 <= P; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += NT) {
                        int ixj = i ^ j;
                        if (ixj > i) {
                            double va = keyv[i], vb = keyv[ixj];
                            uint16_t la = lagk[i], lb = lagk[ixj];
                            bool a_first = (va > vb) || (va == vb && la > lb);
                            bool up = (i & k2) == 0;
                            if (up ? !a_first : a_first) {
                                keyv[i] = vb; keyv[ixj] = va;
                                lagk[i] = lb; lagk[ixj] = la;
                            }
                        }
                    }
                    ESYNC();
                }
            }
        for (int r = tid; r < Kp; r += NT) {
            rk[r] = inplace ? lagk[r] : (uint16_t)r;
            if (dbg.lag) { dbg.lag[r] = lagk[r]; dbg.corval[r] = keyv[r]; }   // (debug seam always sorts)
        }
        if (tid == 0 && dbg.n_ranked) *dbg.n_ranked = Kp;
        ESYNC();
        }

        ESTAMP(4);   // ranking
        if (skip_lvl >= 2) continue;
        // ---- window_slide (rafft/rafft.py:36-83).  Small regions: one lane per ranked lag.  Big regions:
        // each diagonal is cut into C chunks handled by different lanes; a lane first walks back to the last
        // zero cell before its chunk and replays the recurrence from there (same fp64 operation order, so
        // values are bit-identical), then the chunk results are merged with the reference's `>=` rule.
        // (chunking only for regions ranked by selection; the partial results go behind the lag values)
        const int C = (NT >= 256 && selected) ? max(1, min(8, NT / max(Kp, 1))) : 1;
        struct WsPart { double score; int nb, mi, mj, any; };
        WsPart *parts = (WsPart *)(lds + lay.offA + 8 * Pk);      // big regions only: behind the lag values (and the masks)
        // The diagonal of a lag as bit masks: pairing cells per pair type (base masks AND shifted reversed base
        // masks, 64 cells per word), contiguity with the previous cell as a mask too.  Only the pairing cells
        // are visited - zero cells never change the result: same fp64 recurrence on the visited cells in the
        // same order, same `>=` rule.  A chunk first walks back over the run of pairing cells that ends just
        // before it and replays the recurrence over that run (zero cells reset it, so nothing older matters).
        // (negative weights or the forced-FFT test mode take the cell-by-cell form below)
        const bool ws_masks = PROD || (d.gc >= 0.0 && d.au >= 0.0 && d.gu >= 0.0 && !force_fft);
        if (ws_masks) {
            // forward masks F[0..3] = A,C,G,U, F[4] = contiguity with the previous position; R[] = reversed strings (build_masks).
            // Region A: behind the lag values (8 P bytes) unless those were sorted in place and are dead; the
            // partial results of chunked diagonals follow the masks.
            const int W = (n + 63) >> 6;
            unsigned long long *F = (unsigned long long *)(lds + lay.offA + (inplace ? 0 : 8 * Pk));
            unsigned long long *R = F + MASK_F_WORDS * W;
            parts = (WsPart *)(R + 5 * W);
            if (LONGSEQ != 2 && (!mw || inplace))   // (the direct correlation on multi-word masks has built them already - behind the lag values)
            for (int rep_ = 0; rep_ < 1 + ((rep >> 7) & 1); rep_++) {
                build_masks<NT>(F, R, W, n, code_at, pos, tid);
                ESYNC();
            }
            // (round 5) The cells of a diagonal are taken 32 at a time, counted from the diagonal's FIRST cell: chunk k holds the cells
            // ip0 + 32 k .. of the forward strings and - the reversed strings all being read at bit n - 1 - lag + ip - the bits
            // n - 1 - lag + ip0 + 32 k .. of the reversed ones; a window of 32 bits at any bit offset is two adjacent words and one
            // v_alignbit.  Half a diagonal of a region of up to 64 positions is ONE chunk, and the loop over its pairing cells runs on
            // 32-bit masks (rounds 1-4: 64-bit words aligned to the region, 64-bit shifts and tests per cell, windows assembled from
            // range-checked loads: 38 % of the kernel's vector instructions, tools/pmc_phases.sh).  Bits read past a string's end are
            // cells past the half-diagonal's eligible prefix: masked.
            const uint32_t *F32 = (const uint32_t *)F, *R32 = (const uint32_t *)R;
            const int W2 = 2 * W;
            // (the three pair weights in vector registers of their own: a select between two scalar operands is not encodable, and
            //  the compiler would rather copy them into vector registers again for every cell of the loop below - 6 of its 42 instructions)
            double wgc = d.gc, wau = d.au, wgu = d.gu;
            asm volatile("" : "+v"(wgc), "+v"(wau), "+v"(wgu));
            auto win = [](const uint32_t *X, int start) -> uint32_t { const int q_ = start >> 5; return __builtin_amdgcn_alignbit(X[q_ + 1], X[q_], (uint32_t)(start & 31)); };
            for (int rep_ = 0; rep_ < 1 + ((rep >> 2) & 1); rep_++) {
            for (int q = tid; q < Kp * C; q += NT) {
                const int r = q / C, c = q - r * C;
                const int lagp = rk[r];
                const int len = lagp < n ? lagp + 1 : 2 * n - lagp - 1;
                const int len2 = (len >> 1) + (len & 1);
                const int ip0 = lagp < n ? 0 : lagp - n + 1, jp0 = lagp < n ? lagp : n - 1;
                // eligible cells (pos[jp]-pos[ip] > min_hp) form a prefix.  Positions are strictly increasing, so pos[jp] - pos[ip] >=
                // jp - ip = len - 1 - 2 i: every cell with len - 1 - 2 i > min_hp is eligible without looking, and the search only
                // covers the (min_hp + 3) / 2 cells that remain at the inner end of the half-diagonal (two steps for min_hp = 3
                // where the search over all of it took log2(len / 2) dependent pairs of LDS reads)
                const int csure = len - 1 - d.min_hp;
                int lo = csure > 0 ? min((csure + 1) >> 1, len2) : 0, hi = len2;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)pos[jp0 - mid] - (int)pos[ip0 + mid] > d.min_hp) lo = mid + 1; else hi = mid;
                }
                const int lim = lo;
                // this lane's share of the eligible cells [ca, ce), counted from the diagonal's first cell
                const int ca = (int)((long long)lim * c / C), ce = (int)((long long)lim * (c + 1) / C);
                const int rs0 = n - 1 - lagp + ip0;          // bit of the reversed strings that belongs to the first cell (>= 0)
                // pairing cells of chunk k by pair type, and the cells contiguous with their predecessor
                auto cells = [&](int k, uint32_t &pGC, uint32_t &pAU, uint32_t &pGU, uint32_t &cm) {
                    const int cs = ip0 + 32 * k, rs = rs0 + 32 * k;
                    const uint32_t fA = win(F32 + 0 * W2, cs), fC = win(F32 + 1 * W2, cs), fG = win(F32 + 2 * W2, cs), fU = win(F32 + 3 * W2, cs);
                    const uint32_t xA = win(R32 + 0 * W2, rs), xC = win(R32 + 1 * W2, rs), xG = win(R32 + 2 * W2, rs), xU = win(R32 + 3 * W2, rs);
                    pGC = d.gc != 0.0 ? ((fG & xC) | (fC & xG)) : 0u;
                    pAU = d.au != 0.0 ? ((fA & xU) | (fU & xA)) : 0u;
                    pGU = d.gu != 0.0 ? ((fG & xU) | (fU & xG)) : 0u;
                    cm = win(F32 + 4 * W2, cs) & win(R32 + 4 * W2, rs);       // contiguous with the previous cell on both strands
                    if (k == 0) cm &= ~1u;                                    // never for the first cell
                };
                auto span = [](int lo_, int hi_, int cb) -> uint32_t {                       // bits of cells [lo_, hi_) inside the chunk that starts at cell cb
                    uint32_t m_ = ~0u;
                    if (lo_ > cb) m_ &= ~0u << (lo_ - cb);
                    if (hi_ < cb + 32) m_ &= (1u << (hi_ - cb)) - 1u;
                    return m_;
                };
                double mx_s = 0.0, prev = 0.0;
                int mx_nb = 0, mx_c = 0, last_c = -2, runlen = 0, mx_i = 0, mx_j = 0;
                if (ce > ca) {
                    int z = ca;                              // replay start: first cell of the run of pairing cells ending at ca - 1
                    if (ca > 0) {
                        for (int kz = (ca - 1) >> 5;; kz--) {
                            const int cb = kz << 5;
                            uint32_t pGC, pAU, pGU, cm;
                            cells(kz, pGC, pAU, pGU, cm);
                            const uint32_t zeros = ~(pGC | pAU | pGU) & span(0, ca, cb);
                            if (zeros) { z = cb + 32 - __clz((int)zeros); break; }
                            if (cb == 0) { z = 0; break; }
                        }
                    }
                    for (int k = z >> 5; k <= (ce - 1) >> 5; k++) {
                        const int cb = k << 5;
                        uint32_t pGC, pAU, pGU, cm;
                        cells(k, pGC, pAU, pGU, cm);
                        uint32_t any = (pGC | pAU | pGU) & span(z, ce, cb);
                        while (any) {
                            const int bi = __ffs((int)any) - 1;
                            any &= any - 1u;
                            const int cc = cb + bi;
                            const double w8 = ((pGC >> bi) & 1u) ? wgc : ((pAU >> bi) & 1u) ? wau : wgu;
                            if (cc != last_c + 1) { prev = 0.0; runlen = 0; }   // previous cell was a zero cell
                            double t = w8;
                            if ((cm >> bi) & 1u) t = (prev + w8) * w8;
                            runlen++;
                            if ((C == 1 || cc >= ca) && t >= mx_s) { mx_s = t; mx_nb = runlen; mx_c = cc; }
                            prev = t; last_c = cc;
                        }
                    }
                    if (mx_nb == 0) mx_c = ce - 1;           // no pairing cell in the share: its last eligible (zero) cell, nb = 0
                    mx_i = ip0 + mx_c; mx_j = lagp - mx_i;
                }
                if (C == 1) {
                    wnb[r] = (uint16_t)mx_nb; wmi[r] = (uint16_t)mx_i;      // (mj = lag - mi)
                    if (dbg.nb) { dbg.nb[r] = mx_nb; dbg.mi[r] = mx_i; dbg.mj[r] = mx_j; dbg.score[r] = mx_s; }
                } else {
                    WsPart wp; wp.score = mx_s; wp.nb = mx_nb; wp.mi = mx_i; wp.mj = mx_j; wp.any = ce > ca ? 1 : 0;
                    parts[q] = wp;
                }
            }
            if (C > 1) {
                ESYNC();
                for (int r = tid; r < Kp; r += NT) {
                    double mx_s = 0.0;
                    int mx_nb = 0, mx_i = 0, mx_j = 0;
                    for (int c = 0; c < C; c++) {
                        const WsPart wp = parts[r * C + c];
                        if (wp.any && wp.score >= mx_s) { mx_s = wp.score; mx_nb = wp.nb; mx_i = wp.mi; mx_j = wp.mj; }
                    }
                    wnb[r] = (uint16_t)mx_nb; wmi[r] = (uint16_t)mx_i;      // (mj = lag - mi)
                    if (dbg.nb) { dbg.nb[r] = mx_nb; dbg.mi[r] = mx_i; dbg.mj[r] = mx_j; dbg.score[r] = mx_s; }
                }
            }
            }
        } else
        for (int rep_ = 0; rep_ < 1 + ((rep >> 2) & 1); rep_++) {
            for (int q = tid; q < Kp * C; q += NT) {
                const int r = q / C, c = q - r * C;
                const int lagp = rk[r];
                const int len = lagp < n ? lagp + 1 : 2 * n - lagp - 1;
                const int len2 = (len >> 1) + (len & 1);
                const int a = (int)((long long)len2 * c / C), e = (int)((long long)len2 * (c + 1) / C);
                const int ip0 = lagp < n ? 0 : lagp - n + 1, jp0 = lagp < n ? lagp : n - 1;   // cell i: (ip0+i, jp0-i)
                int z = a;                                  // replay start: just after the last zero cell before `a`
                while (z > 0 && wtab[code_at(ip0 + z - 1) * 5 + code_at(jp0 - (z - 1))] != 0.0) z--;
                double prev = 0.0, mx_s = 0.0;
                int tmp = 0, mx_nb = 0, mx_i = 0, mx_j = 0, any = 0;
                for (int i = z; i < e; i++) {
                    const int ip = ip0 + i, jp = jp0 - i;
                    double t = wtab[code_at(ip) * 5 + code_at(jp)];
                    if (i > 0 && (int)pos[ip] - (int)pos[ip - 1] == 1 && (int)pos[jp + 1] - (int)pos[jp] == 1)
                        t = (prev + t) * t;
                    tmp = (t == 0.0) ? 0 : tmp + 1;
                    if (i >= a && t >= mx_s && (int)pos[jp] - (int)pos[ip] > d.min_hp) {
                        mx_s = t; mx_nb = tmp; mx_i = ip; mx_j = jp; any = 1;
                    }
                    prev = t;
                }
                if (C == 1) {
                    wnb[r] = (uint16_t)mx_nb; wmi[r] = (uint16_t)mx_i;      // (mj = lag - mi)
                    if (dbg.nb) { dbg.nb[r] = mx_nb; dbg.mi[r] = mx_i; dbg.mj[r] = mx_j; dbg.score[r] = mx_s; }
                } else {
                    WsPart w; w.score = mx_s; w.nb = mx_nb; w.mi = mx_i; w.mj = mx_j; w.any = any;
                    parts[q] = w;
                }
            }
            if (C > 1) {
                ESYNC();
                for (int r = tid; r < Kp; r += NT) {
                    double mx_s = 0.0;
                    int mx_nb = 0, mx_i = 0, mx_j = 0;
                    for (int c = 0; c < C; c++) {
                        const WsPart w = parts[r * C + c];
                        if (w.any && w.score >= mx_s) { mx_s = w.score; mx_nb = w.nb; mx_i = w.mi; mx_j = w.mj; }
                    }
                    wnb[r] = (uint16_t)mx_nb; wmi[r] = (uint16_t)mx_i;      // (mj = lag - mi)
                    if (dbg.nb) { dbg.nb[r] = mx_nb; dbg.mi[r] = mx_i; dbg.mj[r] = mx_j; dbg.score[r] = mx_s; }
                }
            }
        }
        ESYNC();

        ESTAMP(5);   // window_slide
        if (skip_lvl >= 1) continue;
        // ---- dE of every candidate stem: only the loops it changes, from the branch list
        const double par_e = dcal_to_energy(par_dcal);
        // prefix sums of the branches' stem terms (region A is free now except, when nothing was ranked, the
        // lag values at its head), so that every loop below costs O(1) whatever its number of branches
        int *pe_ext = (int *)(lds + lay.offA + (inplace ? 0 : 8 * Pk));
        int *pe_ml = pe_ext + (nbr + 1);
        uint16_t *psp = (uint16_t *)(pe_ml + (nbr + 1));
        if (tid < 64) {
            int c_e = 0, c_m = 0, c_s = 0;
            for (int base = 0; base < nbr; base += 64) {
                const int i = base + tid;
                int ve = 0, vm = 0, vs = 0;
                if (i < nbr) {
                    const uint32_t u = brl[i];
                    const int p = (int)(u & 0xffffu), q = (int)(u >> 16);
                    const int tt = pair_type(Sl[p], Sl[q]);
                    if (ci < 0) ve = e_stem(T, tt, p > 0 ? (int)Sl[p - 1] : -1, q < L - 1 ? (int)Sl[q + 1] : -1, true);
                    vm = e_stem(T, tt, p > 0 ? (int)Sl[p - 1] : 0, q < L - 1 ? (int)Sl[q + 1] : 0, false);
                    vs = q - p + 1;
                }
                const int xe = wave_incl_scan(ve), xm = wave_incl_scan(vm), xs = wave_incl_scan(vs);
                if (i < nbr) { pe_ext[i] = c_e + xe - ve; pe_ml[i] = c_m + xm - vm; psp[i] = (uint16_t)(c_s + xs - vs); }
                c_e += __builtin_amdgcn_readlane(xe, 63); c_m += __builtin_amdgcn_readlane(xm, 63); c_s += __builtin_amdgcn_readlane(xs, 63);
            }
            if (tid == 0) { pe_ext[nbr] = c_e; pe_ml[nbr] = c_m; psp[nbr] = (uint16_t)c_s; }
        }
        ESYNC();
        FSTAMP(10);  // (dE: branch prefix sums)
        const BrPrefix pf{pe_ext, pe_ml, psp};
        const BrList all_br{brl, 0, nbr, 0, 0, 0, 0, 0};
        int g_old = 0;           // (g: the energy involves a rule / model value of the built-in tables - SmallT::lsb)
        const int e_old = loop_energy_pre(T, B, Sl, L, ci, cj, all_br, pf, g_old);      // the loop as it is (same for every stem)
        // (round 5) the lags that gave a stem, compacted: two lags in three do, and the loop below - a lane per stem, every lane on
        // its own path through the loop energies - takes ceil(stems / 64) rounds instead of ceil(lags / 64): one instead of two for
        // half of the regions of the one-wavefront class
        int nst = 0;
        for (int base = 0; base < Kp; base += NT) {
            const int r = base + tid;
            const int f = (r < Kp && wnb[r] > 0) ? 1 : 0;
            if (r < Kp) keep[r] = 0;
            int tot, ex = block_exscan_flag<NT>(f, misc + 16, &tot);
            if (f) widx[nst + ex] = (uint16_t)r;
            nst += tot;
        }
        ESYNC();
        for (int rep_ = 0; rep_ < 1 + ((rep >> 3) & 1); rep_++)
            for (int si = tid; si < nst; si += NT) {
                const int r = widx[si];
                const int nb = wnb[r];
                {
                    int g = g_old;
                    const int mi = wmi[r], mj = (int)rk[r] - mi;
                    const int a0 = pos[mi], b0 = pos[mj], ao = pos[mi - nb + 1], bo = pos[mj + nb - 1];
                    int lo, hi, lo_o, hi_o;
                    br_lower4(brl, nbr, a0, b0, ao, bo, lo, hi, lo_o, hi_o);
                    BrList outer{brl, 0, lo_o, hi_o, nbr, 1, ao, bo};
                    int e_new = loop_energy_pre(T, B, Sl, L, ci, cj, outer, pf, g);
                    BrList inner{brl, lo, hi, 0, 0, 0, 0, 0};
                    e_new += loop_energy_pre(T, B, Sl, L, a0, b0, inner, pf, g);
                    // the stem itself: a contiguous one (both strands without a gap - nearly all of them) of up to 16 pairs takes its
                    // stacking energies from the packed strands, one look-up per pair (stem_stack_windows); the others pair by pair
                    if (CODE_LDS && nb <= 16 && a0 - ao == nb - 1 && bo - b0 == nb - 1)
                        e_new += stem_stack_windows(T, strand_window(P2, mi - nb + 1), strand_window(P2, mj), nb);
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
                    dd[r] = ddc;
                    const double dE = dcal_to_energy(par_dcal + ddc) - par_e;
                    keep[r] = (uint16_t)(((dE < d.min_nrj) ? 1 : 0) | (g ? 2 : 0) | 4);     // bit 0 kept, bit 1 involves a rule / model value, bit 2 evaluated
                    if (dbg.ddcal) dbg.ddcal[r] = ddc;
                }
            }
        if (dbg.ddcal) for (int r = tid; r < Kp; r += NT) if (wnb[r] == 0) dbg.ddcal[r] = INT_MIN;
        ESYNC();

        FSTAMP(11);  // (dE: the loop as it is + every candidate)
        ESTAMP(6);   // dE
        // ---- stable sort of the kept candidates by dE (ties keep lag-rank order), emit
        int nkept = 0;
        if constexpr (NT == 64) {
            // (round 5) One wavefront: no key array is built.  The kept flags of every slab of 64 lags are a ballot (kept in the team's
            // LDS: nb_mode may ask for up to eight slabs) and the kept lags are compacted in place; the candidate slots are handed out
            // by lane 0 and reach the other lanes through readfirstlane instead of an LDS word and a fence; a kept candidate finds its
            // rank by walking the ballots - a scalar loop over the handful of kept lags, their dE read as LDS broadcasts - and only
            // a dE tie looks at (value, lag).  (Rounds 1-4: packed keys in region A, three fences, two of them around a one-lane
            // section - a quarter of the kernel's cycles for five candidates per region.)
            unsigned long long *kbs = (unsigned long long *)&misc[8];      // [8] kept ballots by slab
            for (int base = 0; base < Kp; base += 64) {
                const int r = base + tid;
                const int kf = (r < Kp) ? keep[r] : 0;
                const unsigned long long bal = __ballot((kf & 1) != 0);
                if (T->lsb) {                 // (built-in tables: how many stem energies of this launch involved a rule / model value)
                    const int ne = __popcll(__ballot((kf & 4) != 0)), ng = __popcll(__ballot((kf & 6) == 6)), nk = __popcll(__ballot((kf & 3) == 3));
                    if (tid == 0) { atomicAdd(&misc[24], ne); if (ng) atomicAdd(&misc[25], ng); if (nk) atomicAdd(&misc[26], nk); }
                }
                if (tid == 0) kbs[base >> 6] = bal;
                // (the kept lags, compacted in place: one pass of the emit body below serves them all, whichever slab they came from;
                //  nkept + pre <= r - a flag that has not been read yet is never overwritten)
                if (kf & 1) keep[nkept + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (uint16_t)r;
                nkept += __popcll(bal);
            }
            st_items++; st_n += n; st_lags += Kp; st_nbr += nbr;
            unsigned long long cbase = 0;
            int ovf_i = 0;
            if (nkept) {
                unsigned long long b0 = 0;
                const bool fresh = (unsigned)nkept > slab_left;      // reserve a new slab of candidate slots (the rest of the old one is dropped)
                const unsigned slab = d.cand_shard_cap >= 64u * (unsigned)d.cand_slab ? (unsigned)d.cand_slab : 16u;
                const unsigned want = (unsigned)nkept > slab ? (unsigned)nkept : slab;
                if (fresh) {
                    if (tid == 0) b0 = atomicAdd(&d.c->cand[shard].v, (unsigned long long)want);
                    b0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b0);
                    if (b0 + want > d.cand_shard_cap) { if (tid == 0) atomicOr(&d.c->overflow, OVF_CAND); ovf_i = 1; slab_left = 0; }
                    else { slab_base = (unsigned long long)shard * d.cand_shard_cap + b0; slab_left = want; }
                }
                if (!ovf_i) { cbase = slab_base; slab_base += nkept; slab_left -= nkept; }
            }
            wave_sync();                      // the ballots are in LDS
            FSTAMP(13);  // (emit: counts, candidate slots)
            if (nkept && !ovf_i)
            for (int rep_ = 0; rep_ < 1 + ((rep >> 6) & 1); rep_++)
            for (int x = tid; x < nkept; x += 64) {
                {
                    const int r = keep[x];
                    const int my = dd[r];
                    int rank = 0;
                    for (int b2 = 0; b2 < Kp; b2 += 64) {
                        const unsigned long long mv = kbs[b2 >> 6];
                        unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)mv);
                        while (m) {
                            const int y = b2 + __ffsll((long long)m) - 1;
                            m &= m - 1;
                            const int dy = dd[y];
                            if (dy < my) rank++;
                            else if (dy == my && y != r) {               // dE tie: lag-rank order, i.e. (value desc, lag desc)
                                if (inplace) rank += y < r ? 1 : 0;      // (sorted in place: the index IS the lag's rank)
                                else {
                                    const int lagq = rk[y], lagr = rk[r];
                                    const double qv = keyv[lagq], myv = keyv[lagr];
                                    rank += ((qv > myv) || (qv == myv && lagq > lagr)) ? 1 : 0;
                                }
                            }
                        }
                    }
                    FSTAMP(14);  // (emit: rank)
                    const int mi = wmi[r], mj = (int)rk[r] - mi, nb = wnb[r];
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
                    cd.ddcal = my; cd.mi = (uint16_t)mi; cd.mj = (uint16_t)mj; cd.nb = (uint16_t)nb;
                    { int c0, c1, c2, c3; br_lower4(brl, nbr, a0, b0, ao, bo, c0, c1, c2, c3); cd.set_cuts(c0, c1, c2, c3); }
                    cd.h1 = h1; cd.h2 = h2;
                    if (!dry) { d.cand[cbase + rank] = cd; d.cslot[cbase + rank] = 0ULL; }   // (both child slots: nobody has asked yet)
                    if (dbg.kept) dbg.kept[rank] = r;
                }
            }
            if (tid == 0 && !dry) {
                d.nd[nid].cand = cbase;
                d.nd[nid].ncand = ovf_i ? 0 : nkept;
                if (dbg.n_ranked) dbg.n_ranked[1] = nkept;
            }
        } else {
        // compact the kept lags (keep[] becomes the list of their indices)
        {
            int *wave_tot = misc + 16;
            const int lane = tid & 63, wv = tid >> 6;
            for (int base = 0; base < Kp; base += NT) {
                const int r = base + tid;
                const int kf = (r < Kp) ? keep[r] : 0;
                const int f = kf & 1;
                ESYNC();                      // everyone has read keep[] of this slab
                const unsigned long long bal = __ballot(f != 0);
                if (T->lsb) {                 // (built-in tables: how many stem energies of this launch involved a rule / model value -
                    //  counted in the team's LDS, not in registers that would live across the whole region loop)
                    const int ne = __popcll(__ballot((kf & 4) != 0)), ng = __popcll(__ballot((kf & 6) == 6)), nk = __popcll(__ballot((kf & 3) == 3));
                    if (lane == 0) { atomicAdd(&misc[24], ne); if (ng) atomicAdd(&misc[25], ng); if (nk) atomicAdd(&misc[26], nk); }
                }
                int pre = __popcll(bal & ((1ULL << lane) - 1));
                if (NT > 64) {
                    if (lane == 0) wave_tot[wv] = __popcll(bal);
                    ESYNC();
                    int tot = 0;
                    for (int w = 0; w < NT / 64; w++) { if (w < wv) pre += wave_tot[w]; tot += wave_tot[w]; }
                    if (f) keep[nkept + pre] = (uint16_t)r;   // nkept + pre <= r: never clobbers an unread flag
                    nkept += tot;
                    ESYNC();
                } else {
                    if (f) keep[nkept + pre] = (uint16_t)r;
                    nkept += __popcll(bal);
                }
            }
            ESYNC();
        }
        FSTAMP(12);  // (emit: compaction)
        if (tid == 0) {
            unsigned long long base = 0;
            misc[2] = 0;
            if (nkept) {
                if ((unsigned)nkept > slab_left) {      // reserve a new slab of candidate slots (the rest of the old one is dropped)
                    const unsigned slab = d.cand_shard_cap >= 64u * (unsigned)d.cand_slab ? (unsigned)d.cand_slab : 16u;
                    const unsigned want = (unsigned)nkept > slab ? (unsigned)nkept : slab;
                    unsigned long long b0 = atomicAdd(&d.c->cand[shard].v, (unsigned long long)want);
                    if (b0 + want > d.cand_shard_cap) { atomicOr(&d.c->overflow, OVF_CAND); misc[2] = 1; slab_left = 0; }
                    else { slab_base = (unsigned long long)shard * d.cand_shard_cap + b0; slab_left = want; }
                }
                if (!misc[2]) { base = slab_base; slab_base += nkept; slab_left -= nkept; }
            }
            *(unsigned long long *)&misc[4] = base;
            st_items++; st_n += n; st_lags += Kp; st_nbr += nbr;
        }
        ESYNC();
        FSTAMP(13);  // (emit: candidate slots)
        const unsigned long long cbase = *(unsigned long long *)&misc[4];
        const bool ovf = misc[2] != 0;
        if (!ovf)
        for (int rep_ = 0; rep_ < 1 + ((rep >> 6) & 1); rep_++) {
            // packed sort key of every kept candidate: (dE biased to unsigned) << 32 | lag rank.  (They take the place of
            // the branch prefix sums in region A, which dE is done with: 8 * Kp bytes behind the lag values.)
            unsigned long long *ck = (unsigned long long *)(lds + lay.offA + (inplace ? 0 : 8 * Pk));
            for (int x = tid; x < nkept; x += NT) {
                const int r = keep[x];
                ck[x] = ((unsigned long long)((unsigned)dd[r] ^ 0x80000000u) << 32) | (unsigned)r;
            }
            ESYNC();
            for (int x = tid; x < nkept; x += NT) {
                const unsigned long long kx = ck[x];
                const int r = (int)(kx & 0xFFFFFFFFu);
                const int my = dd[r];
                int rank = 0;
                if (inplace) {                                   // r is the lag's rank
                    for (int y = 0; y < nkept; y++) rank += ck[y] < kx ? 1 : 0;
                } else {                                         // rk[] is in no particular order: compare (value, lag)
                    const int lagr = rk[r];
                    const double myv = keyv[lagr];
                    for (int y = 0; y < nkept; y++) {
                        const unsigned long long ky = ck[y];
                        if ((ky >> 32) == (kx >> 32)) {          // dE tie: (value desc, lag desc)
                            const int q = (int)(ky & 0xFFFFFFFFu), lagq = rk[q];
                            const double qv = keyv[lagq];
                            rank += (q != r && ((qv > myv) || (qv == myv && lagq > lagr))) ? 1 : 0;
                        } else
                            rank += ky < kx ? 1 : 0;
                    }
                }
                FSTAMP(14);  // (emit: sort keys, rank)
                int mi = wmi[r], mj = (int)rk[r] - mi, nb = wnb[r];
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
                cd.ddcal = my; cd.mi = (uint16_t)mi; cd.mj = (uint16_t)mj; cd.nb = (uint16_t)nb;
                { int c0, c1, c2, c3; br_lower4(brl, nbr, a0, b0, ao, bo, c0, c1, c2, c3); cd.set_cuts(c0, c1, c2, c3); }
                cd.h1 = h1; cd.h2 = h2;
                if (!dry) { d.cand[cbase + rank] = cd; d.cslot[cbase + rank] = 0ULL; }   // (both child slots: nobody has asked yet)
                if (dbg.kept) dbg.kept[rank] = r;
            }
        }
        if (tid == 0 && !dry) {
            d.nd[nid].cand = cbase;
            d.nd[nid].ncand = ovf ? 0 : nkept;
            if (dbg.n_ranked) dbg.n_ranked[1] = nkept;
        }
        }
        FSTAMP(15);  // (emit: pair hashes, cuts, stores)
        ESTAMP(7);   // emit
        if (NT == 64 && prof_e != nullptr) {         // diagnostic: regions without any candidate stem / without a kept one, by size
            int has = 0;
            for (int r = tid; r < Kp; r += NT) has |= wnb[r] > 0 ? 1 : 0;
            const bool anystem = __ballot(has) != 0ULL;
            if (tid == 0 && !anystem) atomicAdd(&prof_e[cls * PROF_E + 80 + size_bk], 1ULL);
            if (tid == 0 && nkept == 0) atomicAdd(&prof_e[cls * PROF_E + 88 + size_bk], 1ULL);
        }
        if (eprof) { atomicAdd(&prof_e[cls * PROF_E + 8 + size_bk], 1ULL); atomicAdd(&prof_e[cls * PROF_E + 16 + size_bk], (unsigned long long)(clock64() - t_region0)); }
    }
    if (eprof) {
        for (int k = 0; k < 8; k++) atomicAdd(&prof_e[cls * PROF_E + k], eacc[k]);
        atomicAdd(&prof_e[cls * PROF_E + 40], eacc[8]); atomicAdd(&prof_e[cls * PROF_E + 41], eacc[9]);
        for (int k = 10; k < 16; k++) atomicAdd(&prof_e[cls * PROF_E + 32 + k], eacc[k]);
    }
#undef ESTAMP
#undef FSTAMP
    if (tid == 0 && st_items) {
        Counters::StatLine *sl = &d.c->xstat[cls][gteam & (NSHARD - 1)];
        atomicAdd(&sl->items, st_items);
        atomicAdd(&sl->n, st_n);
        atomicAdd(&sl->lags, st_lags);
        atomicAdd(&sl->nbr, st_nbr);
    }
    if (tid == 0 && misc[24]) {
        Counters::StatLine *sl = &d.c->xstat[cls][gteam & (NSHARD - 1)];
        atomicAdd(&sl->evals, (unsigned long long)(unsigned)misc[24]);
        if (misc[25]) atomicAdd(&sl->guessed, (unsigned long long)(unsigned)misc[25]);
        if (misc[26]) atomicAdd(&sl->kept_guessed, (unsigned long long)(unsigned)misc[26]);
    }
}

#include "rafft_expand_small.hip"

// --------------------------------------------------------- beam step kernel

// (round 5: four slots per round trip.  A wavefront waits for the longest probe chain among its 64 lanes - at half load that was four or
//  five dependent trips for a lookup whose expected length is 1.5; the four 16-byte loads are independent and mostly one 64-byte line.
//  `free_sl`: the empty slot that ended the search - where seen_insert_at starts, every slot before it holds another key for good)
__device__ inline bool seen_lookup(const uint64_t *tab, uint32_t cap, uint64_t h1, uint64_t h2, uint32_t &free_sl)
{
    const uint32_t mask = cap - 1;
    uint32_t sl = (uint32_t)h1 & mask;
    const ulonglong2 *t2 = (const ulonglong2 *)tab;
    for (;;) {
        const uint32_t s1 = (sl + 1) & mask, s2 = (sl + 2) & mask, s3 = (sl + 3) & mask;
        ulonglong2 e0 = t2[sl], e1 = t2[s1], e2 = t2[s2], e3 = t2[s3];
        pin(e0); pin(e1); pin(e2); pin(e3);           // (all four in flight: without this the compiler loads a slot when the one before did not decide)
        if (e0.x == 0) { free_sl = sl; return false; }
        if (e0.x == h1 && e0.y == h2) return true;
        if (e1.x == 0) { free_sl = s1; return false; }
        if (e1.x == h1 && e1.y == h2) return true;
        if (e2.x == 0) { free_sl = s2; return false; }
        if (e2.x == h1 && e2.y == h2) return true;
        if (e3.x == 0) { free_sl = s3; return false; }
        if (e3.x == h1 && e3.y == h2) return true;
        sl = (sl + 4) & mask;
    }
}
// insert a key that seen_lookup did not find, starting at the empty slot it stopped at (other threads of the pass may have taken it since)
__device__ inline void seen_insert_at(uint64_t *tab, uint32_t cap, uint64_t h1, uint64_t h2, uint32_t sl)
{
    const uint32_t mask = cap - 1;
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long *)&tab[2 * (uint64_t)sl], 0ULL, (unsigned long long)h1);
        if (old == 0) { tab[2 * (uint64_t)sl + 1] = h2; return; }
        if (old == h1 && tab[2 * (uint64_t)sl + 1] == h2) return;
        sl = (sl + 1) & mask;
    }
}
// insert unless present; true if it was new.  Keys inserted concurrently by other threads are always
// different structures (distinct combos of one parent), so a half-written entry can only be someone else's.
__device__ inline bool seen_insert_new(uint64_t *tab, uint32_t cap, uint64_t h1, uint64_t h2)
{
    uint32_t mask = cap - 1, sl = (uint32_t)h1 & mask;
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long *)&tab[2 * (uint64_t)sl], 0ULL, (unsigned long long)h1);
        if (old == 0) { tab[2 * (uint64_t)sl + 1] = h2; return true; }
        if (old == h1 && tab[2 * (uint64_t)sl + 1] == h2) return false;
        sl = (sl + 1) & mask;
    }
}
__device__ inline void seen_insert(uint64_t *tab, uint32_t cap, uint64_t h1, uint64_t h2)
{
    uint32_t mask = cap - 1, sl = (uint32_t)h1 & mask;
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long *)&tab[2 * (uint64_t)sl], 0ULL, (unsigned long long)h1);
        if (old == 0) { tab[2 * (uint64_t)sl + 1] = h2; return; }
        if (old == h1 && tab[2 * (uint64_t)sl + 1] == h2) return;
        sl = (sl + 1) & mask;
    }
}


// move every key of a sequence's `seen` set into a bigger, zeroed table.  (round 5: four slots per thread read together and their
// compare-and-swaps issued together - one slot at a time was a chain of two or three dependent round trips per slot, 32 slots per
// thread for the first growth: ~60 us of a workgroup's ~250)
template <int NT>
__device__ inline void seen_rehash(const uint64_t *stab, uint32_t scap, uint64_t *ntab, uint32_t ncap, int tid)
{
    const ulonglong2 *src = (const ulonglong2 *)stab;
    const uint32_t mask = ncap - 1;
    for (uint32_t base = 0; base < scap; base += NT * 4) {
        ulonglong2 e[4];
        unsigned long long old[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = base + (uint32_t)u * NT + (uint32_t)tid; e[u] = i < scap ? src[i] : make_ulonglong2(0ULL, 0ULL); }
#pragma unroll
        for (int u = 0; u < 4; u++) { old[u] = 1; if (e[u].x) old[u] = atomicCAS((unsigned long long *)&ntab[2 * (uint64_t)((uint32_t)e[u].x & mask)], 0ULL, (unsigned long long)e[u].x); }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (e[u].x) {
                if (old[u] == 0) ntab[2 * (uint64_t)((uint32_t)e[u].x & mask) + 1] = e[u].y;
                else seen_insert(ntab, ncap, e[u].x, e[u].y);        // home slot taken: the probing insert
            }
    }
}

struct ParentInfo {         // filled by the parallel prepass, one entry per beam member
    unsigned long long total, cur;        // product size, cursor
    unsigned long long h1, h2;            // pair-set hash of combo 0 (absolute)
    int dcal0, flag;        // energy of combo 0; flag: 0 live, 1 nothing to produce, 2 cursor already moved
    int sid, nprod;         // structure id; productive regions
    unsigned long long prod;              // productive-region list (global)
    int rl0, nrl;           // this member's regions with >= 2 candidates in the LDS list (rl0 < 0: not resident)
};
static_assert(sizeof(ParentInfo) == 64, "ParentInfo layout");

__device__ __forceinline__ unsigned long long sat_mul(unsigned long long a, unsigned long long b)
{
    const unsigned long long lim = 1ULL << 62;
    if (a == 0 || b == 0) return 0;
    return (a > lim / b) ? lim : a * b;
}

// LDS: sort keys (dynamic) + product description + per-parent prepass records
template <int BS_NT, bool PROD = false>      // (PROD: the diagnostic stamps compiled out - see expand_kernel)
__global__ __launch_bounds__(BS_NT, BS_NT == 256 ? 5 : 1) void beam_step_kernel(Dev d, int sort_cap)
{
    extern __shared__ __align__(16) unsigned char lds[];
    // region 0 is time-shared: scratch of the product walk (per-thread keys + dedupe table), then the sort keys
    const size_t r0 = max((size_t)8 * (size_t)sort_cap, (size_t)24 * BS_NT);
    unsigned long long *skey = (unsigned long long *)lds;                       // [sort_cap]
    unsigned long long *wk_h1 = (unsigned long long *)lds;                      // [BS_NT] keys of this chunk's combos
    unsigned long long *wk_h2 = wk_h1 + BS_NT;                                  // [BS_NT]
    unsigned int *wk_tab = (unsigned int *)(wk_h2 + BS_NT);                     // [2 BS_NT] first claimant of a key
    unsigned long long *rl_off = (unsigned long long *)(lds + r0);              // [RL_CAP] candidate offset of a region
    int *rl_cnt = (int *)(rl_off + RL_CAP);                                     // [RL_CAP] its candidate count (>= 2)
    ParentInfo *pinfo = (ParentInfo *)(rl_cnt + RL_CAP);                        // [B]
    unsigned long long *ppre = (unsigned long long *)(pinfo + d.B);             // [B + 1] flat positions of the products
    int *oldbeam = (int *)(ppre + d.B + 1);                                     // [B]
    int *sh = oldbeam + ((d.B + 3) & ~3);                                       // scratch [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sq = blockIdx.x;
    // snapshot of the region allocators: whatever materialize adds after this kernel is "new"
    if (sq == 0 && tid < NSHARD) d.c->node_prev[tid].v = d.c->node[tid].v;
    // the expand kernels of this step are done with their work lists: reset them for dedupe_kernel / the next step
    if (sq == 0 && tid < NCLS) d.c->n_work[tid].v = 0;
    if (sq == 0) for (int i = tid; i < NCLS * NSHARD; i += BS_NT) d.c->wcur[i / NSHARD][i % NSHARD].v = 0;
    if (sq == 0 && tid < NCLS) d.c->wdone[tid] = 0;
    if (d.done[sq]) return;
    // an arena overflowed while the last step's structures were materialized: some child slots were claimed and never filled, some
    // node lists point at them.  Nothing of that step may be read; the host sees the flag in this step's read-back and regrows.
    // (one thread looks: other workgroups of this launch may set the flag while this one starts)
    if (tid == 0) sh[27] = d.c->overflow != 0 ? 1 : 0;
    __syncthreads();
    if (sh[27]) return;
    const bool prof = !PROD && d.prof && (d.prof_seq < 0 || sq == d.prof_seq) && tid == 0;   // diagnostic stamps (RAFFT_TRACE=3; RAFFT_PROF_SEQ=-1: summed over all sequences)
    unsigned long long tprev = prof ? clock64() : 0;
    unsigned long long *const prof_ws = PROD ? nullptr : d.prof_ws;
    const unsigned long long t_begin = prof_ws ? clock64() : 0;
    unsigned long long n_chunks = 0, n_par = 0, n_combos = 0;
#define WS_END() do { if (prof_ws && tid == 0) { unsigned long long dt_ = clock64() - t_begin; prof_ws[3 * sq] += dt_; prof_ws[3 * sq + 1] += n_chunks | (n_combos << 24); \
        prof_ws[3 * sq + 2] += n_par; } } while (0)
#define STAMP(k) do { if (prof) { unsigned long long tn_ = clock64(); atomicAdd(&d.prof[k], tn_ - tprev); tprev = tn_; } } while (0)
    const int nbeam = d.beam_n[sq];
    const int step_no = d.nsteps[sq];          // (read by everyone before the barrier below; thread 0 counts the step after it)
    int *beam = d.beam + (size_t)sq * d.B;
    for (int i = tid; i < nbeam; i += BS_NT) oldbeam[i] = beam[i];
    __syncthreads();

    // glob_traj += [glob_tree]   (rafft/rafft.py:161)
    if (d.traj) {
        if (tid == 0) {
            unsigned long long r = atomicAdd(&d.c->trec_n, 1ULL);
            unsigned long long o = atomicAdd(&d.c->tsid_top, (unsigned long long)nbeam);
            if (r >= d.trec_cap || o + nbeam > d.tsid_cap) { atomicOr(&d.c->overflow, OVF_TRAJ); sh[0] = -1; }
            else { d.trec[r] = make_int4(sq, d.nsteps[sq], nbeam, (int)o); sh[0] = (int)o; }
        }
        __syncthreads();
        int o = sh[0];
        if (o >= 0) for (int i = tid; i < nbeam; i += BS_NT) d.tsid[o + i] = oldbeam[i];
        __syncthreads();
    }
    if (tid == 0) d.nsteps[sq] += 1;

    // ---- prepass: product size and combo 0 of every parent, its productive regions as a compact list in HBM (written on the first
    // visit, read back by later product walks and by materialize_kernel) and its regions with a real choice (>= 2 candidates) as a
    // (count, offset) list in LDS.
    // Round 5: FLAT over (beam member, region) items.  Every member costs a chain of dependent loads - structure row -> node list ->
    // child slot -> region header -> first candidate - and the kernel lives on how many of those chains are in flight at once: a group
    // of 8 / 16 / 64 lanes per member walked the 50 members of a beam in two to thirteen passes of nine dependent round trips each (the
    // row's fields, the list allocation and the header's two words were trips of their own).  Now one thread per member reads its whole
    // row (one trip), a prefix sum over the region counts numbers the items, and one thread per item runs the remaining four trips -
    // 200 items of a short sequence's beam in ONE pass of a 256-thread workgroup; sums go to the member's record with LDS atomics, the
    // lists are placed by a workgroup-wide prefix sum (member-major, node order: rafft/rafft.py:166-171).
    {
        unsigned long long *istart = ppre;                 // [B + 1] first item of a member (ppre proper is written after the prepass)
        int *pnode0 = sh + 32;                             // [B] first node-list entry of a member on its first visit
        unsigned int irun = 0;
        if (tid == 0) sh[28] = 0;                          // regions of this step's first visits
        __syncthreads();
        for (int b0 = 0; b0 < nbeam; b0 += BS_NT) {
            const int b = b0 + tid;
            int ic = 0;
            if (b < nbeam) {
                const int sid = oldbeam[b];
                // the whole row, one round trip (seven 16-byte loads pinned: as a struct copy the compiler split it into the fields each
                // branch below uses and loaded them there - two or three dependent trips)
                StRec r;
                {
                    const uint4 *rp = (const uint4 *)&d.st[sid];
                    uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3], q4 = rp[4], q5 = rp[5], q6 = rp[6];
                    pin(q0); pin(q1); pin(q2); pin(q3); pin(q4); pin(q5); pin(q6);
                    auto u64 = [](unsigned int lo, unsigned int hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); };
                    r.dcal = (int)q0.y; r.node0 = (int)q0.z; r.nnodes = (int)q0.w; r.nprod = (int)q1.y; r.c0d = (int)q1.z;
                    r.h1 = u64(q2.x, q2.y); r.h2 = u64(q2.z, q2.w); r.cursor = u64(q3.z, q3.w); r.total = u64(q4.z, q4.w);
                    r.prod = u64(q5.x, q5.y); r.c0h1 = u64(q5.z, q5.w); r.c0h2 = u64(q6.x, q6.y);
                }
                ParentInfo pi;
                pi.sid = sid; pi.rl0 = 0; pi.nrl = 0; pi.prod = 0; pi.nprod = 0; pi.total = 0; pi.cur = 0; pi.h1 = 0; pi.h2 = 0; pi.dcal0 = 0;
                if (r.total && r.cursor >= r.total) pi.flag = 1;                       // product exhausted
                else if (r.total && r.cursor > 0) {      // expanded in an earlier step: cursor, total and combo 0 are on record
                    pi.flag = 2; pi.total = r.total; pi.cur = r.cursor; pi.h1 = r.c0h1; pi.h2 = r.c0h2; pi.dcal0 = r.c0d;
                    pi.prod = r.prod; pi.nprod = r.nprod; ic = r.nprod;
                } else {                                  // first visit: combo 0 = the first candidate of every region that has one
                    pi.flag = 0; pi.total = 1; pi.h1 = r.h1; pi.h2 = r.h2; pi.dcal0 = r.dcal;
                    pnode0[b] = r.node0; ic = r.nnodes;
                    if (ic) atomicAdd(&sh[28], ic);
                }
                pinfo[b] = pi;
            }
            int tot, ex = block_exscan<BS_NT>(ic, sh, &tot);
            if (b < nbeam) istart[b] = irun + (unsigned int)ex;
            irun += (unsigned int)tot;
            __syncthreads();
        }
        const int nitems = (int)irun;
        // ONE allocation for the productive-region lists of all first visits of the step (at most one entry per region): a returning
        // atomic per member was 50 per sequence and step on the 64 sub-arena counters - same-address atomics are served one after the
        // other (1.20 -> 1.29 ms per batch).  The sub-arena rotates with the step, so that a lone sequence spreads over all of them.
        // The answer is needed when the lists are written, i.e. after the item loads below are under way: kept in a register till then.
        const int nfirst = sh[28], pshard = (sq + step_no) & (NSHARD - 1);
        unsigned long long pb_raw = 0, pball = 0;
        if (tid == BS_NT - 1 && nfirst) pb_raw = atomicAdd(&d.c->prod[pshard].v, (unsigned long long)nfirst);
        unsigned int run1 = 0, run2 = 0;
        for (int t0 = 0; t0 < nitems; t0 += BS_NT) {
            const int t = t0 + tid;
            int b = -1, i = 0, cnt = 0, cn = -1, first = 0;
            unsigned long long coff = 0;
            if (t < nitems) {
                int lo = 0, hi = nbeam - 1;              // the last member whose items start at or before t
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (istart[mid] <= (unsigned long long)t) lo = mid; else hi = mid - 1; }
                b = lo; i = t - (int)istart[b];
                first = pinfo[b].flag == 0 ? 1 : 0;
                if (first) {
                    cn = d.nlist[pnode0[b] + i];
                    // (written by materialize_kernel as -(slot + 1): the region that hangs in that child slot - created there by
                    //  whichever beam member asked first, or the known loop dedupe_kernel found for it)
                    if (cn < 0) cn = (int)(((const uint32_t *)d.cslot)[-cn - 1] & 0x7FFFFFFFu) - 1;
                    if (cn >= 0) { cnt = d.nd[cn].ncand; coff = d.nd[cn].cand; }
                    if (cnt > 0) {
                        const Cand *cp = &d.cand[coff];
                        const int dd = cp->ddcal;
                        const ulonglong2 hh = *(const ulonglong2 *)&cp->h1;
                        atomicAdd(&pinfo[b].dcal0, dd);
                        atomicAdd(&pinfo[b].h1, hh.x); atomicAdd(&pinfo[b].h2, hh.y);
                        if (cnt >= 2) {                  // product size: a saturating product commutes (every factor >= 1)
                            unsigned long long old = pinfo[b].total, seen_;
                            do { seen_ = old; old = atomicCAS(&pinfo[b].total, seen_, sat_mul(seen_, (unsigned long long)cnt)); } while (old != seen_);
                        }
                    }
                } else {
                    const ProdEnt pe = d.prod[pinfo[b].prod + i];
                    cnt = (int)pe.cnt; coff = pe.off; cn = pe.node;
                }
            }
            if (t0 == 0 && tid == BS_NT - 1) {
                unsigned long long v = pb_raw;
                if (v + (unsigned long long)nfirst > d.prod_shard_cap) { atomicOr(&d.c->overflow, OVF_PRODLIST); v = ~0ULL; }
                else v += (unsigned long long)pshard * d.prod_shard_cap;
                *(unsigned long long *)&sh[30] = v;
            }
            const int f1 = (first && cnt > 0) ? 1 : 0, f2 = cnt >= 2 ? 1 : 0;
            int tot12, ex12 = block_exscan<BS_NT>(f1 | (f2 << 16), sh, &tot12);      // (barriers inside: sh[30] is there for everyone)
            if (t0 == 0) pball = *(const unsigned long long *)&sh[30];
            const unsigned int pos1 = run1 + (unsigned int)(ex12 & 0xFFFF), pos2 = run2 + (unsigned int)(ex12 >> 16);
            if (b >= 0) {
                if (i == 0) { pinfo[b].rl0 = (int)pos2; if (first) pinfo[b].prod = pball == ~0ULL ? ~0ULL : pball + pos1; }
                if (f1) {
                    atomicAdd(&pinfo[b].nprod, 1);
                    if (pball != ~0ULL) { ProdEnt pe; pe.cnt = (uint32_t)cnt; pe.node = cn; pe.off = coff; d.prod[pball + pos1] = pe; }
                }
                if (f2) {
                    atomicAdd(&pinfo[b].nrl, 1);
                    if ((int)pos2 < d.rl_cap) { rl_cnt[pos2] = cnt; rl_off[pos2] = coff; }
                }
            }
            run1 += (unsigned int)(tot12 & 0xFFFF); run2 += (unsigned int)(tot12 >> 16);
            __syncthreads();
        }
        for (int b = tid; b < nbeam; b += BS_NT) {
            ParentInfo pi = pinfo[b];
            if (pi.flag == 1) continue;
            if (pi.nrl == 0) pi.rl0 = 0;
            else if (pi.rl0 + pi.nrl > d.rl_cap) pi.rl0 = -1;      // a list that does not fit whole is read from HBM by the walk
            if (pi.flag == 0) {
                if (pi.prod == ~0ULL) { pi.prod = 0; pi.nprod = 0; pi.nrl = 0; pi.rl0 = 0; pi.total = 1; }
                const int np = pi.nprod;
                if (np > d.max_prod) atomicOr(&d.c->overflow, OVF_PROD);     // materialize_kernel's limit
                if (np > 64) atomicMax(&d.c->max_nprod, (unsigned int)np);
                StRec *sr = &d.st[pi.sid];
                sr->prod = pi.prod; sr->nprod = np; sr->c0h1 = pi.h1; sr->c0h2 = pi.h2; sr->c0d = pi.dcal0;
                if (np == 0) { pi.flag = 1; sr->total = 1; sr->cursor = 1; }
            }
            pinfo[b] = pi;
        }
    }
    __syncthreads();
    STAMP(0);

    // ---- the product walk (rafft/rafft.py:173-204), flat over all parents: position p of the walk is combo
    // cur_b + (p - ppre[b]) of the parent b whose range holds p, in beam order and itertools.product order.
    // One chunk of BS_NT consecutive positions per pass - usually several whole parents at once.
    if (wv == 0) {
        unsigned long long carry = 0;
        const unsigned long long LIM = 1ULL << 63;
        for (int base = 0; base < nbeam; base += 64) {
            const int b = base + lane;
            unsigned long long rem = 0;
            if (b < nbeam && !(pinfo[b].flag & 1)) rem = pinfo[b].total - pinfo[b].cur;
            unsigned long long x = rem;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long y = __shfl_up(x, o, 64);
                if (lane >= o) x = (x > LIM - y) ? LIM : x + y;
            }
            unsigned long long incl = (x > LIM - carry) ? LIM : x + carry;
            if (b < nbeam) ppre[b + 1] = incl;
            carry = __shfl(incl, 63, 64);
        }
        if (lane == 0) ppre[0] = 0;
    }
    for (int i = tid; i < 2 * BS_NT; i += BS_NT) wk_tab[i] = 0;
    __syncthreads();
    uint64_t *stab = d.seen + 2 * d.seen_off[sq];
    uint32_t scap = d.seen_cap[sq], scnt = d.seen_cnt[sq];
    const size_t chb = (size_t)sq * d.ch_cap;
    int nb_branch = 0, nchild = 0;
    int single_from = nbeam;
    // once nb_branch >= max_branch every later parent only replays its combo 0
    // (rafft/rafft.py:202-203): those are handled together, in parallel, after this loop
    if (d.max_branch <= 0) single_from = 0;
    const unsigned long long Ptot = ppre[nbeam];
    unsigned long long W = 0;
    while (single_from == nbeam && W < Ptot) {
        const unsigned long long left = Ptot - W;
        const int chunk = left < (unsigned long long)BS_NT ? (int)left : BS_NT;
        if ((unsigned long long)(scnt + chunk) * 2 > scap) {   // grow the seen set (rehash into a zeroed region)
            uint32_t ncap = scap;
            while ((unsigned long long)(scnt + BS_NT) * 2 > ncap) ncap <<= 1;
            STAMP(7);
            if (prof) atomicAdd(&d.prof[14], 1ULL);
            if (tid == 0) {
                unsigned long long o = atomicAdd(&d.c->seen_top, (unsigned long long)ncap);
                if (o + ncap > d.seen_cap_total) { atomicOr(&d.c->overflow, OVF_SEEN); *(unsigned long long *)&sh[8] = ~0ULL; }
                else *(unsigned long long *)&sh[8] = o;
            }
            __syncthreads();
            unsigned long long o = *(unsigned long long *)&sh[8];
            __syncthreads();
            if (o == ~0ULL) { d.done[sq] = 1; return; }
            uint64_t *ntab = d.seen + 2 * o;
            STAMP(11);
            for (uint32_t i = tid; i < ncap; i += BS_NT) ((ulonglong2 *)ntab)[i] = make_ulonglong2(0ULL, 0ULL);   // arena is not pre-zeroed
            __syncthreads();
            STAMP(12);
            seen_rehash<BS_NT>(stab, scap, ntab, ncap, tid);
            __syncthreads();
            STAMP(13);
            stab = ntab; scap = ncap;
            if (tid == 0) { d.seen_off[sq] = o; d.seen_cap[sq] = ncap; }
        }
        n_chunks++; n_combos += chunk;
        STAMP(6);   // loop head / seen growth
        const int need = d.max_branch - nb_branch;      // > 0
        int b = 0, sidb = 0, cd = 0, slot = -1;
        uint32_t free_sl = 0;
        bool cand_new = false, last_combo = false;
        unsigned long long idx = 0, totb = 0, h1 = 0, h2 = 0;
        const unsigned long long pos = W + (unsigned long long)tid;
        if (tid < chunk) {
            int lo = 0, hi = nbeam - 1;                  // the last member whose range starts at or before pos
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ppre[mid] <= pos) lo = mid; else hi = mid - 1; }
            b = lo;
            const ParentInfo pi = pinfo[b];
            sidb = pi.sid; totb = pi.total;
            idx = pi.cur + (pos - ppre[b]);
            last_combo = idx == totb - 1;
            // combo idx = combo 0 with the digits of idx (mixed radix over the regions with a choice, last
            // region fastest) swapped in
            unsigned long long a1 = pi.h1, a2 = pi.h2, rest = idx;
            int ad = pi.dcal0;
            auto divmod = [&](unsigned int c, unsigned int &r) {
                if (rest < (1ULL << 24)) {
                    const unsigned int v = (unsigned int)rest;
                    unsigned int q = (unsigned int)((float)v * __frcp_rn((float)c));       // off by one at most
                    int rr = (int)(v - q * c);
                    if (rr < 0) { q--; rr += (int)c; } else if (rr >= (int)c) { q++; rr -= (int)c; }
                    r = (unsigned int)rr; rest = q;
                } else { const unsigned long long q = rest / c; r = (unsigned int)(rest - q * c); rest = q; }
            };
            if (pi.rl0 >= 0) {
                int j = pi.nrl - 1;
                // up to four changed digits are located first and their candidates loaded together
                // (round 5: digits that did not change point both at the arena's first record - the eight records are loaded unconditionally,
                //  i.e. together, one round trip; loads behind `if (changed)` were one dependent trip per changed digit)
                const Cand *const same = d.cand;
                const Cand *pn0 = same, *pn1 = same, *pn2 = same, *pn3 = same, *po0 = same, *po1 = same, *po2 = same, *po3 = same;
                auto next = [&](const Cand *&pn, const Cand *&po) {
                    while (j >= 0 && rest) {
                        unsigned int r;
                        divmod((unsigned int)rl_cnt[pi.rl0 + j], r);
                        j--;
                        if (r) { po = &d.cand[rl_off[pi.rl0 + j + 1]]; pn = po + r; return; }
                    }
                };
                next(pn0, po0); next(pn1, po1); next(pn2, po2); next(pn3, po3);
                {
                    const int dn0 = pn0->ddcal, dn1 = pn1->ddcal, dn2 = pn2->ddcal, dn3 = pn3->ddcal, do0 = po0->ddcal, do1 = po1->ddcal, do2 = po2->ddcal, do3 = po3->ddcal;
                    const ulonglong2 hn0 = *(const ulonglong2 *)&pn0->h1, hn1 = *(const ulonglong2 *)&pn1->h1, hn2 = *(const ulonglong2 *)&pn2->h1, hn3 = *(const ulonglong2 *)&pn3->h1;
                    const ulonglong2 ho0 = *(const ulonglong2 *)&po0->h1, ho1 = *(const ulonglong2 *)&po1->h1, ho2 = *(const ulonglong2 *)&po2->h1, ho3 = *(const ulonglong2 *)&po3->h1;
                    ad += (dn0 - do0) + (dn1 - do1) + (dn2 - do2) + (dn3 - do3);
                    a1 += (hn0.x - ho0.x) + (hn1.x - ho1.x) + (hn2.x - ho2.x) + (hn3.x - ho3.x);
                    a2 += (hn0.y - ho0.y) + (hn1.y - ho1.y) + (hn2.y - ho2.y) + (hn3.y - ho3.y);
                }
                while (j >= 0 && rest) {
                    const Cand *pn = same, *po = same;
                    next(pn, po);
                    ad += pn->ddcal - po->ddcal; a1 += pn->h1 - po->h1; a2 += pn->h2 - po->h2;
                }
            } else {
                // region list not resident in LDS (more than RL_CAP regions with a choice in this beam)
                for (int j = pi.nprod - 1; j >= 0 && rest; j--) {
                    const ProdEnt pe = d.prod[pi.prod + j];
                    if (pe.cnt < 2) continue;
                    unsigned int r;
                    divmod(pe.cnt, r);
                    if (r) { const Cand *po = &d.cand[pe.off], *pn = po + r; ad += pn->ddcal - po->ddcal; a1 += pn->h1 - po->h1; a2 += pn->h2 - po->h2; }
                }
            }
            h1 = a1 ? a1 : 1; h2 = a2 ? a2 : 1; cd = ad;
            wk_h1[tid] = h1; wk_h2[tid] = h2;
            cand_new = !seen_lookup(stab, scap, h1, h2, free_sl);
        }
        // the same structure can come from several parents of this chunk: its first position wins (`seen` order)
        if (cand_new) {
            unsigned int sl = (unsigned int)(h1 ^ (h1 >> 32)) & (2 * BS_NT - 1);
            for (;;) {
                const unsigned int old = atomicCAS(&wk_tab[sl], 0u, (unsigned int)tid + 1u);
                if (old == 0) break;
                if (wk_h1[old - 1] == h1 && wk_h2[old - 1] == h2) { atomicMin(&wk_tab[sl], (unsigned int)tid + 1u); break; }
                sl = (sl + 1) & (2 * BS_NT - 1);
            }
            slot = (int)sl;
        }
        STAMP(8);   // decode + seen lookups
        __syncthreads();
        const bool isnew = cand_new && wk_tab[slot < 0 ? 0 : slot] == (unsigned int)tid + 1u;
        const unsigned long long bal = __ballot(isnew);
        if (lane == 0) sh[wv] = __popcll(bal);
        __syncthreads();
        int ex = __popcll(bal & ((1ULL << lane) - 1)), tot = 0;
        for (int w = 0; w < BS_NT / 64; w++) { const int t = sh[w]; if (w < wv) ex += t; tot += t; }
        const bool hit = tot >= need;
        const bool accepted = isnew && ex < need;
        if (accepted) {
            const int ci2 = nchild + ex;
            if (ci2 < d.ch_cap) {
                d.ch_parent[chb + ci2] = (uint16_t)b;
                d.ch_combo[chb + ci2] = idx;
                d.ch_dcal[chb + ci2] = cd;
                d.ch_h[2 * (chb + ci2)] = h1;
                d.ch_h[2 * (chb + ci2) + 1] = h2;
            } else atomicOr(&d.c->overflow, OVF_SORT);
            seen_insert_at(stab, scap, h1, h2, free_sl);
        }
        STAMP(9);
        if (hit) {
            // the reference stops walking after the combo that brings nb_branch to max_branch
            if (accepted && ex == need - 1) {
                sh[21] = b; *(unsigned long long *)&sh[22] = pos;
                d.st[sidb].cursor = idx + 1; d.st[sidb].total = totb;
            }
            __syncthreads();
            const unsigned long long hpos = *(unsigned long long *)&sh[22];
            if (tid < chunk && last_combo && pos < hpos) { d.st[sidb].cursor = totb; d.st[sidb].total = totb; }
            nchild += need; nb_branch += need; scnt += need;
            single_from = sh[21] + 1;
            __syncthreads();
            break;
        }
        if (tid < chunk && last_combo) { d.st[sidb].cursor = totb; d.st[sidb].total = totb; }   // product exhausted
        nchild += tot; nb_branch += tot; scnt += tot;
        W += (unsigned long long)chunk;
        for (int i = tid; i < 2 * BS_NT; i += BS_NT) wk_tab[i] = 0;
        __syncthreads();
        STAMP(10);  // child records + seen insert
    }
    STAMP(1);
    if (single_from < nbeam) {
        // ---- parents in "one combo then break" mode: combo 0 of each (from the prepass), accepted in
        // beam order if its structure is new; a parent whose cursor already moved replays a known combo
        const int nrest = nbeam - single_from;
        if ((unsigned long long)(scnt + nrest) * 2 > scap) {
            uint32_t ncap = scap;
            while ((unsigned long long)(scnt + nrest + BS_NT) * 2 > ncap) ncap <<= 1;
            if (tid == 0) {
                unsigned long long o = atomicAdd(&d.c->seen_top, (unsigned long long)ncap);
                if (o + ncap > d.seen_cap_total) { atomicOr(&d.c->overflow, OVF_SEEN); *(unsigned long long *)&sh[8] = ~0ULL; }
                else *(unsigned long long *)&sh[8] = o;
            }
            __syncthreads();
            unsigned long long o = *(unsigned long long *)&sh[8];
            __syncthreads();
            if (o == ~0ULL) { d.done[sq] = 1; return; }
            uint64_t *ntab = d.seen + 2 * o;
            for (uint32_t i = tid; i < ncap; i += BS_NT) ((ulonglong2 *)ntab)[i] = make_ulonglong2(0ULL, 0ULL);   // arena is not pre-zeroed
            __syncthreads();
            seen_rehash<BS_NT>(stab, scap, ntab, ncap, tid);
            __syncthreads();
            stab = ntab; scap = ncap;
            if (tid == 0) { d.seen_off[sq] = o; d.seen_cap[sq] = ncap; }
        }
        for (int base = single_from; base < nbeam; base += BS_NT) {
            const int b = base + tid;
            int isnew = 0;
            uint32_t free_sl = 0;
            uint64_t h1 = 0, h2 = 0;
            if (b < nbeam && pinfo[b].flag == 0) {       // live and cursor == 0
                h1 = pinfo[b].h1; h2 = pinfo[b].h2;
                if (h1 == 0) h1 = 1;
                if (h2 == 0) h2 = 1;
                isnew = seen_lookup(stab, scap, h1, h2, free_sl) ? 0 : 1;
                // an earlier parent of this phase producing the same structure wins (`seen` order)
                for (int e = single_from; isnew && e < b; e++)
                    if (pinfo[e].flag == 0) {
                        uint64_t g1 = pinfo[e].h1, g2 = pinfo[e].h2;
                        if (g1 == 0) g1 = 1;
                        if (g2 == 0) g2 = 1;
                        if (g1 == h1 && g2 == h2) isnew = 0;
                    }
            }
            int tot, ex = block_exscan_flag<BS_NT>(isnew, sh, &tot);
            if (isnew) {
                const int ci2 = nchild + ex;
                if (ci2 < d.ch_cap) {
                    d.ch_parent[chb + ci2] = (uint16_t)b;
                    d.ch_combo[chb + ci2] = 0;
                    d.ch_dcal[chb + ci2] = pinfo[b].dcal0;
                    d.ch_h[2 * (chb + ci2)] = h1;
                    d.ch_h[2 * (chb + ci2) + 1] = h2;
                } else atomicOr(&d.c->overflow, OVF_SORT);
                seen_insert_at(stab, scap, h1, h2, free_sl);
            }
            if (b < nbeam && pinfo[b].flag == 0) { d.st[oldbeam[b]].cursor = 1; d.st[oldbeam[b]].total = pinfo[b].total; }
            nchild += tot; nb_branch += tot; scnt += tot;
            __syncthreads();
        }
    }
    STAMP(2);
    if (tid == 0) { d.seen_cnt[sq] = scnt; atomicAdd(&d.c->xstat[1][sq & (NSHARD - 1)].children, (unsigned long long)nchild); }
    if (nchild > d.ch_cap) nchild = d.ch_cap;

    // ---- new = children + beam, stable sort by energy, cut (rafft/rafft.py:206-210)
    const int N = nchild + nbeam;
    if (N > sort_cap) { if (tid == 0) atomicOr(&d.c->overflow, OVF_SORT); d.done[sq] = 1; return; }
    for (int i = tid; i < N; i += BS_NT) {
        unsigned long long key;
        if (i < nchild) key = ((unsigned long long)(uint32_t)(d.ch_dcal[chb + i] + 0x40000000) << 32) | (uint32_t)i;
        else key = ((unsigned long long)(uint32_t)(d.st[oldbeam[i - nchild]].dcal + 0x40000000) << 32) | (uint32_t)i;
        skey[i] = key;
    }
    __syncthreads();
    {
        // only the max_stack best survive: select them exactly (radix select), then sort just those
        const int K = N < d.B ? N : d.B;
        select_smallest_inplace<BS_NT>(skey, N, K, rl_cnt, sh);
        if (K <= RL_CAP) {
            // order the selected keys by counting (round 4): the rank of a key is the number of smaller ones among the K (they are
            // distinct), and it goes straight to its place - two barriers where the bitonic sort of 64 keys takes 21
            unsigned long long *outk = rl_off;              // (the region list of the product walk is dead by now; RL_CAP entries)
            for (int i = tid; i < K; i += BS_NT) {
                const unsigned long long ki = skey[i];
                int r = 0;
                for (int j = 0; j < K; j++) r += skey[j] < ki ? 1 : 0;
                outk[r] = ki;
            }
            __syncthreads();
            for (int i = tid; i < K; i += BS_NT) skey[i] = outk[i];
            __syncthreads();
        } else {
        int M = 2; while (M < K) M <<= 1;
        for (int i = K + tid; i < M; i += BS_NT) skey[i] = ~0ULL;
        __syncthreads();
        for (int k2 = 2; k2 <= M; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < M; i += BS_NT) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        unsigned long long a = skey[i], bb = skey[ixj];
                        bool up = (i & k2) == 0;
                        if (up ? a > bb : a < bb) { skey[i] = bb; skey[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }
    STAMP(3);
    const int nnew = N < d.B ? N : d.B;
    // children among the survivors
    int nsurv_child = 0;
    for (int base = 0; base < nnew; base += BS_NT) {
        int i = base + tid, f = 0;
        if (i < nnew) f = ((uint32_t)skey[i] < (uint32_t)nchild) ? 1 : 0;
        int tot, ex = block_exscan_flag<BS_NT>(f, sh, &tot);
        (void)ex;
        nsurv_child += tot;
        __syncthreads();
    }
    if (nsurv_child == 0) {   // same structures as before: fixed point (rafft/rafft.py:213-214)
        if (tid == 0) {
            d.done[sq] = 1;
            atomicAdd(&d.c->n_done, 1u);
            if (!d.traj) {
                unsigned long long r = atomicAdd(&d.c->trec_n, 1ULL);
                unsigned long long o = atomicAdd(&d.c->tsid_top, (unsigned long long)nbeam);
                if (r >= d.trec_cap || o + nbeam > d.tsid_cap) atomicOr(&d.c->overflow, OVF_TRAJ);
                else {
                    d.trec[r] = make_int4(sq, 0, nbeam, (int)o);
                    for (int i = 0; i < nbeam; i++) d.tsid[o + i] = oldbeam[i];
                }
            }
        }
        WS_END();
        return;
    }
    if (tid == 0) {
        unsigned long long sb = atomicAdd(&d.c->n_struct, (unsigned long long)nsurv_child);
        unsigned int mb = atomicAdd(&d.c->n_mat, (unsigned int)nsurv_child);
        atomicAdd(&d.c->xstat[1][sq & (NSHARD - 1)].struct_len, (unsigned long long)nsurv_child * (unsigned long long)d.seq_len[sq]);
        if (sb + nsurv_child > d.st_cap || mb + nsurv_child > d.mat_cap) { atomicOr(&d.c->overflow, OVF_STRUCT); sh[24] = -1; }
        else { sh[24] = (int)sb; sh[25] = (int)mb; }
    }
    __syncthreads();
    const int sbase = sh[24], mbase = sh[25];
    __syncthreads();
    if (sbase < 0) { d.done[sq] = 1; return; }
    int run = 0;
    const int Lsq = d.seq_len[sq];
    const unsigned long long soff_sq = (unsigned long long)d.seq_off[sq];
    for (int base = 0; base < nnew; base += BS_NT) {
        int i = base + tid, f = 0;
        uint32_t ord = 0;
        if (i < nnew) { ord = (uint32_t)skey[i]; f = (ord < (uint32_t)nchild) ? 1 : 0; }
        // the child's record: five loads issued together, under way while the ranks below are counted (interleaved with the stores they
        // feed they were three or four dependent round trips)
        int c_dcal = 0;
        unsigned int c_par = 0;
        unsigned long long c_combo = 0;
        ulonglong2 c_h = make_ulonglong2(0ULL, 0ULL);
        if (f) {
            const size_t c = chb + ord;
            c_dcal = d.ch_dcal[c]; c_par = d.ch_parent[c]; c_combo = d.ch_combo[c]; c_h = *(const ulonglong2 *)&d.ch_h[2 * c];
        }
        int tot, ex = block_exscan_flag<BS_NT>(f, sh, &tot);
        if (i < nnew) {
            if (f) {
                pin(c_dcal); pin(c_par); pin(c_combo); pin(c_h);
                const int sid = sbase + run + ex;
                const ParentInfo &pp_ = pinfo[c_par];
                StRec *sr = &d.st[sid];
                sr->seq = sq; sr->dcal = c_dcal; sr->h1 = c_h.x; sr->h2 = c_h.y; sr->parent = oldbeam[c_par]; sr->combo = c_combo;
                sr->cursor = 0; sr->total = 0; sr->nnodes = 0;
                MatRec mr;
                mr.sid = sid; mr.sq = sq; mr.L = Lsq; mr.dcal = c_dcal; mr.nprod = pp_.nprod; mr.pad = 0;
                mr.combo = c_combo; mr.prod = pp_.prod; mr.soff = soff_sq;
                d.mat[mbase + run + ex] = mr;
                beam[i] = sid;
            } else
                beam[i] = oldbeam[ord - nchild];
        }
        run += tot;
        __syncthreads();
    }
    if (tid == 0) d.beam_n[sq] = nnew;
    STAMP(4);
    if (prof) atomicAdd(&d.prof[5], 1ULL);
    WS_END();
#undef WS_END
#undef STAMP
}

// ------------------------------------------------------- materialize kernel

#define MAT_NT 64
// One wavefront per new beam member.  Child regions are spliced from the parent's
// regions: inner = positions/branches strictly inside the innermost stem pair, outer =
// the rest of the parent's loop with the whole stem as one new branch.
//
// One lane describes one productive region of the parent (chosen stem, the branch indices it cuts the
// loop at, sizes of the two child regions); the copies then run FLAT over all output elements of the
// tile (binary search element -> child region), so every load of the wavefront is independent and in
// flight at once instead of one dependent round trip per region.
struct MatDesc {
    unsigned long long srcpos, srcbr, cidx;      // cidx: the candidate (its two child slots are cslot[2 cidx], cslot[2 cidx + 1])
    int pn, mi, mj, nb, n, nbr, ci, cj, lo0, hi0, loo, hio, a0, b0, ao, bo, flags, win;   // flags: which children exist (1 inner, 2 outer); win: which of them THIS structure creates
    uint32_t newbr;        // the stem as a branch of the outer child: outermost pair, in the arena's (packed) form
    int nnod, npos_in, npos_out, nbr_in, nbr_out;
};
// (`cidx` comes from the parent's productive-region list: the candidate record and the region header are independent loads)
__device__ inline MatDesc mat_describe(const Dev &d, int pn, unsigned long long cidx)
{
    MatDesc m;
    m.pn = pn;
    const Cand cd = d.cand[cidx];
    // (the header as three 16-byte loads issued together: as single fields the compiler loaded `nbr` where it is first used - after the
    //  loads of the four stem positions below, whose round trip it then waited for before the arena allocations could be issued)
    const uint4 *hp = (const uint4 *)&d.nd[pn];
    const uint4 hq0 = hp[0], hq1 = hp[1], hq2 = hp[2];      // seq pdcal n ci | cj nbr ncand L | pos br
    m.n = (int)hq0.z; m.ci = (int)hq0.w; m.cj = (int)hq1.x; m.nbr = (int)hq1.y;
    m.srcpos = (unsigned long long)hq2.x | ((unsigned long long)hq2.y << 32); m.srcbr = (unsigned long long)hq2.z | ((unsigned long long)hq2.w << 32);
    m.cidx = cidx;
    m.mi = cd.mi; m.mj = cd.mj; m.nb = cd.nb;
    const uint16_t *pp = d.pos + m.srcpos;
    const int pm = d.pos_packed ? 0x0FFF : 0xFFFF;
    const uint32_t rao = pp[m.mi - m.nb + 1], rbo = pp[m.mj + m.nb - 1];
    m.a0 = pp[m.mi] & pm; m.b0 = pp[m.mj] & pm; m.ao = (int)rao & pm; m.bo = (int)rbo & pm;
    m.newbr = rao | (rbo << 16);          // (with Dev::pos_packed the base codes ride in bits 12-15 and 28-31)
    cd.get_cuts(m.lo0, m.hi0, m.loo, m.hio);      // where the stem cuts the branch list (found by expand_kernel)
    // (no branches: every header field is used right here, so all of them are loaded together - see above)
    const bool has_in = m.mj - m.mi > 1, has_out = m.mi - (m.nb - 1) > 0 || m.mj + m.nb < m.n;
    m.win = 0;
    m.flags = (has_in ? 1 : 0) | (has_out ? 2 : 0); m.nnod = (has_in ? 1 : 0) + (has_out ? 1 : 0);
    m.npos_in = has_in ? m.mj - m.mi - 1 : 0; m.nbr_in = has_in ? m.hi0 - m.lo0 : 0;
    m.npos_out = has_out ? (m.mi - m.nb + 1) + (m.n - (m.mj + m.nb)) : 0; m.nbr_out = has_out ? m.loo + 1 + (m.nbr - m.hio) : 0;
    return m;
}

// The flat copies of one tile of the materialize kernels: unpaired positions and branch helices of the regions created, pairs of
// the stems.  (Round 5: U elements per lane are located and LOADED before the first of them is stored - with one element per
// iteration every load was waited for before its store and the next load issued after it: a dependent HBM round trip per 16 (64)
// elements, five or six per structure on the benchmark set, a dozen and more on long sequences.)
// `l`: my lane in the team, STR lanes; descriptor kk of the tile sits at index kb + kk of the k_* arrays.
template <int STR, int U>
__device__ __forceinline__ void mat_copy_tile(const Dev &d, int l, int kb, int kt, const int *ps, const int *bs, const int *ns,
                                              const unsigned long long *k_srcpos, const unsigned long long *k_srcbr, const int *k_mi, const int *k_mj,
                                              const int *k_nb, const int *k_lo0, const int *k_loo, const int *k_hio, const int *k_newbr,
                                              int tp, int tbr, int ts, unsigned long long pdst, unsigned long long bdst, unsigned long long sdst, int pmask)
{
    for (int f0 = l; f0 < tp; f0 += STR * U) {           // unpaired positions of the regions created here
        uint32_t v[U];          // (32-bit: two 16-bit values packed into one register are a wait after every load)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int f = f0 + u * STR;
            v[u] = 0;
            if (f < tp) {
                int lo = 0, hi = 2 * kt - 1;             // last slot starting at or before f (empty slots share starts)
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ps[mid] <= f) lo = mid; else hi = mid - 1; }
                const int kk = kb + (lo >> 1), off = f - ps[lo];
                const uint16_t *pp = d.pos + k_srcpos[kk];
                int src;
                if (!(lo & 1)) src = k_mi[kk] + 1 + off;
                else { const int left = k_mi[kk] - k_nb[kk] + 1; src = off < left ? off : k_mj[kk] + k_nb[kk] + (off - left); }
                v[u] = pp[src];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int f = f0 + u * STR; if (f < tp) d.pos[pdst + f] = (uint16_t)v[u]; }
    }
    for (int f0 = l; f0 < tbr; f0 += STR * U) {          // their branch helices
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int f = f0 + u * STR;
            v[u] = 0;
            if (f < tbr) {
                int lo = 0, hi = 2 * kt - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (bs[mid] <= f) lo = mid; else hi = mid - 1; }
                const int kk = kb + (lo >> 1), off = f - bs[lo];
                const uint32_t *bb = d.br + k_srcbr[kk];
                if (!(lo & 1)) v[u] = bb[k_lo0[kk] + off];
                else {
                    const int loo = k_loo[kk];
                    v[u] = off < loo ? bb[off] : off == loo ? (uint32_t)k_newbr[kk] : bb[k_hio[kk] + (off - loo - 1)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int f = f0 + u * STR; if (f < tbr) d.br[bdst + f] = v[u]; }
    }
    // the pairs of the stems (rafft/rafft.py:97,127-128 marks them in the parent's dot-bracket row; here the row is implicit)
    for (int f0 = l; f0 < ts; f0 += STR * U) {
        uint32_t va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int f = f0 + u * STR;
            va[u] = 0; vb[u] = 0;
            if (f < ts) {
                int lo = 0, hi = kt - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ns[mid] <= f) lo = mid; else hi = mid - 1; }
                const int t = f - ns[lo];
                const uint16_t *pp = d.pos + k_srcpos[kb + lo];
                va[u] = pp[k_mi[kb + lo] - t]; vb[u] = pp[k_mj[kb + lo] + t];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { const int f = f0 + u * STR; if (f < ts) d.sp[sdst + f] = (uint32_t)(va[u] & pmask) | ((uint32_t)(vb[u] & pmask) << 16); }
    }
}

#ifndef RAFFT_MAT_WAVES
#define RAFFT_MAT_WAVES 1
#endif
// (dynamic LDS: the productive-region lists only - a structure is stored as the pairs it adds to its parent's, no dot-bracket row is
//  staged or written here; a latency-bound kernel of one-wavefront workgroups lives on the number of them a CU holds)
template <bool PROD>      // (PROD: the phase stamps of RAFFT_TRACE=3 compiled out - see expand_kernel)
__global__ __launch_bounds__(MAT_NT, RAFFT_MAT_WAVES) void materialize_kernel(Dev d)
{
    extern __shared__ __align__(16) uint8_t mat_dyn[];
    // dynamic LDS: the productive-region lists (d.max_prod entries each)
    unsigned long long *prod_off = (unsigned long long *)mat_dyn;
    int *prod_node = (int *)(prod_off + d.max_prod);
    int *prod_cnt = prod_node + d.max_prod;
    int *sel = prod_cnt + d.max_prod;
    // per-tile descriptors (one lane per productive region) and the flat-copy prefix sums (two slots per region)
    __shared__ unsigned long long k_srcpos[64], k_srcbr[64];
    __shared__ int k_mi[64], k_mj[64], k_nb[64], k_lo0[64], k_loo[64], k_hio[64], k_newbr[64];
    __shared__ int ps[129], bs[129], ns[65];
    __shared__ unsigned long long sh64[5];
    __shared__ int shi[8];
    const int tid = threadIdx.x;
    // diagnostic phase stamps (RAFFT_TRACE=3) of every 64th workgroup, kept in the slots of class 0
    const bool mprof = !PROD && d.prof_e != nullptr && tid == 0 && (blockIdx.x & 63) == 0;
    unsigned long long mt = mprof ? clock64() : 0, macc[7] = {0, 0, 0, 0, 0, 0, 0};
#define MSTAMP(k) do { if (mprof) { const unsigned long long tn_ = clock64(); macc[k] += tn_ - mt; mt = tn_; } } while (0)
    const MatRec rec = d.mat[blockIdx.x];              // written by the beam step: no chain of look-ups to get started
    const int sid = rec.sid, sq = rec.sq, L = rec.L, my_dcal = rec.dcal;
    const uint64_t soff = rec.soff;
    const int pmask = d.pos_packed ? 0x0FFF : 0xFFFF;
    int mprod = rec.nprod;
    if (mprod > d.max_prod) mprod = d.max_prod;
    {
        const ProdEnt *pl = d.prod + rec.prod;             // the parent's productive regions (beam_step prepass)
        for (int k = tid; k < mprod; k += MAT_NT) { const ProdEnt pe = pl[k]; prod_node[k] = pe.node; prod_cnt[k] = (int)pe.cnt; prod_off[k] = pe.off; sel[k] = 0; }
    }
    __syncthreads();
    if (tid == 0) {      // digits of the combo, last region fastest; high digits of a small index stay 0
        unsigned long long idx = rec.combo;
        for (int k = mprod - 1; k >= 0 && idx; k--) {
            const unsigned int c = (unsigned int)prod_cnt[k];
            if (idx < (1ULL << 24)) {
                const unsigned int v = (unsigned int)idx;
                unsigned int q = (unsigned int)((float)v * __frcp_rn((float)c));       // off by one at most
                int r = (int)(v - q * c);
                if (r < 0) { q--; r += (int)c; } else if (r >= (int)c) { q++; r -= (int)c; }
                sel[k] = r; idx = q;
            } else { const unsigned long long q = idx / c; sel[k] = (int)(idx - q * c); idx = q; }
        }
    }
    __syncthreads();
    MSTAMP(0);   // header, productive-region list, combo digits

    // pass 1: sizes, and who creates what.  A child region is a function of (parent region, candidate, side) alone
    // (rafft/rafft.py:127-152, rafft/utils.py:141-152): the beam member whose compare-and-swap finds the slot empty creates it, everybody
    // else - the other members of this step that picked the same stem, and every later step - only notes the slot number in its node
    // list (the next beam step reads the region id out of the slot, once this kernel and dedupe_kernel are done: nobody reads a slot's
    // value in here).  Without memoization (min_nrj != 0: a region's filter depends on its parent's energy) every member creates its own.
    // (a single tile - the usual case - keeps its descriptors in registers for pass 2; with several the claims ride in sel[])
    const int TILE = d.mat_tile;          // 64; smaller only in tests (several tiles per structure)
    const bool one_tile = mprod <= TILE;
    const bool memo = d.memo != 0;
    MatDesc md;
    md.flags = 0; md.win = 0; md.nnod = 0; md.npos_in = md.npos_out = md.nbr_in = md.nbr_out = 0; md.nb = 0; md.cidx = 0;
    int tot_nodes = 0, tot_new = 0, tot_pos = 0, tot_br = 0, tot_sp = 0;
    for (int base = 0; base < mprod; base += TILE) {
        const int k = base + tid;
        int nnod = 0, nnew = 0, npos = 0, nbrr = 0, nsp = 0;
        if (k < mprod && tid < TILE) {
            // the claim of both child slots of the chosen candidate: ONE returning atomic, issued before anything else is loaded (its
            // round trip runs beside those of the region header, the candidate and the positions).  A slot word is inner | outer << 32;
            // bit 31 of a half says "claimed", and whoever finds it clear has claimed that half.  (Claiming the half of a child that
            // does not exist - an empty inside, nothing left outside - is harmless: nobody ever looks at it.)
            const unsigned long long cidx = prod_off[k] + (unsigned long long)sel[k];
            unsigned long long old = 0;
            if (memo) old = atomicOr(&d.cslot[cidx], 0x8000000080000000ULL);
            md = mat_describe(d, prod_node[k], cidx);
            int win = md.flags;
            if (memo) win &= ((old >> 31) & 1ULL ? 0 : 1) | ((old >> 63) & 1ULL ? 0 : 2);
            md.win = win;
            if (!one_tile) sel[k] |= win << 28;
            nnod = md.nnod; nnew = (win & 1) + (win >> 1); nsp = md.nb;
            npos = ((win & 1) ? md.npos_in : 0) + ((win & 2) ? md.npos_out : 0);
            nbrr = ((win & 1) ? md.nbr_in : 0) + ((win & 2) ? md.nbr_out : 0);
        }
        for (int o = 32; o > 0; o >>= 1) {
            nnod += __shfl_xor(nnod, o, 64); nnew += __shfl_xor(nnew, o, 64); npos += __shfl_xor(npos, o, 64); nbrr += __shfl_xor(nbrr, o, 64);
            nsp += __shfl_xor(nsp, o, 64);
        }
        tot_nodes += nnod; tot_new += nnew; tot_pos += npos; tot_br += nbrr; tot_sp += nsp;
    }
    MSTAMP(1);   // pass 1
    if (tid < 5) {
        // bump allocation from one of NSHARD sub-arenas (spreads the same-address atomics); one lane per arena
        const int shd = blockIdx.x & (NSHARD - 1);
        unsigned long long *ctr = tid == 0 ? &d.c->node[shd].v : tid == 1 ? &d.c->pos[shd].v : tid == 2 ? &d.c->sp[shd].v : tid == 3 ? &d.c->br[shd].v : &d.c->nlist[shd].v;
        const unsigned long long want = tid == 0 ? (unsigned long long)tot_new : tid == 1 ? (unsigned long long)tot_pos
                                      : tid == 2 ? (unsigned long long)tot_sp : tid == 3 ? (unsigned long long)tot_br : (unsigned long long)tot_nodes;
        const unsigned long long cap = tid == 0 || tid == 4 ? d.nd_shard_cap : tid == 1 ? d.pos_shard_cap : tid == 2 ? d.sp_shard_cap : d.br_shard_cap;
        const unsigned long long b0 = want ? atomicAdd(ctr, want) : 0ULL;
        const bool bad = b0 + want > cap;
        if (bad) atomicOr(&d.c->overflow, tid == 0 || tid == 4 ? OVF_NODE : tid == 1 ? OVF_POS : tid == 2 ? OVF_SP : OVF_BR);
        const unsigned long long origin = tid == 0 || tid == 4 ? d.nd_base : tid == 1 ? d.pos_base : 0ULL;
        sh64[tid] = origin + (unsigned long long)shd * cap + b0;
        const unsigned long long anybad = __ballot(bad);
        if (tid == 0) shi[0] = anybad ? 0 : 1;
    }
    __syncthreads();
    if (!shi[0]) { if (tid == 0) { d.st[sid].nnodes = 0; d.st[sid].node0 = 0; d.st[sid].sp = 0; d.st[sid].nsp = 0; } return; }
    const unsigned long long nbase = sh64[0], pbase = sh64[1], sbase = sh64[2], bbase = sh64[3], lbase = sh64[4];
    MSTAMP(2);   // allocation

    // pass 2: per tile: descriptors -> LDS, prefix sums, node-list entries, records and flat copies of the regions created here
    int run_nodes = 0, run_new = 0, run_pos = 0, run_br = 0, run_sp = 0;
    for (int base = 0; base < mprod; base += TILE) {
        const int k = base + tid;
        const int kt = mprod - base < TILE ? mprod - base : TILE;
        if (!one_tile) {
            md.flags = 0; md.win = 0; md.nnod = 0; md.npos_in = md.npos_out = md.nbr_in = md.nbr_out = 0; md.nb = 0;
            if (k < mprod && tid < TILE) { md = mat_describe(d, prod_node[k], prod_off[k] + (unsigned long long)(sel[k] & 0x0FFFFFFF)); md.win = (sel[k] >> 28) & 3; }
        }
        const bool act = k < mprod && tid < TILE;
        const int cp_in = act && (md.win & 1) ? md.npos_in : 0, cp_out = act && (md.win & 2) ? md.npos_out : 0;
        const int cb_in = act && (md.win & 1) ? md.nbr_in : 0, cb_out = act && (md.win & 2) ? md.nbr_out : 0;
        // inclusive scans over the tile: node-list entries, regions created, their pos and branch elements, stem pairs
        int xn = act ? md.nnod : 0, xw = act ? (md.win & 1) + (md.win >> 1) : 0, xp = cp_in + cp_out, xb = cb_in + cb_out, xs = act ? md.nb : 0;
        const int vn = xn, vw = xw, vp = xp, vb = xb, vs = xs;
        for (int o = 1; o < 64; o <<= 1) {
            const int yn = __shfl_up(xn, o, 64), yw = __shfl_up(xw, o, 64), yp = __shfl_up(xp, o, 64), yb = __shfl_up(xb, o, 64), ys = __shfl_up(xs, o, 64);
            if (tid >= o) { xn += yn; xw += yw; xp += yp; xb += yb; xs += ys; }
        }
        const int tn = __shfl(xn, 63, 64), tw = __shfl(xw, 63, 64), tp = __shfl(xp, 63, 64), tb = __shfl(xb, 63, 64), ts = __shfl(xs, 63, 64);
        const int p0 = xp - vp, b0 = xb - vb;          // exclusive
        ps[2 * tid] = p0; ps[2 * tid + 1] = p0 + cp_in;
        bs[2 * tid] = b0; bs[2 * tid + 1] = b0 + cb_in;
        ns[tid] = xs - vs;
        if (tid == 0) { ps[128] = tp; bs[128] = tb; ns[64] = ts; }
        if (act) {
            k_srcpos[tid] = md.srcpos; k_srcbr[tid] = md.srcbr;
            k_mi[tid] = md.mi; k_mj[tid] = md.mj; k_nb[tid] = md.nb; k_lo0[tid] = md.lo0; k_loo[tid] = md.loo; k_hio[tid] = md.hio;
            k_newbr[tid] = (int)md.newbr;
            // region records (rafft/utils.py:141-152) of the children created here, and the node list (rafft/rafft.py:187-190): inner, then outer
            int nid = (int)(nbase + run_new + (xw - vw));
            unsigned long long le = lbase + run_nodes + (xn - vn);
            const unsigned long long poff = pbase + run_pos + p0, boff = bbase + run_br + b0;
            const int slot0 = (int)(2 * md.cidx);
            if (md.flags & 1) {
                if (md.win & 1) {
                    d.nd[nid].seq = sq; d.nd[nid].pdcal = my_dcal; d.nd[nid].pos = poff; d.nd[nid].n = md.npos_in;
                    d.nd[nid].L = L; d.nd[nid].soff = soff;
                    d.nd[nid].ci = md.a0; d.nd[nid].cj = md.b0; d.nd[nid].br = boff; d.nd[nid].nbr = md.nbr_in;
                    d.nd[nid].ncand = -1; d.nd[nid].cand = 0;
                    if (memo) { d.nd_slot[nid] = (uint32_t)slot0; ((uint32_t *)d.cslot)[slot0] = (uint32_t)(nid + 1) | 0x80000000u; }
                    d.nlist[le] = memo ? -(slot0 + 1) : nid;
                    nid++;
                } else d.nlist[le] = -(slot0 + 1);
                le++;
            }
            if (md.flags & 2) {
                if (md.win & 2) {
                    d.nd[nid].seq = sq; d.nd[nid].pdcal = my_dcal; d.nd[nid].pos = poff + cp_in; d.nd[nid].n = md.npos_out;
                    d.nd[nid].L = L; d.nd[nid].soff = soff;
                    d.nd[nid].ci = md.ci; d.nd[nid].cj = md.cj; d.nd[nid].br = boff + cb_in; d.nd[nid].nbr = md.nbr_out;
                    d.nd[nid].ncand = -1; d.nd[nid].cand = 0;
                    if (memo) { d.nd_slot[nid] = (uint32_t)(slot0 + 1); ((uint32_t *)d.cslot)[slot0 + 1] = (uint32_t)(nid + 1) | 0x80000000u; }
                    d.nlist[le] = memo ? -(slot0 + 2) : nid;
                } else d.nlist[le] = -(slot0 + 2);
            }
        }
        __syncthreads();
        MSTAMP(4);   // pass 2 descriptors + records
        mat_copy_tile<MAT_NT, 4>(d, tid, 0, kt, ps, bs, ns, k_srcpos, k_srcbr, k_mi, k_mj, k_nb, k_lo0, k_loo, k_hio, k_newbr, tp, tb, ts,
                                 pbase + run_pos, bbase + run_br, sbase + run_sp, pmask);
        run_nodes += tn; run_new += tw; run_pos += tp; run_br += tb; run_sp += ts;
        __syncthreads();
        MSTAMP(5);   // region copies
    }
    if (tid == 0) { d.st[sid].node0 = (int)lbase; d.st[sid].nnodes = tot_nodes; d.st[sid].sp = sbase; d.st[sid].nsp = tot_sp; }
    MSTAMP(6);   // structure record
    if (mprof) for (int k = 0; k < 7; k++) atomicAdd(&d.prof_e[k], macc[k]);
#undef MSTAMP
}

// The same with TEAMS of 16 lanes: four new beam members per wavefront (round 4).  materialize_kernel is a chain of four dependent
// round trips per structure (record -> parent's lists -> slot claim / region headers / candidates -> stem positions and
// arena allocation -> writes) in which a lane stands for one productive region of the parent - three to five of them on the benchmark
// set - so a wavefront per structure keeps 60 lanes idle through the chain, and what a CU holds of such wavefronts (20, by registers)
// bounds the structures in flight.  With four structures per wavefront the same CU holds four times as many.  Used when the
// productive-region lists are the short ones (max_prod <= MAT4_PROD; host: Wave::after_beam); identical results -
// the arenas are bump allocated, so only the PLACES of records and lists differ from the one-structure form.
#define MAT4_TL 16
#define MAT4_TEAMS (64 / MAT4_TL)
#define MAT4_PROD 64
__global__ __launch_bounds__(64, RAFFT_MAT_WAVES) void materialize_team_kernel(Dev d, int n_mat)
{
    const int tid = threadIdx.x, team = tid / MAT4_TL, tl = tid % MAT4_TL;
    // LDS per team: the productive-region lists (MAT4_PROD entries each)
    __shared__ unsigned long long prod_off_[MAT4_TEAMS][MAT4_PROD];
    __shared__ int prod_node_[MAT4_TEAMS][MAT4_PROD], prod_cnt_[MAT4_TEAMS][MAT4_PROD], sel_[MAT4_TEAMS][MAT4_PROD];
    unsigned long long *prod_off = prod_off_[team];
    int *prod_node = prod_node_[team], *prod_cnt = prod_cnt_[team], *sel = sel_[team];
    // per-tile descriptors (one lane per productive region; index = lane of the wavefront) and the flat-copy prefix sums of every team
    __shared__ unsigned long long k_srcpos[64], k_srcbr[64];
    __shared__ int k_mi[64], k_mj[64], k_nb[64], k_lo0[64], k_loo[64], k_hio[64], k_newbr[64];
    __shared__ int ps_[MAT4_TEAMS][2 * MAT4_TL + 1], bs_[MAT4_TEAMS][2 * MAT4_TL + 1], ns_[MAT4_TEAMS][MAT4_TL + 1];
    __shared__ unsigned long long sh64_[MAT4_TEAMS][5];
    int *ps = ps_[team], *bs = bs_[team], *ns = ns_[team];
    const int tb = team * MAT4_TL;                       // first lane of my team
    const unsigned long long tmask = ((1ULL << MAT4_TL) - 1ULL) << tb;
    // (round 5) n_mat < 0: the count is the device's own (the beam step's counter) and the grid whatever the host guessed - it issues
    // this kernel before it has read the step's counters back; the workgroups stride over the list
    if (d.c->overflow) return;                           // (see expand_kernel)
    if (n_mat < 0) n_mat = (int)d.c->n_mat;
    for (int mat_i0 = blockIdx.x * MAT4_TEAMS; mat_i0 < n_mat; mat_i0 += gridDim.x * MAT4_TEAMS) {
    const int mat_i = mat_i0 + team;
    const bool live = mat_i < n_mat;
    MatRec rec;
    rec.sid = 0; rec.sq = 0; rec.L = 0; rec.dcal = 0; rec.nprod = 0; rec.combo = 0; rec.prod = 0; rec.soff = 0;
    if (live) rec = d.mat[mat_i];
    const int sid = rec.sid, sq = rec.sq, L = rec.L, my_dcal = rec.dcal;
    const uint64_t soff = rec.soff;
    const int pmask = d.pos_packed ? 0x0FFF : 0xFFFF;
    int mprod = rec.nprod;
    if (mprod > MAT4_PROD) mprod = MAT4_PROD;
    {
        const ProdEnt *pl = d.prod + rec.prod;             // the parent's productive regions (beam_step prepass)
        for (int k = tl; k < mprod; k += MAT4_TL) { const ProdEnt pe = pl[k]; prod_node[k] = pe.node; prod_cnt[k] = (int)pe.cnt; prod_off[k] = pe.off; sel[k] = 0; }
    }
    wave_sync();
    if (tl == 0) {       // digits of the combo, last region fastest; high digits of a small index stay 0
        unsigned long long idx = rec.combo;
        for (int k = mprod - 1; k >= 0 && idx; k--) {
            const unsigned int c = (unsigned int)prod_cnt[k];
            if (idx < (1ULL << 24)) {
                const unsigned int v = (unsigned int)idx;
                unsigned int q = (unsigned int)((float)v * __frcp_rn((float)c));       // off by one at most
                int r = (int)(v - q * c);
                if (r < 0) { q--; r += (int)c; } else if (r >= (int)c) { q++; r -= (int)c; }
                sel[k] = r; idx = q;
            } else { const unsigned long long q = idx / c; sel[k] = (int)(idx - q * c); idx = q; }
        }
    }
    wave_sync();
    // pass 1: sizes and slot claims (see materialize_kernel)
    const int TILE = d.mat_tile < MAT4_TL ? d.mat_tile : MAT4_TL;
    const bool one_tile = mprod <= TILE;
    const bool memo = d.memo != 0;
    MatDesc md;
    md.flags = 0; md.win = 0; md.nnod = 0; md.npos_in = md.npos_out = md.nbr_in = md.nbr_out = 0; md.nb = 0; md.cidx = 0;
    int tot_nodes = 0, tot_new = 0, tot_pos = 0, tot_br = 0, tot_sp = 0;
    for (int base = 0; base < mprod; base += TILE) {
        const int k = base + tl;
        int nnod = 0, nnew = 0, npos = 0, nbrr = 0, nsp = 0;
        if (k < mprod && tl < TILE) {
            const unsigned long long cidx = prod_off[k] + (unsigned long long)sel[k];
            unsigned long long old = 0;
            if (memo) old = atomicOr(&d.cslot[cidx], 0x8000000080000000ULL);
            md = mat_describe(d, prod_node[k], cidx);
            int win = md.flags;
            if (memo) win &= ((old >> 31) & 1ULL ? 0 : 1) | ((old >> 63) & 1ULL ? 0 : 2);
            md.win = win;
            if (!one_tile) sel[k] |= win << 28;
            nnod = md.nnod; nnew = (win & 1) + (win >> 1); nsp = md.nb;
            npos = ((win & 1) ? md.npos_in : 0) + ((win & 2) ? md.npos_out : 0);
            nbrr = ((win & 1) ? md.nbr_in : 0) + ((win & 2) ? md.nbr_out : 0);
        }
        // (round 5: sums over the team - a row of 16 lanes - by DPP row scans and one read of the row's last lane each, instead of four
        //  rounds of five __shfl_xor through the LDS crossbar)
        static_assert(MAT4_TL == 16, "a team is one DPP row");
        nnod = __shfl(row16_incl_scan(nnod), MAT4_TL - 1, MAT4_TL); nnew = __shfl(row16_incl_scan(nnew), MAT4_TL - 1, MAT4_TL);
        npos = __shfl(row16_incl_scan(npos), MAT4_TL - 1, MAT4_TL); nbrr = __shfl(row16_incl_scan(nbrr), MAT4_TL - 1, MAT4_TL);
        nsp = __shfl(row16_incl_scan(nsp), MAT4_TL - 1, MAT4_TL);
        tot_nodes += nnod; tot_new += nnew; tot_pos += npos; tot_br += nbrr; tot_sp += nsp;
    }
    bool ok = live;
    if (tl < 5 && live) {
        // bump allocation from one of NSHARD sub-arenas; one lane per arena
        const int shd = mat_i & (NSHARD - 1);
        unsigned long long *ctr = tl == 0 ? &d.c->node[shd].v : tl == 1 ? &d.c->pos[shd].v : tl == 2 ? &d.c->sp[shd].v : tl == 3 ? &d.c->br[shd].v : &d.c->nlist[shd].v;
        const unsigned long long want = tl == 0 ? (unsigned long long)tot_new : tl == 1 ? (unsigned long long)tot_pos
                                      : tl == 2 ? (unsigned long long)tot_sp : tl == 3 ? (unsigned long long)tot_br : (unsigned long long)tot_nodes;
        const unsigned long long cap = tl == 0 || tl == 4 ? d.nd_shard_cap : tl == 1 ? d.pos_shard_cap : tl == 2 ? d.sp_shard_cap : d.br_shard_cap;
        const unsigned long long b0 = want ? atomicAdd(ctr, want) : 0ULL;
        const bool bad = b0 + want > cap;
        if (bad) atomicOr(&d.c->overflow, tl == 0 || tl == 4 ? OVF_NODE : tl == 1 ? OVF_POS : tl == 2 ? OVF_SP : OVF_BR);
        const unsigned long long origin = tl == 0 || tl == 4 ? d.nd_base : tl == 1 ? d.pos_base : 0ULL;
        sh64_[team][tl] = origin + (unsigned long long)shd * cap + b0;
        ok = !bad;
    }
    // (every lane of the team learns whether all five allocations fit)
    ok = ((__ballot(!ok) & tmask) == 0ULL) && live;
    wave_sync();
    if (!ok) { if (tl == 0 && live) { d.st[sid].nnodes = 0; d.st[sid].node0 = 0; d.st[sid].sp = 0; d.st[sid].nsp = 0; } }
    const unsigned long long nbase = sh64_[team][0], pbase = sh64_[team][1], sbase = sh64_[team][2], bbase = sh64_[team][3], lbase = sh64_[team][4];

    // pass 2: per tile: descriptors -> LDS, prefix sums, node-list entries, records and flat copies of the regions created here
    int run_nodes = 0, run_new = 0, run_pos = 0, run_br = 0, run_sp = 0;
    const int mp2 = ok ? mprod : 0;
    for (int base = 0; base < mp2; base += TILE) {
        const int k = base + tl;
        const int kt = mp2 - base < TILE ? mp2 - base : TILE;
        if (!one_tile) {
            md.flags = 0; md.win = 0; md.nnod = 0; md.npos_in = md.npos_out = md.nbr_in = md.nbr_out = 0; md.nb = 0;
            if (k < mp2 && tl < TILE) { md = mat_describe(d, prod_node[k], prod_off[k] + (unsigned long long)(sel[k] & 0x0FFFFFFF)); md.win = (sel[k] >> 28) & 3; }
        }
        const bool act = k < mp2 && tl < TILE;
        const int cp_in = act && (md.win & 1) ? md.npos_in : 0, cp_out = act && (md.win & 2) ? md.npos_out : 0;
        const int cb_in = act && (md.win & 1) ? md.nbr_in : 0, cb_out = act && (md.win & 2) ? md.nbr_out : 0;
        int xn = act ? md.nnod : 0, xw = act ? (md.win & 1) + (md.win >> 1) : 0, xp = cp_in + cp_out, xb = cb_in + cb_out, xs = act ? md.nb : 0;
        const int vn = xn, vw = xw, vp = xp, vb = xb, vs = xs;
        xn = row16_incl_scan(xn); xw = row16_incl_scan(xw); xp = row16_incl_scan(xp); xb = row16_incl_scan(xb); xs = row16_incl_scan(xs);
        const int tn = __shfl(xn, MAT4_TL - 1, MAT4_TL), tw = __shfl(xw, MAT4_TL - 1, MAT4_TL), tp = __shfl(xp, MAT4_TL - 1, MAT4_TL),
                  tbr = __shfl(xb, MAT4_TL - 1, MAT4_TL), ts = __shfl(xs, MAT4_TL - 1, MAT4_TL);
        const int p0 = xp - vp, b0 = xb - vb;          // exclusive
        ps[2 * tl] = p0; ps[2 * tl + 1] = p0 + cp_in;
        bs[2 * tl] = b0; bs[2 * tl + 1] = b0 + cb_in;
        ns[tl] = xs - vs;
        if (tl == 0) { ps[2 * MAT4_TL] = tp; bs[2 * MAT4_TL] = tbr; ns[MAT4_TL] = ts; }
        if (act) {
            k_srcpos[tid] = md.srcpos; k_srcbr[tid] = md.srcbr;
            k_mi[tid] = md.mi; k_mj[tid] = md.mj; k_nb[tid] = md.nb; k_lo0[tid] = md.lo0; k_loo[tid] = md.loo; k_hio[tid] = md.hio;
            k_newbr[tid] = (int)md.newbr;
            int nid = (int)(nbase + run_new + (xw - vw));
            unsigned long long le = lbase + run_nodes + (xn - vn);
            const unsigned long long poff = pbase + run_pos + p0, boff = bbase + run_br + b0;
            const int slot0 = (int)(2 * md.cidx);
            if (md.flags & 1) {
                if (md.win & 1) {
                    d.nd[nid].seq = sq; d.nd[nid].pdcal = my_dcal; d.nd[nid].pos = poff; d.nd[nid].n = md.npos_in;
                    d.nd[nid].L = L; d.nd[nid].soff = soff;
                    d.nd[nid].ci = md.a0; d.nd[nid].cj = md.b0; d.nd[nid].br = boff; d.nd[nid].nbr = md.nbr_in;
                    d.nd[nid].ncand = -1; d.nd[nid].cand = 0;
                    if (memo) { d.nd_slot[nid] = (uint32_t)slot0; ((uint32_t *)d.cslot)[slot0] = (uint32_t)(nid + 1) | 0x80000000u; }
                    d.nlist[le] = memo ? -(slot0 + 1) : nid;
                    nid++;
                } else d.nlist[le] = -(slot0 + 1);
                le++;
            }
            if (md.flags & 2) {
                if (md.win & 2) {
                    d.nd[nid].seq = sq; d.nd[nid].pdcal = my_dcal; d.nd[nid].pos = poff + cp_in; d.nd[nid].n = md.npos_out;
                    d.nd[nid].L = L; d.nd[nid].soff = soff;
                    d.nd[nid].ci = md.ci; d.nd[nid].cj = md.cj; d.nd[nid].br = boff + cb_in; d.nd[nid].nbr = md.nbr_out;
                    d.nd[nid].ncand = -1; d.nd[nid].cand = 0;
                    if (memo) { d.nd_slot[nid] = (uint32_t)(slot0 + 1); ((uint32_t *)d.cslot)[slot0 + 1] = (uint32_t)(nid + 1) | 0x80000000u; }
                    d.nlist[le] = memo ? -(slot0 + 2) : nid;
                } else d.nlist[le] = -(slot0 + 2);
            }
        }
        wave_sync();
        // (descriptor kk of my team sits at lane tb + kk)
        mat_copy_tile<MAT4_TL, 4>(d, tl, tb, kt, ps, bs, ns, k_srcpos, k_srcbr, k_mi, k_mj, k_nb, k_lo0, k_loo, k_hio, k_newbr, tp, tbr, ts,
                                  pbase + run_pos, bbase + run_br, sbase + run_sp, pmask);
        run_nodes += tn; run_new += tw; run_pos += tp; run_br += tbr; run_sp += ts;
        wave_sync();
    }
    if (ok && tl == 0) { d.st[sid].node0 = (int)lbase; d.st[sid].nnodes = tot_nodes; d.st[sid].sp = sbase; d.st[sid].nsp = tot_sp; }
    wave_sync();
    }
}

// ------------------------------------------------------------ dedupe kernel

// a region header as four 16-byte words, loaded together: seq pdcal n ci | cj nbr ncand L | pos br | cand soff
struct NodeWords { uint4 q0, q1, q2; };
__device__ __forceinline__ NodeWords load_node_words(const Dev &d, int nid)
{
    const uint4 *hp = (const uint4 *)&d.nd[nid];
    NodeWords w;
    w.q0 = hp[0]; w.q1 = hp[1]; w.q2 = hp[2];
    pin(w.q0); pin(w.q1); pin(w.q2);
    return w;
}
// (round 5: the other region's header in one round trip and the branch lists four entries at a time - field by field, each
//  comparison behind the one before, this was seven dependent round trips)
__device__ inline bool same_loop(const Dev &d, const NodeWords &a, int b)
{
    const NodeWords o = load_node_words(d, b);
    // seq, n, ci | cj, nbr
    if (a.q0.x != o.q0.x || a.q0.z != o.q0.z || a.q0.w != o.q0.w || a.q1.x != o.q1.x || a.q1.y != o.q1.y) return false;
    const uint32_t *x = d.br + ((unsigned long long)a.q2.z | ((unsigned long long)a.q2.w << 32));
    const uint32_t *y = d.br + ((unsigned long long)o.q2.z | ((unsigned long long)o.q2.w << 32));
    const int k = (int)a.q1.y;
    for (int i = 0; i < k; i += 4) {
        unsigned int xa[4], ya[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { xa[u] = i + u < k ? x[i + u] : 0u; ya[u] = i + u < k ? y[i + u] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) { pin(xa[u]); pin(ya[u]); }
        if (xa[0] != ya[0] || xa[1] != ya[1] || xa[2] != ya[2] || xa[3] != ya[3]) return false;
    }
    return true;
}

// One thread per region created in this step (the new node ids are the ranges the
// materialize kernel bumped in each allocation shard since the last snapshot).  The first
// region to claim a loop key becomes canonical and goes to the expand work list; later
// identical loops alias it.  Work-list appends are aggregated per wavefront.
#ifndef DEDUPE_NT
#define DEDUPE_NT 512         // (256 / 512 / 1024 measured with eight batches in flight: 282 / 285 / 279 k sequences/s)
#endif
__global__ __launch_bounds__(DEDUPE_NT) void dedupe_kernel(Dev d)
{
    __shared__ unsigned int pre[NSHARD + 1];
    __shared__ unsigned int prev[NSHARD];
    __shared__ unsigned int wcnt[DEDUPE_NT / 64][NCLS], wbase[DEDUPE_NT / 64][NCLS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // an arena overflowed while materializing: some region records of this step were never written.
    // Nothing may be read from them; the host sees the flag at its next read-back and regrows.
    if (d.c->overflow) return;
    if (tid < NSHARD) {
        prev[tid] = (unsigned int)d.c->node_prev[tid].v;
        pre[tid + 1] = (unsigned int)(d.c->node[tid].v - d.c->node_prev[tid].v);
    }
    if (tid == 0) pre[0] = 0;
    __syncthreads();
    if (tid == 0) for (int i = 1; i <= NSHARD; i++) pre[i] += pre[i - 1];
    __syncthreads();
    const unsigned int total = pre[NSHARD];
    unsigned long long aliases = 0;
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int f0 = blockIdx.x * blockDim.x; f0 < total; f0 += stride) {
        const unsigned int f = f0 + tid;
        int cls = -1, nid = 0;
        if (f < total) {
            int lo = 0, hi = NSHARD;             // shard with pre[lo] <= f < pre[lo+1]
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (pre[mid] <= f) lo = mid; else hi = mid; }
            nid = (int)(d.nd_base + (unsigned long long)lo * d.nd_shard_cap + prev[lo] + (f - pre[lo]));
            int canon = nid;
            // the header once, in one round trip (round 5: as single fields it was loaded in three trips here and AGAIN field by field
            // after the table look-up - the compiler cannot keep a loaded value across the compare-and-swap and the stores between)
            const NodeWords hw = load_node_words(d, nid);
            const int h_seq = (int)hw.q0.x, h_n = (int)hw.q0.z, h_ci = (int)hw.q0.w, h_cj = (int)hw.q1.x, h_nbr = (int)hw.q1.y, h_L = (int)hw.q1.w;
            if (d.memo) {
                const uint32_t *bb = d.br + ((unsigned long long)hw.q2.z | ((unsigned long long)hw.q2.w << 32));
                const int nbr = h_nbr;
                uint64_t h = mix64(((uint64_t)(uint32_t)h_seq << 32) ^ ((uint64_t)(uint32_t)(h_ci + 1) << 16) ^ (uint32_t)h_cj);
                for (int t = 0; t < nbr; t += 4) {           // (four branch helices per round trip)
                    unsigned int bv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) bv[u] = t + u < nbr ? bb[t + u] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) pin(bv[u]);
#pragma unroll
                    for (int u = 0; u < 4; u++) if (t + u < nbr) h += mix64((uint64_t)bv[u] ^ 0x5bd1e9955bd1e995ULL);
                }
                const unsigned long long tag = (h >> 32) | 0x80000000ULL;
                const uint64_t mask = d.looptab_cap - 1;
                uint64_t sl = h & mask;
                for (unsigned probe = 0;; probe++) {
                    unsigned long long old = atomicCAS(&d.looptab[sl], 0ULL, (tag << 32) | (unsigned long long)(nid + 1));
                    if (old == 0) break;
                    if ((old >> 32) == tag) {
                        int other = (int)(old & 0xffffffffULL) - 1;
                        if (same_loop(d, hw, other)) { canon = other; break; }
                    }
                    sl = (sl + 1) & mask;
                    if (probe > d.looptab_cap) { atomicOr(&d.c->overflow, OVF_LOOPTAB); break; }
                }
            }
            if (canon == nid) {
                // a stem needs two unpaired positions: a lone position (bulge remnant) has no candidates
                const int n = h_n;
                if (n < 2) d.nd[nid].ncand = 0;
                else {
                    // (sequences beyond 4096 nt keep out of the one-wavefront class whatever the span: see expand_kernel's Sl)
                    const int Ls = h_L, span = (h_ci < 0 || Ls > LDS_SEQ) ? Ls : h_cj + 1 - h_ci;
                    cls = node_class(n, span, h_nbr, d.merge_cls, d.cls1_P, d.cls1_br, d.K, d.sm_n4, d.sm_n5);
                }
            }
            else { ((uint32_t *)d.cslot)[d.nd_slot[nid]] = (uint32_t)(canon + 1) | 0x80000000u; aliases++; }      // (the loop is known - reached along another path: the slot points at it)
        }
        // work-list appends, aggregated over the WORKGROUP: one atomic per class and pass (per wavefront they were
        // 4096 x 4-6 returning atomics on one cache line per pass - the kernel's whole duration)
        unsigned long long mybal = 0;
        for (int c = 0; c < NCLS; c++) {
            const unsigned long long bal = __ballot(cls == c);
            if (cls == c) mybal = bal;
            if (lane == 0) wcnt[wv][c] = (unsigned int)__popcll(bal);
        }
        __syncthreads();
        if (tid < NCLS) {
            unsigned int tot = 0;
            for (int w = 0; w < DEDUPE_NT / 64; w++) tot += wcnt[w][tid];
            unsigned int b = tot ? atomicAdd(&d.c->n_work[tid].v, tot) : 0u;
            for (int w = 0; w < DEDUPE_NT / 64; w++) { wbase[w][tid] = b; b += wcnt[w][tid]; }
        }
        __syncthreads();
        if (cls >= 0) {
            const unsigned int w = wbase[wv][cls] + (unsigned int)__popcll(mybal & ((1ULL << lane) - 1));
            if (w < d.work_cap) d.work[cls][w] = nid; else atomicOr(&d.c->overflow, OVF_WORK);
        }
    }
    for (int o = 32; o > 0; o >>= 1) aliases += __shfl_xor(aliases, o, 64);
    if (lane == 0 && aliases) atomicAdd(&d.c->xstat[0][blockIdx.x & (NSHARD - 1)].alias, aliases);
}

// ------------------------------------------------------------- init kernel

// The inputs of a wave, from its pinned staging chunk into the device buffers: up to eight segments copied by one kernel that reads
// the host memory itself (hipHostMalloc memory is mapped into the device's address space).  Round 5: as hipMemcpyAsync calls the first
// of these uploads now and then kept the scheduler thread - i.e. every wave in flight - for 12-19 ms (six bench runs in ten on one
// box, the runtime's copy path waiting for something of its own); a kernel launch never waits.
struct StageIn { const uint32_t *src[8]; uint32_t *dst[8]; unsigned long long words[8]; int n; };
__global__ __launch_bounds__(256) void stage_in_kernel(StageIn si)
{
    for (int k = 0; k < si.n; k++) {
        const uint32_t *src = si.src[k];
        uint32_t *dst = si.dst[k];
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < si.words[k]; i += (unsigned long long)gridDim.x * 256) dst[i] = src[i];
    }
}

__global__ void init_roots_kernel(Dev d)
{
    const int sq = blockIdx.x, tid = threadIdx.x;
    const int L = d.seq_len[sq];
    // structure sq / node sq are the unfolded structure and its single region (rafft.py:224-231)
    const unsigned long long off = (unsigned long long)d.seq_off[sq];
    for (int x = tid; x < L; x += blockDim.x) d.pos[off + x] = (uint16_t)(d.pos_packed ? x | (d.codes[off + x] << 12) : x);

    if (tid == 0) {
        d.st[sq].seq = sq; d.st[sq].dcal = 0; d.st[sq].h1 = 0; d.st[sq].h2 = 0;
        d.st[sq].sp = 0; d.st[sq].nsp = 0; d.st[sq].node0 = sq; d.st[sq].nnodes = L > 0 ? 1 : 0; d.st[sq].cursor = 0; d.st[sq].total = 0;
        d.st[sq].parent = -1; d.st[sq].combo = 0;
        d.nd[sq].seq = sq; d.nd[sq].pdcal = 0; d.nd[sq].pos = off; d.nd[sq].n = L; d.nd[sq].ci = -1; d.nd[sq].cj = L;
        d.nd[sq].L = L; d.nd[sq].soff = off;
        d.nd[sq].br = 0; d.nd[sq].nbr = 0; d.nlist[sq] = sq;
        d.nd[sq].ncand = -1; d.nd[sq].cand = 0;
        d.beam[(size_t)sq * d.B] = sq; d.beam_n[sq] = 1; d.nsteps[sq] = 0;
        d.done[sq] = L > 0 ? 0 : 1;
        d.seen_cnt[sq] = 0;       // (seen_off / seen_cap: uploaded by the host - tables sized from the lengths, zeroed by its memset)
        if (L > 0) {
            int cls = node_class(L, L, 0, 0, d.cls1_P, d.cls1_br);
            unsigned int w = atomicAdd(&d.c->n_work[cls].v, 1u);
            d.work[cls][w] = sq;
        }
    }
}

// ----------------------------------------------------------- output kernel

// One record = the beam of one sequence at one step (all of them with traj, the last one otherwise); its rows
// go out back to back, `off` bytes into the result buffer, row numbers from `row0`.
struct OutRec { long long off; int row0, w, cnt, L; };
// The dot-bracket rows are made HERE: a structure is stored as the pairs it added to its parent's (materialize kernels), so a row is
// the unfolded one (rafft.py:224-231) with the stems of the whole lineage marked (rafft/rafft.py:97,127-128) - built in LDS (dynamic,
// the longest sequence of the wave) and written out once.
__global__ void output_kernel(Dev d, int nrows, int nrec, const OutRec *recs, char *out_db, int *out_dcal)
{
    extern __shared__ __align__(16) uint8_t out_row[];
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        int lo = 0, hi = nrec - 1;                      // record holding row r
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (recs[mid].row0 <= r) lo = mid; else hi = mid - 1; }
        const OutRec rc = recs[lo];
        const int k = r - rc.row0;
        const int sid = d.tsid[rc.w + k];
        const int L = rc.L;
        for (int x = threadIdx.x; x < L; x += blockDim.x) out_row[x] = '.';
        __syncthreads();
        // the lineage, child to root (the unfolded structure has parent -1 and no pairs).  The next ancestor's row is asked for before
        // this one's pairs are read: one dependent round trip per generation instead of two (a row of the benchmark set has 5-25)
        int s = sid, par = -1, np = 0;
        unsigned long long spo = 0;
        if (s >= 0) { par = d.st[s].parent; np = d.st[s].nsp; spo = d.st[s].sp; }
        while (s >= 0) {
            const int s2 = par;
            int par2 = -1, np2 = 0;
            unsigned long long spo2 = 0;
            if (s2 >= 0) { par2 = d.st[s2].parent; np2 = d.st[s2].nsp; spo2 = d.st[s2].sp; }
            const uint32_t *pl = d.sp + spo;
            for (int x = threadIdx.x; x < np; x += blockDim.x) { const uint32_t u = pl[x]; out_row[u & 0xFFFFu] = '('; out_row[u >> 16] = ')'; }
            s = s2; par = par2; np = np2; spo = spo2;
        }
        __syncthreads();
        char *o = out_db + rc.off + (long long)k * (L + 1);
        for (int x = threadIdx.x; x < L; x += blockDim.x) o[x] = (char)out_row[x];
        if (threadIdx.x == 0) { o[L] = 0; out_dcal[r] = d.st[sid].dcal; }
        __syncthreads();
    }
}

// ------------------------------------------------------------- eval kernel

// one wavefront per structure: sum of loop energies (rafft/utils.py:135-138)
__global__ __launch_bounds__(64) void eval_kernel(const EnergyTables *ET, int n, const uint8_t *codes, const int16_t *pts,
                                                  const long long *off, const int *len, int *out, int *status, int *guessed)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= n) return;
    const int L = len[s];
    const uint8_t *S = codes + off[s];
    PlainView pv{pts + off[s]};
    const SmallT *T = &ET->s;
    const BigT *B = &ET->b;
    int e = 0, bad = 0;
    if (lane == 0) e += loop_energy(T, B, S, L, pv, -1, L, &bad);
    for (int i = lane; i < L; i += 64) {
        int j = pv(i);
        if (j > i) e += loop_energy(T, B, S, L, pv, i, j, &bad);
    }
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_xor(e, o, 64); bad |= __shfl_xor(bad, o, 64); }
    if (lane == 0) { out[s] = e; status[s] = (bad & 1) ? 8 : 0; if (guessed) guessed[s] = (bad >> 1) & 1; }
}
