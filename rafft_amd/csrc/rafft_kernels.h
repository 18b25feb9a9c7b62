// rafft_kernels.h - device state of the RAFFT fold engine (gfx950).
//
// Data layout in HBM (one batch = many independent sequences; SoA tables + bump
// arenas, monotonic inside a batch so no kernel ever frees):
//   codes[sum L]        uint8 base codes (N=0 A=1 C=2 G=3 U=4)
//   struct table st[]   one 128-byte row per beam survivor: energy (dcal), 128-bit pair-set hash,
//                       its stem pairs, node range, product cursor, lineage
//   sp arena            the pairs a survivor ADDED to its parent's (uint32 i | j << 16): a structure is its lineage's
//                       lists; dot-bracket rows only exist in the result buffers (output_kernel walks the lineage)
//   node table nd[]     one 64-byte row per unpaired region.  A region is exactly one loop of
//                       the structure: closing pair (ci,cj) (ci<0: exterior loop), the
//                       ordered unpaired positions `pos` and the ordered branch helices
//                       `br` hanging in that loop.  Its candidate stems depend on
//                       nothing else, so identical loops met in different structures
//                       are created and expanded ONCE:
//   child slots cslot[] one word per candidate, two halves: the region INSIDE / OUTSIDE the stem (rafft/utils.py:141-152).  A child
//                       region is a function of (parent region, candidate, side) alone, so the first beam member that
//                       picks the candidate claims the slot (one compare-and-swap) and creates the region; everybody
//                       else only notes the slot in its node list.  Loops reached along different paths (stems formed
//                       in another order) still meet in the `loop table`, which sees created regions only.
//   node lists nlist[]  the regions of a structure, in the reference's node_list order (rafft/rafft.py:187-190): -(slot + 1) as
//                       materialize_kernel writes them (the beam step reads the region id out of the slot when it first
//                       visits the structure - by then dedupe_kernel has run), region ids without memoization and for roots
//   pos arena           uint16 root positions of every node, ascending
//   br arena            uint32 (p | q<<16) outermost pair of every branch, ascending
//   cand arena          32-byte stem candidates, dE-sorted per canonical node
//   seen arena          per-sequence open-addressing sets of 128-bit structure hashes
//   children[S][cap]    per-step accepted children (parent, combo, dcal, hash)
// Names follow the reference: Node/Structure (rafft/utils.py:24-39), beam =
// glob_tree, trajectory = glob_traj (rafft/rafft.py:156-216).
#pragma once
#include "rafft_device.h"

struct alignas(16) Cand {
    int32_t ddcal;
    uint16_t mi, mj, nb, cut_hi;      // stem: innermost pair (mi,mj) in region coordinates, nb stacked pairs
    uint32_t cut_lo;                  // cut_hi:cut_lo = where the stem cuts the region's branch list, 4 x 12 bits
    uint64_t h1, h2;                  // 128-bit hash of the stem's pairs
    // number of branches starting before the innermost 5' / innermost 3' / outermost 5' / outermost 3' position
    __host__ __device__ void set_cuts(int lo0, int hi0, int loo, int hio)
    {
        const unsigned long long v = (unsigned long long)lo0 | ((unsigned long long)hi0 << 12) | ((unsigned long long)loo << 24) | ((unsigned long long)hio << 36);
        cut_lo = (uint32_t)v; cut_hi = (uint16_t)(v >> 32);
    }
    __host__ __device__ void get_cuts(int &lo0, int &hi0, int &loo, int &hio) const
    {
        const unsigned long long v = (unsigned long long)cut_lo | ((unsigned long long)cut_hi << 32);
        lo0 = (int)(v & 4095); hi0 = (int)((v >> 12) & 4095); loo = (int)((v >> 24) & 4095); hio = (int)((v >> 36) & 4095);
    }
};
static_assert(sizeof(Cand) == 32, "Cand must be 32 bytes");

#define NSHARD 64
#ifndef SEEN0
#define SEEN0 8192      // initial slots of a sequence's `seen` set (grows x2 by rehash)
#endif
#define NCLS 6          // expand size classes: 0-3 the general kernel (NGEN), 4-5 the small-region kernel (teams of 16 / 32 lanes)
#define NGEN 4
#define PROF_E 96      // RAFFT_TRACE=3: 64-bit diagnostic slots per expand class (Dev::prof_e)
struct ShardCtr { unsigned long long v; unsigned long long pad[7]; };   // one 64-byte line each

// structure row: one beam survivor (see the file header)
struct alignas(128) StRec {
    int32_t seq, dcal, node0, nnodes, parent, nprod;   // sequence, energy, region range, lineage, productive regions
    int32_t c0d, nsp;                                  // energy of its combo 0 (kept for resumed product walks); pairs it added to its parent's
    uint64_t h1, h2;                                   // 128-bit pair-set hash
    uint64_t sp, cursor, combo, total, prod;           // its added pairs (sp arena), product cursor, combo it came from, product size, productive-region list
    uint64_t c0h1, c0h2;                               // hash of its combo 0
    uint64_t pad2[3];
};
static_assert(sizeof(StRec) == 128, "StRec is two cache lines");

// region row: one loop of one structure (see the file header)
struct alignas(64) NodeRec {
    int32_t seq, pdcal, n, ci, cj, nbr, ncand, L;      // (L, soff: the sequence's length and offset into the base codes, so that
    uint64_t pos, br, cand, soff;                      //  expanding a region needs no second look-up keyed on `seq`)
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be one cache line");

struct ProdEnt { uint32_t cnt; int32_t node; uint64_t off; };   // one productive region of a structure
// one new beam member to materialize: everything materialize_kernel needs to start, in one 48-byte read
struct alignas(16) MatRec { int32_t sid, sq, L, dcal, nprod, pad; uint64_t combo, prod, soff; };     // (soff: the sequence's offset into the base codes - one look-up less in the materialize kernels)

struct Counters {
    // hot part: read back by the host once per folding step.  Every counter that the kernels of a step add to with a RETURNING atomic
    // has a 64-byte line of its own: same-line atomics are served one after the other, 11.4 ns each chip-wide
    // (tools/micro/atomic_spacing.hip) - a beam step of 11 500 sequences does 23 000 of them on n_struct and n_mat alone.
    unsigned int n_mat;            // structures to materialize (filled by beam_step_kernel)
    unsigned int overflow;         // bit mask of which arena overflowed
    unsigned int n_done;
    unsigned int max_nprod;        // largest number of productive regions seen in one structure
    unsigned int pad0_[12];
    unsigned long long n_struct, pad1_[7];
    unsigned long long trec_n, tsid_top, pad2_[6];       // (one finishing sequence adds to both)
    unsigned long long seen_top, pad3_[7];
    struct WorkCtr { unsigned int v; unsigned int pad[15]; } n_work[NCLS];     // expand work items per size class (filled by dedupe_kernel)
    // statistics
    unsigned long long n_expand, sum_n, sum_lags, n_children, sum_struct_len, n_alias, sum_nbr;
    unsigned long long cls_items[NCLS], cls_sum_n[NCLS], cls_sum_lags[NCLS];   // per size class
    // sharded bump pointers of the arenas filled by materialize / expand
    ShardCtr node[NSHARD], pos[NSHARD], br[NSHARD], sp[NSHARD], cand[NSHARD], node_prev[NSHARD], prod[NSHARD];
    ShardCtr nlist[NSHARD];      // node-list entries (one per region of a structure; `node` counts the regions CREATED)
    // statistics the kernels add to once per wavefront / workgroup: one 64-byte line per (size class, shard), summed by the
    // host at the end of the wave.  (As single counters they were a same-address atomic storm at the end of every expand
    // launch - 3000-4000 wavefronts x 7 atomics on one line - a fixed 150-300 us per launch.)
    // Work cursors of the persistent expand kernels: the work list of a class is cut in chunks, chunk c belongs to shard
    // c % NSHARD, and wcur[cls][s] counts the chunks of shard s handed out (see fetch_chunk).  ONE cursor per class was a
    // same-address returning atomic per wavefront and round: ~12 ns each, served one after the other - 50 us for the 4096
    // wavefronts of a launch to learn that there is nothing (left) to do, and as much for every round they start together.
    ShardCtr wcur[NCLS][NSHARD];
    unsigned long long wdone[NCLS];           // bit s: the last chunk of shard s has been claimed
    unsigned long long wdone_pad[8 - NCLS];
    struct StatLine { unsigned long long items, n, lags, nbr, alias, children, struct_len, evals, guessed, kept_guessed, pad[6]; } xstat[NCLS][NSHARD];
};

enum { OVF_STRUCT = 1, OVF_NODE = 2, OVF_POS = 4, OVF_SP = 8, OVF_CAND = 16, OVF_SEEN = 32,
       OVF_TRAJ = 64, OVF_WORK = 128, OVF_PROD = 256, OVF_SORT = 512, OVF_BR = 1024, OVF_LOOPTAB = 2048, OVF_PRODLIST = 4096 };

struct DebugOut {       // kernel-level seam (rafft_expand_node); null in production
    int *n_ranked, *lag, *nb, *mi, *mj, *ddcal, *kept;
    double *corval, *score;
};

struct Dev {
    const EnergyTables *T;
    const float2 *tw;            // exp(-2 pi i m / 8192), m < 4096
    int S;
    const uint8_t *codes;
    const int *seq_off, *seq_len;
    int K, B, max_branch, min_hp, traj, memo, force_fft, rl_cap, mat_tile, merge_cls;
    double *big_keyv; size_t big_stride;   // lag values of regions too big for LDS: one slice of `big_stride` doubles per workgroup
    int max_prod;                // productive regions per structure that materialize_kernel's LDS lists hold
    int fetch_bulk, taper_pct;   // work chunks: regions (rounds of the small-region kernel) per claim in the bulk of a list; percent of the list handed out that way
    int direct_n;                // wide classes: multi-word popcount correlation up to this region size, FFT beyond (RAFFT_DIRECT_N)
    int sm_n4, sm_n5;            // small-region classes: regions of up to sm_n4 positions go to class 4 (teams of 16 lanes), up to
    int cand_slab;                           // candidate slots an expand wavefront reserves at a time (one returning atomic each)
                                 // sm_n5 to class 5 (teams of 32); 0 = class unused (see node_class)
    int cls1_P, cls1_br;         // limits of the one-wavefront expand class (FFT size, branches): they set its LDS per wavefront
    int c3_switch;               // class 3: up to this many regions in a step the FFT plan works, beyond it the FFT-free kernel (launch_expand_cls)
    double min_nrj, gc, au, gu;
    int *beam, *beam_n, *done, *nsteps;
    // children of the current step
    int ch_cap;
    uint16_t *ch_parent; uint64_t *ch_combo; int *ch_dcal; uint64_t *ch_h;
    // seen sets
    uint64_t *seen; uint64_t seen_cap_total;
    uint64_t *seen_off; uint32_t *seen_cap, *seen_cnt;
    // structures
    uint32_t st_cap;
    StRec *st;                   // one 128-byte record per structure (two cache lines)
    ProdEnt *prod; uint64_t prod_shard_cap;
    // nodes
    uint32_t nd_cap;
    uint64_t nd_base, nd_shard_cap, pos_base, pos_shard_cap, br_shard_cap, sp_shard_cap, cand_shard_cap;
    NodeRec *nd;                 // one 64-byte record per region (one cache line: header reads and writes are one transaction)
    int *nlist;                  // node lists of the structures (st.node0, st.nnodes): region id, or -(child slot + 1) (see the file header)
    unsigned long long *cslot;   // child slots, one word per candidate: inner | outer << 32; a half is 0 (nobody has asked yet), bit 31 alone
                                 // (claimed: being created in this step) or bit 31 | (region id + 1)
    uint32_t *nd_slot;           // the slot a created region hangs in (dedupe_kernel points it at the canonical region when the loop is known already)
    // loop table: open addressing, word = (hash tag << 32) | (node id + 1)
    unsigned long long *looptab; uint64_t looptab_cap;   // power of two
    // arenas
    uint16_t *pos; uint64_t pos_cap;
    int pos_packed;              // no sequence beyond 4096 nt: an entry of `pos` is position | base code << 12 (the code rides along:
                                 // expanding a region reads it with the position instead of through a dependent second load)
    uint32_t *br; uint64_t br_cap;
    uint32_t *sp; uint64_t sp_cap;       // stem pairs added by every structure (i | j << 16)
    Cand *cand; uint64_t cand_cap;
    // trajectory records: (seq, step, count, offset into tsid)
    int4 *trec; uint32_t trec_cap;
    int *tsid; uint64_t tsid_cap;
    // work lists
    int *work[NCLS]; uint32_t work_cap;
    MatRec *mat; uint32_t mat_cap;
    Counters *c;
    DebugOut dbg;
    unsigned long long *prof; int prof_seq;   // diagnostic stamps of beam_step_kernel (RAFFT_TRACE=3)
    unsigned long long *prof_e;               // RAFFT_TRACE=3: expand_kernel phase cycles, regions and cycles by region size, phase cycles by region size [NCLS][PROF_E]
    unsigned long long *prof_ws;              // RAFFT_TRACE=3: per sequence [cycles, chunks, max cycles of one step]
    int rep;                     // profiling only (RAFFT_REP env): bit k doubles phase k of expand_kernel
};

// expand-kernel size classes: 1 small (FFT size P <= 512: one wavefront per region), 2 medium (P <= 2048: 256 threads),
// 3 large (P <= 8192: 512 threads, one workgroup per CU), 0 regions beyond that (n > 4096; 512 threads, no FFT)
#define CLS0_P 4096        // (only sizes the LDS scratch of class 0: 16 bytes x CLS0_P)
#define CLS1_P 512
#define CLS01_L 1280
#define CLS1_BR 256
#define CLS2_P 2048
#define MAX_P 8192
#define MAX_BR 1024        // branches of a loop, classes 1-3
#define BIG_BR 4095        // ... of the class for the biggest regions (the 12-bit cut points of a candidate)
#define BIG_N 32768        // positions of a region of that class = RAFFT_MAX_LEN
#define LDS_SEQ 4096       // sequences up to this length have the bases of a loop staged in LDS by classes 2 and 3
#define MAX_PROD 64        // productive regions per structure (sequences up to LDS_SEQ); a wave that meets more is folded again with the long lists
#define MAX_PROD_LONG 1024 // ... for longer sequences
#define RL_CAP 1024        // beam_step_kernel: regions with >= 2 candidates of all beam members, kept in LDS

__host__ __device__ inline int next_pow2_ge(int x) { int p = 2; while (p < x) p <<= 1; return p; }
// `span`: the stretch of the sequence the loop lies in (closing pair to closing pair; the whole sequence for the exterior
// loop) - the bases the expand kernel stages in LDS.  The one-wavefront class has room for CLS01_L of them, so a small
// loop of a LONG sequence (the inside of a hairpin of a 16S rRNA) still is one wavefront's work, not a workgroup's.
// Small regions (classes 4 and 5, expand_small_kernel): every lag is searched (2n-1 <= nb_mode: nothing to rank), the whole
// region sits in one team of 16 / 32 lanes, its loop spans at most SM_SPAN bases and has at most one branch per lane.
#define SM_SPAN 448
__host__ __device__ inline int node_class(int n, int span, int nbr, int merge_cls = 0, int cls1_P = CLS1_P, int cls1_br = CLS1_BR,
                                          int K = 0, int sm_n4 = 0, int sm_n5 = 0)
{
    if (merge_cls == 0 && n >= 2 && n <= sm_n5 && 2 * n - 1 <= K && span <= SM_SPAN) {
        if (n <= sm_n4 && nbr <= 16) return 4;
        if (sm_n5 > sm_n4 && nbr <= 32) return 5;       // (equal limits: class 5 is off - issue_step never launches it - and the region takes the general kernel)
    }
    // few regions in this step (the tail of a batch): all of them go to one kernel, the widest one that is
    // configured - one launch and one region per workgroup instead of three nearly empty kernels in a row
    int P = next_pow2_ge(2 * n - 1);
    // class 0: regions whose FFT buffers (P > 8192) or branch list do not fit the LDS plan of the other classes - exact
    // direct correlation on bit masks, lag values in HBM.  Only sequences longer than 4096 nt can have such regions.
    if (P > MAX_P || nbr > MAX_BR) return 0;
    if (merge_cls == 3) return 3;
    if (merge_cls == 2) return P <= CLS2_P ? 2 : 3;
    if (span <= CLS01_L && P <= cls1_P && nbr <= cls1_br) return 1;
    if (P <= CLS2_P) return 2;
    return 3;
}

// bit masks of a region (build_masks): 5 forward strings (A, C, G, U, contiguity with the previous position) and 5 reversed
// ones, W = ceil(n / 64) words each
#define MASK_F_WORDS 5
#define MASK_WORDS 10
// LDS layout of the expand kernel (bytes).  Region A is time-shared between the FFT
// buffers and the sort keys; region B holds the loop itself.
struct ExpandLds {
    int offA, szA, off_pos, off_code, off_p2, off_S, off_br, off_rk, off_nb, off_mi, off_mj, off_dd, off_keep, off_w, off_misc,
        per_team,             // bytes of one team (wavefront or workgroup); a workgroup of `wpb` teams holds wpb of them ...
        off_tab, off_tw,      // ... followed by ONE shared area: energy tables, twiddles (offsets inside that area)
        total;
};
// `nofft`: the class never runs the FFT (every region is correlated by popcounts on bit masks): region A only holds the lag
// values (8 P bytes) and what follows them in turn - bit masks, select histogram (at 9 P), branch prefix sums, sort keys.
// `code_lds`: the region's base codes are staged in LDS (nmax bytes) - every class but the one for regions beyond 4096 positions,
// which reads them through the positions from HBM/L2 (its 32 768 positions alone take 64 KiB).
__host__ __device__ inline ExpandLds expand_lds(int Pmax, int Lmax, int nmax, int brmax, int Kmax, bool tab_lds, int wpb = 1, bool nofft = false, int nt = 512,
                                                bool code_lds = true)
{
    ExpandLds l;
    auto al = [](int x) { return (x + 15) & ~15; };
    l.offA = 0;
    l.szA = al(16 * Pmax);
    if (nofft) {
        const int need1 = 8 * Pmax + 8 * MASK_WORDS * ((nmax + 63) / 64) + 8 + (nt > 64 && 24 * nt > 1152 ? 24 * nt : 1152) /* select histogram, later the partial results of chunked diagonals (wide classes) */, need2 = 8 * Pmax + 10 * (brmax + 1) + 16, need3 = 8 * Pmax + 8 * Kmax + 64;
        l.szA = al(need1 > need2 ? (need1 > need3 ? need1 : need3) : (need2 > need3 ? need2 : need3));
    }
    int o = l.szA;
    l.off_pos = o; o += al(2 * nmax + 2);
    l.off_code = o; if (code_lds) o += al(nmax);
    l.off_p2 = o; if (code_lds) o += al(4 * ((nmax + 15) / 16 + 1));       // (round 5) the bases again, 2 bits per position (stem_stack_windows)
    l.off_S = o; o += al(Lmax + 8);
    l.off_br = o; o += al(4 * brmax);
    // per-lag arrays (ranked lags, window_slide results, dE, kept list): 14 bytes per searched lag.  In the class with the
    // 128-KiB FFT buffers (P = 8192) a big nb_mode does not fit beside them - but they are only written once the FFTs are
    // done, and then the second half of region A is free except for its first 24 KiB (bit masks 5 KiB at 8 P, branch prefix
    // sums 10 KiB at 8 P, select histogram 1.2 KiB at 9 P; chunked window_slide partials only exist for nb_mode <= 256):
    // they go there, and nb_mode up to 2047 works for sequences of any length up to 4096 nt.
    const int lag_bytes = al(2 * next_pow2_ge(Kmax)) + 4 * al(2 * Kmax) + al(4 * Kmax);
    const bool lag_in_A = Pmax == MAX_P && Kmax > 256 && 8 * Pmax + 24 * 1024 + lag_bytes <= l.szA;
    int oa = lag_in_A ? 8 * Pmax + 24 * 1024 : 0;
    int &q = lag_in_A ? oa : o;
    l.off_rk = q; q += al(2 * next_pow2_ge(Kmax));
    l.off_nb = q; q += al(2 * Kmax);
    l.off_mi = q; q += al(2 * Kmax);
    l.off_mj = q; q += al(2 * Kmax);
    l.off_dd = q; q += al(4 * Kmax);
    l.off_keep = q; q += al(2 * Kmax);
    l.off_w = o; if (!nofft) o += al(25 * 8);      // (pair weights by base codes: only the cell-by-cell window_slide reads them - never in a class without FFT buffers)
    l.off_misc = o; o += 128;
    l.per_team = o;
    int sh = 0;
    l.off_tab = sh; if (tab_lds) sh += al((int)sizeof(SmallT));
    l.off_tw = sh; if (tab_lds && Pmax <= CLS2_P && !nofft) sh += al(8 * (Pmax / 2));    // (the twiddles of the largest class stay in L2)
    l.total = wpb * l.per_team + sh;
    return l;
}
