// rafft_kernels.h - device state + HIP kernels of the RAFFT fold engine (gfx950).
//
// Data layout in HBM (one batch = many independent sequences, SoA arenas, all
// bump-allocated and monotonic inside a batch so no kernel ever frees):
//   codes[sum L]            uint8 base codes (N=0 A=1 C=2 G=3 U=4)
//   struct table st_*[]     one row per beam survivor: energy (dcal), 128-bit pair-set
//                           hash, pair-table offset, node range, product cursor
//   pt arena                int16 pair table per structure (partner or -1), L entries
//   node table nd_*[]       one row per unpaired region (= one loop of the structure):
//                           pos offset/len, closing pair (ci,cj), candidate list
//   pos arena               uint16 root positions of every node, ascending
//   cand arena              32-byte stem candidates, dE-sorted per node
//   seen arena              per-sequence open-addressing sets of 128-bit hashes
//   children[S][cap]        per-step accepted children (parent, combo, dcal, hash)
// Names follow the reference: Node/Structure (rafft/utils.py:24-39), beam =
// glob_tree, trajectory = glob_traj (rafft/rafft.py:156-216).
#pragma once
#include "rafft_device.h"

struct alignas(16) Cand {
    int32_t ddcal;
    uint16_t mi, mj, nb, pad;
    uint32_t pad2;
    uint64_t h1, h2;
};
static_assert(sizeof(Cand) == 32, "Cand must be 32 bytes");

struct Counters {
    unsigned long long n_struct, n_node, pos_top, pt_top, cand_top, seen_top, trec_n, tsid_top;
    unsigned int n_work[3];
    unsigned int n_mat;
    unsigned int overflow;      // bit mask of which arena overflowed
    unsigned int n_done;
    // statistics
    unsigned long long n_expand, sum_n, sum_lags, n_children, sum_struct_len, sum_span;
};

enum { OVF_STRUCT = 1, OVF_NODE = 2, OVF_POS = 4, OVF_PT = 8, OVF_CAND = 16, OVF_SEEN = 32,
       OVF_TRAJ = 64, OVF_WORK = 128, OVF_PROD = 256, OVF_SORT = 512 };

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
    int K, B, max_branch, min_hp, traj;
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
    int *st_seq, *st_dcal, *st_node0, *st_nnodes, *st_parent;
    uint64_t *st_h, *st_pt, *st_cursor, *st_combo;
    // nodes
    uint32_t nd_cap;
    int *nd_sid, *nd_n, *nd_ci, *nd_cj, *nd_ncand;
    uint64_t *nd_pos, *nd_cand;
    // arenas
    uint16_t *pos; uint64_t pos_cap;
    int16_t *pt; uint64_t pt_cap;
    Cand *cand; uint64_t cand_cap;
    // trajectory records: (seq, step, count, offset into tsid)
    int4 *trec; uint32_t trec_cap;
    int *tsid; uint64_t tsid_cap;
    // work lists
    int *work[3]; uint32_t work_cap;
    int *mat; uint32_t mat_cap;
    Counters *c;
    DebugOut dbg;
    int rep;                     // profiling only (RAFFT_REP env): bit k doubles phase k of expand_kernel
};

// expand-kernel size classes: {max P, max span, threads}
#define CLS0_P 512
#define CLS0_SPAN 1280
#define CLS1_P 2048
#define MAX_P 8192
#define MAX_PROD 1024

__host__ __device__ inline int next_pow2_ge(int x) { int p = 2; while (p < x) p <<= 1; return p; }
__host__ __device__ inline int node_class(int n, int span)
{
    int P = next_pow2_ge(2 * n - 1);
    if (P <= CLS0_P && span <= CLS0_SPAN) return 0;
    if (P <= CLS1_P) return 1;
    return 2;
}

// LDS layout of the expand kernel (bytes).  Region A is time-shared between the FFT
// buffers, the sort keys and the energy window; region B holds the node itself.
struct ExpandLds {
    int offA, szA, off_pos, off_code, off_rk, off_nb, off_mi, off_mj, off_dd, off_keep, off_w, off_misc, total;
};
__host__ __device__ inline ExpandLds expand_lds(int Pmax, int span_max, int nmax, int Kmax)
{
    ExpandLds l;
    auto al = [](int x) { return (x + 15) & ~15; };
    int szE = al(span_max) + 2 * al(2 * span_max);
    l.offA = 0;
    l.szA = al(16 * Pmax > szE ? 16 * Pmax : szE);
    int o = l.szA;
    l.off_pos = o; o += al(2 * nmax);
    l.off_code = o; o += al(nmax);
    l.off_rk = o; o += al(2 * Kmax);
    l.off_nb = o; o += al(2 * Kmax);
    l.off_mi = o; o += al(2 * Kmax);
    l.off_mj = o; o += al(2 * Kmax);
    l.off_dd = o; o += al(4 * Kmax);
    l.off_keep = o; o += al(2 * Kmax);
    l.off_w = o; o += al(25 * 8);
    l.off_misc = o; o += 64;
    l.total = o;
    return l;
}
