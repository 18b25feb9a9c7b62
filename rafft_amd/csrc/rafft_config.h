// rafft_config.h - every environment switch of libraffthip.so in ONE place (round 5).
//
// The library has no configuration file: what is not a rafft_params field (the reference's own parameters, rafft/rafft.py:219-221) is
// an environment variable, read by read_config() below and NOWHERE else.  When they are read is part of the contract:
//   * scheduler settings        once, when the scheduler thread starts (first submission of the process, or the first one after
//                               rafft_shutdown(): the tests restart it to change them);
//   * everything else           at every rafft_fold_submit / rafft_fold_batch / seam call, on the caller's thread: the snapshot travels
//                               with the batch, so a call sees the environment as it was when it was made (batches whose snapshots
//                               differ are never merged into one wave);
//   * process-wide diagnostics  (RAFFT_TRACE_ALLOC, RAFFT_PRIO) once, at rafft_init.
// Kinds: T tuning (defaults measured on MI355X), D diagnostic / profiling, X experiment kept for A/B runs,
//        H test hook (compiled out with -DRAFFT_NO_TEST_HOOKS).
// INTEGRATION.md section 5 lists them for callers.
#pragma once
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Config {
    // ---- the 8-byte fields first (no padding inside the struct: two snapshots are compared as bytes)
    double reserve_frac = 0.10;// RAFFT_RESERVE_FRAC  T  workspaces are reserved for the merge cap when that stays below this share of the HBM
    double est = 0.0;          // RAFFT_EST           H  arena estimate (survivors per beam slot); 0: from the lengths
    long merge_seqs = 16384;   // RAFFT_MERGE_SEQS    T  sequences one merged wave may hold
    long linger_us = 600;      // RAFFT_LINGER_US     T  a stream of submissions is merged while they keep coming this close together (round 5: 150 -> 600 us - a Python caller queues a batch of the benchmark set every 250-300 us, and the first one of a burst was folded alone)
    long spin_us = 200, nap_us = 50; // RAFFT_SCHED_SPIN_US / RAFFT_SCHED_NAP_US  T  the scheduler thread polls this long after progress, then naps in slices
    double big_wave_frac = 0.10; // RAFFT_BIG_WAVE_FRAC T  waves whose arenas pass this share of the HBM run one at a time
    long test_cand_limit = 0;  // RAFFT_TEST_CAND_LIMIT H  lower the 31-bit limit of the candidate table (split path on small jobs)
    // ---- T: size classes and kernel plans (per batch)
    int cls1_p = 512;          // RAFFT_CLS1_P        T  FFT size limit of the one-wavefront expand class (256..512)
    int nt2 = 256;             // RAFFT_NT2           X  threads of the medium expand class (256 | 512)
    int tab = 0;               // RAFFT_TAB           X  bit c: energy tables of expand class c in LDS
    int wpb = 0;               // RAFFT_WPB           X  wavefronts per workgroup of the one-wavefront class (0: 16 without FFT buffers, else 12)
    int c1_per_cu = 0;         // RAFFT_C1_PER_CU     X  cap on its teams per CU (0: what the LDS allows)
    int c1_wgs = 0;            // RAFFT_C1_WGS        X  cap on its workgroups (0: none)
    int c3_direct = 1;         // RAFFT_C3_DIRECT     T  regions of 1025-4096 positions without FFT buffers (0: the LDS FFT plan)
    int c3_switch = -1;        // RAFFT_C3_SWITCH     T  ... up to this many regions per step the FFT plan works (-1: one per CU)
    int direct_n = 1024;       // RAFFT_DIRECT_N      T  wide classes: popcount correlation up to this region size, FFT beyond
    int c1_fft = 0, c2_fft = 0;// RAFFT_C1_FFT/C2_FFT X  keep the FFT buffers of the one-wavefront / 256-thread class
    int force_fft = 0;         // RAFFT_FORCE_FFT     D  FFT correlation for short regions too (parity tests)
    int prod = 1;              // RAFFT_PROD          D  0: the general builds of the kernels (seam, stamps compiled in)
    int no_memo = 0;           // RAFFT_NO_MEMO       D  1: every structure expands its own regions (no sharing of identical loops)
    int small_n4 = 16, small_n5 = 32; // RAFFT_SMALL="n4,n5" T  region sizes of the two small-region classes ("0,0": off)
    int small_wg = 4;          // RAFFT_SMALL_WG      T  their workgroups per CU
    int small_first = 0;       // RAFFT_SMALL_FIRST   X  launch them before the big-LDS classes
    int small_step0 = 0;       // RAFFT_SMALL_STEP0   D  launch them (empty) in the first step too
    int small_diag = 0;        // RAFFT_SMALL_DIAG    D  early exits of the small-region kernel
    int slab = 64;             // RAFFT_SLAB          T  candidate slots an expand wavefront reserves at a time
    int fetch = 4, taper = 25; // RAFFT_FETCH/TAPER   T  work chunks: regions per claim in the bulk of a list / percent handed out that way
    int mat_tile = 64;         // RAFFT_MAT_TILE      H  productive regions per tile of the materialize kernels (tests: several tiles)
    int rl_cap = -1;           // RAFFT_RL_CAP        H  beam step: regions with a choice kept in LDS (-1: RL_CAP)
    int mat4 = 1;              // RAFFT_MAT4          X  0: one structure per wavefront in the materialize step
    int dedupe_per_cu = 0;     // RAFFT_DEDUPE_PER_CU X  workgroups per CU of dedupe_kernel (0: default)
    int wide_below = 600;      // RAFFT_WIDE_BELOW    T  fewer unfinished sequences than this: 1024-thread beam step
    int merge_below = -1;      // RAFFT_MERGE_BELOW   T  new structures per step below which every region goes to the widest class (-1: 2 per CU)
    int merge2_below = -1;     // RAFFT_MERGE2_BELOW  T  ... to the 256-thread class (-1: 128 per CU)
    int split = -1;            // RAFFT_SPLIT         T  long-tail cut of a batch: -1 automatic, 0 never, > 0 at that length
    int no_harvest = 0;        // RAFFT_NO_HARVEST    X  1: no early copy-out of finished sequences
    int step_ahead = 0;        // RAFFT_STEP_AHEAD    X  1: the host issues a step ahead of its read-backs (measured: no gain - DESIGN.md 3.8)
    int serial = 0;            // RAFFT_SERIAL        D  every kernel of a step on one stream (per-kernel profiles)
    // ---- D: diagnostics (per batch)
    int trace = 0;             // RAFFT_TRACE         D  1: per-wave summaries, 2: per-step work lists, 3: phase stamps (general builds)
    int spans = -1;            // RAFFT_SPANS         D  HIP-event spans: 0 none, 1 dominant kernel (default), 2 every stage
    int rep = 0;               // RAFFT_REP           D  bit k doubles phase k of the expand kernel
    int twice = 0;             // RAFFT_TWICE         D  the one-wavefront kernel a second time on the same work (phase costs)
    int prof_seq = INT_MIN;    // RAFFT_PROF_SEQ      D  sequence whose beam step is stamped (-1: all)
    // ---- scheduler (read when the scheduler thread starts)
    int max_waves = 3;         // RAFFT_MAX_WAVES     T  bulk waves in flight
    int admit_below = 0;       // RAFFT_ADMIT_BELOW   X  admit the next bulk wave once the running one creates fewer structures per step (0: 128 per CU)
    int tail_slot = 1;         // RAFFT_TAIL_SLOT     X  the long-tail lane has a wave slot of its own
    // ---- process-wide (read at rafft_init)
    int prio = 1;              // RAFFT_PRIO          X  stream priorities of the bulk lanes (0: none, < 0: swapped)
    int trace_alloc = 0;       // RAFFT_TRACE_ALLOC   D  log every device / pinned allocation
    // ---- H: test hooks
    int test_hard_fail = -1;   // RAFFT_TEST_HARD_FAIL  H  a wave of exactly this many sequences fails hard
    int test_max_prod = 0;     // RAFFT_TEST_MAX_PROD   H  short productive-region lists of this length (overflow early)
    int test_ovf_at = -1;      // RAFFT_TEST_OVF_AT     H  pretend an arena overflowed at this step of the first attempt
    int seen_fixed = 0;        // RAFFT_SEEN_FIXED    X  1: every `seen` set starts at SEEN0 slots instead of a table sized from the length (tests: the growth path)
};
static_assert(sizeof(Config) == 8 * 8 + 4 * 48, "Config: 8-byte fields first, an even number of ints - no padding (same_config compares bytes)");

inline Config read_config()
{
    Config c;
    auto I = [](const char *name, int &v) { if (const char *e = getenv(name)) v = atoi(e); };
    auto L = [](const char *name, long &v) { if (const char *e = getenv(name)) v = atol(e); };
    auto F = [](const char *name, double &v) { if (const char *e = getenv(name)) v = atof(e); };
    auto B = [](const char *name, int &v) { if (getenv(name)) v = 1; };          // present = on, whatever the value
    I("RAFFT_CLS1_P", c.cls1_p); I("RAFFT_NT2", c.nt2); I("RAFFT_TAB", c.tab); I("RAFFT_WPB", c.wpb); I("RAFFT_C1_PER_CU", c.c1_per_cu);
    I("RAFFT_C1_WGS", c.c1_wgs); I("RAFFT_C3_DIRECT", c.c3_direct); I("RAFFT_C3_SWITCH", c.c3_switch); I("RAFFT_DIRECT_N", c.direct_n);
    I("RAFFT_C1_FFT", c.c1_fft); I("RAFFT_C2_FFT", c.c2_fft); I("RAFFT_FORCE_FFT", c.force_fft); I("RAFFT_PROD", c.prod); I("RAFFT_NO_MEMO", c.no_memo);
    if (const char *e = getenv("RAFFT_SMALL")) { int a = 16, b = 32; if (sscanf(e, "%d,%d", &a, &b) >= 1) { c.small_n4 = a; c.small_n5 = b; } }
    I("RAFFT_SMALL_WG", c.small_wg); I("RAFFT_SMALL_FIRST", c.small_first); B("RAFFT_SMALL_STEP0", c.small_step0); I("RAFFT_SMALL_DIAG", c.small_diag);
    I("RAFFT_SLAB", c.slab); I("RAFFT_FETCH", c.fetch); I("RAFFT_TAPER", c.taper); I("RAFFT_MAT4", c.mat4);
    I("RAFFT_DEDUPE_PER_CU", c.dedupe_per_cu); I("RAFFT_WIDE_BELOW", c.wide_below); I("RAFFT_MERGE_BELOW", c.merge_below); I("RAFFT_MERGE2_BELOW", c.merge2_below);
    F("RAFFT_RESERVE_FRAC", c.reserve_frac); I("RAFFT_SPLIT", c.split); B("RAFFT_NO_HARVEST", c.no_harvest);
    I("RAFFT_STEP_AHEAD", c.step_ahead); I("RAFFT_SERIAL", c.serial); I("RAFFT_SEEN_FIXED", c.seen_fixed);
    I("RAFFT_TRACE", c.trace); if (getenv("RAFFT_TRACE") && c.trace < 1) c.trace = 1;      // (set to anything: at least the summaries)
    I("RAFFT_SPANS", c.spans); I("RAFFT_REP", c.rep); I("RAFFT_TWICE", c.twice); I("RAFFT_PROF_SEQ", c.prof_seq);
    I("RAFFT_MAX_WAVES", c.max_waves); L("RAFFT_MERGE_SEQS", c.merge_seqs); I("RAFFT_ADMIT_BELOW", c.admit_below); I("RAFFT_TAIL_SLOT", c.tail_slot);
    L("RAFFT_LINGER_US", c.linger_us); L("RAFFT_SCHED_SPIN_US", c.spin_us); L("RAFFT_SCHED_NAP_US", c.nap_us); F("RAFFT_BIG_WAVE_FRAC", c.big_wave_frac);
    I("RAFFT_PRIO", c.prio); B("RAFFT_TRACE_ALLOC", c.trace_alloc);
#ifndef RAFFT_NO_TEST_HOOKS
    I("RAFFT_MAT_TILE", c.mat_tile); I("RAFFT_RL_CAP", c.rl_cap); F("RAFFT_EST", c.est);
    L("RAFFT_TEST_CAND_LIMIT", c.test_cand_limit); I("RAFFT_TEST_HARD_FAIL", c.test_hard_fail); I("RAFFT_TEST_MAX_PROD", c.test_max_prod); I("RAFFT_TEST_OVF_AT", c.test_ovf_at);
#endif
    return c;
}
// (two snapshots are "the same configuration" when they are the same bytes: plain data, value-initialised)
inline bool same_config(const Config &a, const Config &b) { return memcmp(&a, &b, sizeof(Config)) == 0; }
