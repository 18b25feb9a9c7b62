/*
 * rafft_hip.h - C-ABI of libraffthip.so, the MI355X (gfx950) RAFFT folding engine.
 *
 * The reference (lemerleau/RAFFT) is pure Python and has no FFI; its seam for the
 * fold hot path is the Python API + CLI + fast-folding-graph text format
 * (SURVEY.md section 8b).  This header is what a maintainer of the reference binds
 * with ctypes to make `rafft.fold()` / `bin/rafft` run on the
 * GPU.  Plain C types only; no exceptions cross the boundary; the library allocates
 * results and the caller frees them with rafft_free_result().
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference tree).
 */
#ifndef RAFFT_HIP_H
#define RAFFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (per call and per sequence) */
enum {
    RAFFT_OK = 0,
    RAFFT_ERR_BAD_CHAR = 1,      /* reference: KeyError from prep_sequence, rafft/utils.py:73-80 */
    RAFFT_ERR_EMPTY = 2,         /* reference: numpy AxisError from flip(), rafft/utils.py:83 */
    RAFFT_ERR_TOO_LONG = 3,      /* L > RAFFT_MAX_LEN (the LDS plan of the biggest regions: 2 bytes per position; 16-bit pair tables) */
    RAFFT_ERR_TEMP = 4,          /* temp != 37 with the built-in 37 C tables (no enthalpies): load a parameter file first */
    RAFFT_ERR_CAPACITY = 5,      /* an HBM arena overflowed even after regrowth */
    RAFFT_ERR_PARAM = 6,         /* unsupported parameter combination (e.g. max_branch+2*max_stack too large) */
    RAFFT_ERR_HIP = 7,           /* HIP runtime error; see rafft_last_error() */
    RAFFT_ERR_STRUCT = 8,        /* malformed dot-bracket / non-canonical pair in rafft_eval_structure */
    RAFFT_ERR_NO_DEVICE = 9
};

#define RAFFT_MAX_LEN 32768

/* Mirrors the argument list of rafft.fold(), rafft/rafft.py:219-221 (and
 * Glob_parms, rafft/utils.py:9-21). */
typedef struct {
    int32_t nb_mode;     /* -n  : number of positional lags searched per unpaired region */
    int32_t max_stack;   /* -ms : beam width */
    int32_t max_branch;  /* --max_branch */
    int32_t min_hp;      /* -mh */
    double min_nrj;      /* -mn */
    int32_t traj;        /* 0: final beam only; 1: beam of every folding step */
    int32_t _pad;
    double temp;         /* md.temperature (rafft/utils.py:18); != 37.0 needs rafft_load_params() (enthalpy tables) */
    double gc_wei, au_wei, gu_wei;
} rafft_params;

/* Per-sequence result: the fast-folding graph (bin/rafft:73-79) as arrays.
 * n_steps == 1 when params.traj == 0 (the final beam). */
typedef struct {
    int32_t status;
    int32_t length;
    int32_t n_steps;
    int32_t n_structs;        /* total rows over all steps */
    const int32_t *step_size; /* [n_steps] */
    const int32_t *step_off;  /* [n_steps] first row of each step */
    const char *db;           /* n_structs rows of (length+1) bytes, NUL terminated dot-brackets */
    const int32_t *dcal;      /* [n_structs] free energy in dcal/mol; kcal/mol = (float)dcal/100 as
                                 ViennaRNA's eval_structure returns it (rafft/utils.py:135-138) */
} rafft_seq_result;

typedef struct {
    int32_t n_seq;
    int32_t n_failed;         /* sequences whose status != RAFFT_OK */
    rafft_seq_result *seq;    /* [n_seq] in input order */
    void *_owner;             /* private */
} rafft_result;

/* Kernel timing / traffic counters of the last rafft_fold_batch() on this thread's
 * device (HIP events on the library's own stream).  Timing events are not free (a pair around every
 * kernel costs ~7 % of a benchmark batch), so by default only ms_total and ms_expand are measured;
 * the other ms_* fields are filled when the environment has RAFFT_SPANS=2 (or RAFFT_TRACE) at the
 * time of the call, and stay 0 otherwise.  RAFFT_SPANS=0 switches ms_expand off as well. */
typedef struct {
    double ms_total;          /* wall time of the call, host side */
    double ms_expand;         /* sum of the durations of the dominant kernel, expand_kernel<64,true,16,0,1>: regions of 33..256 positions
                                 with up to 128 branches, sixteen one-wavefront teams per workgroup (HIP events on its own stream) */
    double ms_expand_c1;      /* expand_small_kernel<16|32>: regions of up to 32 positions whose every lag is searched, teams of 16 / 32 lanes */
    double ms_expand_c2;      /* expand_kernel<256,...>: regions of up to 1024 positions that do not fit the one-wavefront class; runs concurrently */
    double ms_expand_c3;      /* regions of 1025..4096 positions (expand_kernel<256,false,1,2,3>: direct correlation, lag values in HBM; the LDS FFT
                                 plan expand_kernel<512,...> with RAFFT_C3_DIRECT=0) and the class for regions beyond 4096 positions; concurrently */
    double ms_expand_wall;    /* fork->join wall time of the concurrent expand launches of every step */
    double ms_beam;           /* sum of beam-step kernel durations */
    double ms_materialize;    /* sum of materialize kernel durations */
    double ms_output;         /* output formatting kernel */
    int64_t n_expand_launches; /* launches of the dominant kernel (steps with few new structures send their regions to one of the
                                 wide kernels instead) */
    int64_t n_steps;          /* folding steps executed (max over sequences) */
    int64_t n_node_expansions;/* regions really expanded (identical loops are expanded once) */
    int64_t n_nodes_created;  /* region records written: one per (parent region, candidate, side) that a beam member picked, whoever picked it first */
    int64_t n_nodes_aliased;  /* ... of these: loops reached before along another path, which re-use that region's expansion */
    int64_t sum_node_len;     /* sum of n over expansions */
    int64_t sum_lags;         /* sum of min(nb_mode, 2n-1) over expansions */
    int64_t n_structs;        /* structures materialized (beam survivors) */
    int64_t n_children;       /* children accepted by the combine step */
    int64_t sum_struct_len;   /* sum of L over materialized structures */
    int64_t alg_bytes;        /* algorithmic HBM bytes, SURVEY.md section 8d formula */
    int64_t alg_bytes_expand; /* the part of the dominant kernel expand_kernel<64> (its regions; 3L per structure pro rata) */
    int64_t alg_bytes_expand_all; /* all three expand size classes */
    int64_t n_regrows;        /* times a wave of the call overflowed its HBM arenas and was re-run with larger ones */
    int64_t alg_bytes_expand_small; /* algorithmic bytes of the small-region classes (expand_small_kernel; ms_expand_c1 is their time) */
    int64_t alg_bytes_expand_c2;    /* ... of expand_kernel<256> (ms_expand_c2) */
    int64_t alg_bytes_expand_c3;    /* ... of expand_kernel<512> (ms_expand_c3) */
    int64_t alg_bytes_beam;         /* 2L + 8 per new structure: beam_step_kernel + materialize_kernel (ms_beam + ms_materialize) */
    int64_t n_node_instances;       /* (structure, region) pairs: entries of the structures' node lists (n_nodes_created of them needed a record) */
    int64_t n_dE_evals;             /* built-in tables only (0 with a loaded parameter file): candidate stems whose dE was evaluated, */
    int64_t n_dE_guessed;           /* ... those whose dE read an interior-loop table entry that no reference-held energy row exercises */
    int64_t n_kept_guessed;         /* ... and the ones of these that passed the filter dE < min_nrj (rafft/rafft.py:102): candidates a beam member may pick */
    int64_t n_regrows_prod;         /* ... of n_regrows: a structure had more productive regions than the short lists hold (same arenas, long lists) */
    int64_t n_waves_long_lists;     /* waves of the call folded with the long productive-region lists (1024 per structure instead of 64: a structure with
                                       more than 64 was met under these parameters; eight such waves in a row that never needed them switch back) */
} rafft_stats;

/* Select the GPU (HIP ordinal) and upload the energy tables.  Optional: every other
 * entry point initialises lazily on device 0. */
int rafft_init(int device);

/* Replaces: one rafft.fold() call per sequence (rafft/rafft.py:219-239), i.e. the
 * body of benchmark_results/bench_fft.py:8-22 for a whole batch.  `device` < 0 keeps
 * the current device.  Sequences are independent; results come back in input order. */
int rafft_fold_batch(const rafft_params *p, int n_seq, const char *const *seqs,
                     const int *lens, int device, rafft_result **out);

void rafft_free_result(rafft_result *r);

/* Waits for the batches in flight, stops the library's scheduler thread and joins it.  Registered with atexit() when the
 * first batch is submitted; call it by hand before dlclose().  Later calls start a fresh scheduler. */
void rafft_shutdown(void);

/* Allocation counters of the library, process-wide and monotonic: out[0] device buffers allocated so far (hipMalloc calls),
 * out[1] their bytes, out[2] the slowest of those calls in microseconds, out[3] pinned host chunks allocated, out[4] their
 * bytes.  Workspaces are sized once for the biggest wave the scheduler may merge and are kept, so a steady stream of equal
 * batches allocates nothing after its first waves; a caller that times a region (bench.py) takes the difference around
 * it and reports it - a hipMalloc of gigabytes takes seconds now and then (tools/micro/malloc_busy.hip).
 * No counterpart in the reference (Python objects; benchmark_results/bench_fft.py:10-22 starts a process per sequence). */
void rafft_alloc_counters(unsigned long long out[5]);

/* The same, asynchronously - continuous batching.  rafft_fold_submit() copies the sequences, queues the batch and
 * returns at once; rafft_fold_wait() blocks until that batch is done and hands over its result (then the job
 * handle is gone).  One library thread drives all batches in flight: the last folding steps of a batch - which only
 * its longest sequences still need and which leave the GPU nearly idle - run beside the busy first steps of the
 * next one.  rafft_fold_batch() is submit + wait.  The analogue in the reference is the process pool of
 * benchmark_results/bench_fft.py:17-21, which also keeps several folds in flight.  Batches complete in any order;
 * every job must be waited for exactly once. */
typedef struct rafft_job rafft_job;
int rafft_fold_submit(const rafft_params *p, int n_seq, const char *const *seqs, const int *lens, int device,
                      rafft_job **job);
int rafft_fold_wait(rafft_job *job, rafft_result **out);

/* Thread-local message of the last failing call. */
const char *rafft_last_error(void);

/* Replaces: RNA.fold_compound(seq, md).eval_structure(db), the ViennaRNA call of
 * rafft/utils.py:135-138 (also benchmark_results/scoring.py:125).  Evaluated on the
 * GPU with the same device functions the fold kernels use. */
int rafft_eval_structure(const char *seq, const char *db, int *dcal_out);
int rafft_eval_structures(int n, const char *const *seqs, const char *const *dbs, int *dcal_out,
                          int *status_out);
/* the same at md.temperature = temp (rafft/utils.py:18) */
int rafft_eval_structures_at(double temp, int n, const char *const *seqs, const char *const *dbs, int *dcal_out,
                             int *status_out);

/* The same at 37 C, and for every structure whether its energy reads an entry of the built-in interior-loop tables (1x1, 2x1, 2x2,
 * the interior mismatches, bulge / interior sizes) that no reference-held (sequence, structure, energy) row exercises - a rule or
 * model value, right in ~9 of 10 cases: guessed_out[i] = 1.  Always 0 with a loaded parameter file.  No
 * counterpart in the reference, whose ViennaRNA has the real tables (rafft/utils.py:135-138). */
int rafft_eval_structures_info(int n, const char *const *seqs, const char *const *dbs, int *dcal_out, int *status_out,
                               int *guessed_out);
/* counts[0..2] = entries of the current 1x1 / 2x1 / 2x2 tables that are such rule / model values (0 with a loaded file) */
int rafft_params_unpinned(int counts[3]);

/* Energy parameters.  Replaces: the parameter set behind RNA.md() / RNA.fold_compound(sequence, md)
 * (rafft/utils.py:17-21) - ViennaRNA's compiled-in Turner 2004 set, or whatever the user loaded with
 * RNA.params_load(), rescaled to md.temperature.  rafft_load_params() reads the file format ViennaRNA 2.x reads and
 * writes ("## RNAfold parameter file v2.0": misc/rna_turner2004.par, RNA.params_save()), so ViennaRNA is touched
 * once, up front, for the tables and never inside the fold.  Without it the built-in 37 C tables are used
 * (DESIGN.md section 2 says how those are pinned).  Loading, inspecting and saving need no GPU; the device tables
 * are rebuilt by the next fold/eval call.  Not thread-safe against a fold running in another thread of the process
 * beyond the library's own lock (calls are serialised). */
int rafft_load_params(const char *path);
int rafft_load_params_text(const char *text, const char *source_name);
int rafft_reset_params(void);                                   /* back to the built-in tables */
int rafft_save_params(const char *path);                        /* counterpart of RNA.params_save(path) */
int rafft_params_info(char *source, int source_cap, int *has_enthalpies);
/* one entry of the current set: ViennaRNA table name ("stack", "int21", "mismatch_multi", "ml_closing", ...),
 * 37 C value or enthalpy, flat row-major index in ViennaRNA's array shape (pair axes 0..7, base axes 0..4) */
int rafft_param_value(const char *table, int enthalpy, long index, int *value_out);

/* Kernel-level seam for parity tests; replaces create_childs' search part:
 * auto_cor (rafft/utils.py:125-132) + ranking (rafft/rafft.py:117-118,92) +
 * window_slide (rafft/rafft.py:36-83) + the energy filter/sort of
 * find_best_consecutives (rafft/rafft.py:86-109) for ONE unpaired region `pos[0..n)`
 * of the structure `db`.  Output arrays must hold min(nb_mode, 2n-1) entries. */
int rafft_expand_node(const rafft_params *p, const char *seq, const char *db, const int *pos, int n,
                      int *n_ranked, int *lag, double *corval, int *nb, int *mi, int *mj,
                      double *score, int *ddcal, int *n_kept, int *kept);

int rafft_get_stats(rafft_stats *out);

/* Kinetics on the fast-folding graph (the "next" row 8f-2).  Replaces: get_connected_prev + get_transition_mat,
 * rafft/rafft_kin.py:48-56,68-91 - the O(steps * ms^2 * L) pair-set inclusion search and the Metropolis rate matrix.
 * Input: the graph as the fold returns it (n_steps beams, `rows` = all their dot-brackets back to back, L bytes each,
 * no terminator), `uid[r]` = index of row r in the list of unique structures in order of first appearance
 * (rafft_kin.py:106-112), `energy[u]` = energy of unique structure u, kt = 0.61 in the reference.
 * Output: the dense n_unique x n_unique rate matrix (row-major doubles, diagonal = -row sum) in DEVICE memory
 * `rate_device` (HIP pointer of the caller, e.g. a torch tensor's data_ptr) - at ms=1000 it has 10^8 entries and goes
 * straight into the dense solver. */
int rafft_kin_rate_matrix(int n_steps, const int *step_size, int L, const char *rows, const int *uid, int n_unique,
                          const double *energy, double kt, double *rate_device);

/* library / build information: "gfx950 ..." */
const char *rafft_version(void);

#ifdef __cplusplus
}
#endif
#endif
