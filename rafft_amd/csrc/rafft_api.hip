// rafft_api.hip - host side of libraffthip.so: the C-ABI of include/rafft_hip.h.
//
// One process drives one GPU.  All kernels run on the library's own HIP stream; the
// host loop is the reference's bfs_pairs recursion (rafft/rafft.py:156-216) turned
// into an iteration over folding steps that advances every sequence of the batch at
// once:   expand (new unpaired regions) -> beam step (per sequence) -> materialize
// (new beam members) -> ... until every sequence reached its fixed point.
#include "../../include/rafft_hip.h"
#include "rafft_kernels.h"
#include "rafft_params.h"
#include "rafft_config.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <pthread.h>

// kernels (rafft_kernels.hip is compiled into the same translation unit so the
// templates and the Dev struct are shared without a device-link step)
#include "rafft_kernels.hip"
#include "rafft_kin.hip"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess)                                                                          \
            return fail(RAFFT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                 \
    } while (0)

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};
struct PinBuf { void *p = nullptr; size_t cap = 0; };

#define MAX_PIPES 4
// One workspace = one folding pipeline: own stream set, grow-only device buffers and a pinned slot for the
// per-step read-back.  Even workspaces serve the long-tail lane of a batch, odd ones the bulk lane.
struct Workspace {
    bool ready = false;
    hipStream_t stream = nullptr;
    hipStream_t cls_stream[NCLS] = {};
    hipStream_t copy_stream = nullptr;   // result rows of sequences that finish early leave while the others still fold
    hipEvent_t ev_fork = nullptr, ev_join[NCLS] = {}, ev_hot[2] = {}, ev_copy = nullptr;
    void *hot = nullptr;                 // pinned, 2 x 1 KiB: the read-back slots of two consecutive steps (the host issues a step ahead)
    // named device buffers (grow-only)
    Buf codes, seq_off, seq_len, beam, beam_n, done, nsteps, ch_parent, ch_combo, ch_dcal, ch_h, seen, seen_off,
        seen_cap, seen_cnt, st, prod, nd, nlist, nd_slot, cslot, pos, br, sp, cand, looptab, trec, tsid,
        work0, work1, work2, work3, work4, work5, mat, counters,
        row_sid, row_off, out_db, out_dcal, row_off2, out_db2, out_dcal2, dbg, big;
    // every buffer of this workspace at least as big as its counterpart in `o` (defined after ensure())
    int match(const Workspace &o);
    void release_buffers()
    {
        for (Buf *b : {&codes, &seq_off, &seq_len, &beam, &beam_n, &done, &nsteps, &ch_parent, &ch_combo, &ch_dcal, &ch_h, &seen,
                       &seen_off, &seen_cap, &seen_cnt, &st, &prod, &nd, &nlist, &nd_slot, &cslot, &pos, &br, &sp, &cand, &looptab, &trec, &tsid,
                       &work0, &work1, &work2, &work3, &work4, &work5, &mat, &counters, &row_sid, &row_off, &out_db, &out_dcal, &row_off2, &out_db2,
                       &out_dcal2, &dbg, &big})
            if (b->p) { hipError_t e_ = hipFree(b->p); (void)e_; b->p = nullptr; b->cap = 0; }
    }
    size_t bytes() const
    {
        size_t t = 0;
        for (const Buf *b : {&codes, &seq_off, &seq_len, &beam, &beam_n, &done, &nsteps, &ch_parent, &ch_combo, &ch_dcal, &ch_h, &seen,
                             &seen_off, &seen_cap, &seen_cnt, &st, &prod, &nd, &nlist, &nd_slot, &cslot, &pos, &br, &sp, &cand, &looptab, &trec, &tsid,
                             &work0, &work1, &work2, &work3, &work4, &work5, &mat, &counters, &row_sid, &row_off, &out_db, &out_dcal, &row_off2, &out_db2,
                             &out_dcal2, &dbg, &big})
            t += b->cap;
        return t;
    }
};

struct Ctx {
    bool ready = false;
    int device = -1;
    int n_cu = 256;
    EnergyTables *T = nullptr;
    double T_temp = -1e300;            // temperature the device tables were scaled for
    bool T_dirty = true;               // the parameter set changed since the last upload
    rafft_par::ParamSet *P = nullptr;  // current parameter set (built-in until rafft_load_params)
    float2 *tw = nullptr;
    size_t hbm_total = 0;
    Workspace ws[MAX_PIPES];
    std::vector<PinBuf> pin_free;
    std::mutex pin_mu;                 // the pinned-chunk pool is used by the scheduler thread and by rafft_free_result
    std::vector<hipEvent_t> ev_free;   // timing events (scheduler thread only)
    std::vector<void *> garbage;       // device buffers replaced by bigger ones: freed when no wave is running (hipFree waits
    std::mutex gc_mu;                  //   for the whole device - tens of ms per regrown workspace while kernels are in flight)
    std::mutex ws_mu;                  // held by the seam calls that borrow workspace 0 on the caller's thread (vs idle trimming)
    rafft_stats stats{};               // of the batch that was waited for last
    std::mutex mu;                     // serialises the C-ABI entry points
    // ---- scheduler: one thread drives every wave of every batch in flight (see `scheduler_main`)
    std::mutex qmu;
    std::condition_variable qcv_sched, qcv_done;
    std::deque<std::shared_ptr<struct Batch>> submitted;
    int n_inflight = 0;                // batches submitted and not yet finished
    bool last_submit_async = false;    // under qmu: the last batch came through rafft_fold_submit (its caller may be about to queue more)
    std::chrono::steady_clock::time_point t_last_submit{};   // under qmu: when the last batch was queued (the scheduler lingers on a stream of them)
    Config proc_cfg;                   // read at rafft_init: the process-wide switches (rafft_config.h)
    Config sched_cfg;                  // read when the scheduler thread starts: its own settings
    bool sched_started = false;
    bool stop = false;                 // under qmu: the process is exiting (rafft_shutdown): the scheduler thread returns
    std::thread sched_thread;
};
// Never destroyed: the scheduler thread sleeps on its condition variable for as long as the process lives, and a
// condition variable must not be destroyed under a waiter (glibc's pthread_cond_destroy would block process exit).
Ctx &g = *new Ctx();

// allocations made so far: device buffers (calls, bytes, slowest call in ms) and pinned chunks (calls, bytes) - rafft_alloc_counters()
std::atomic<unsigned long long> g_dev_allocs{0}, g_dev_bytes{0}, g_dev_worst_us{0}, g_pin_allocs{0}, g_pin_bytes{0};

int ensure(Buf &b, size_t bytes, bool exact = false)
{
    if (bytes <= b.cap) return 0;
    const auto t0_ = std::chrono::steady_clock::now();
    const size_t old_cap = b.cap;
    if (b.p) { std::lock_guard<std::mutex> lk(g.gc_mu); g.garbage.push_back(b.p); b.p = nullptr; b.cap = 0; }
    // a buffer that had to grow once will grow again: leave room (at most 256 MB of it)
    size_t want = exact ? bytes : bytes + std::min<size_t>(bytes / (old_cap ? 2 : 8), (size_t)256 << 20) + 256;
    want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);      // whole 2 MiB fragments
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        // out of memory with replaced buffers still waiting for an idle moment to be freed: free them now (hipFree waits for the
        // device - a stall, not a failure) and ask again, for what is needed without the head-room
        (void)hipGetLastError();
        std::vector<void *> junk;
        { std::lock_guard<std::mutex> lk(g.gc_mu); junk.swap(g.garbage); }
        for (void *q : junk) { hipError_t e2 = hipFree(q); (void)e2; }
        want = (bytes + 256 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        e = hipMalloc(&b.p, want);
    }
    {
        const unsigned long long us = (unsigned long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0_).count();
        g_dev_allocs++; g_dev_bytes += want;
        unsigned long long w = g_dev_worst_us.load();
        while (us > w && !g_dev_worst_us.compare_exchange_weak(w, us)) { }
    }
    if (g.proc_cfg.trace_alloc) fprintf(stderr, "[rafft] ptr %p (mod 2MiB %zu KiB) ", b.p, ((size_t)(uintptr_t)b.p & (((size_t)2 << 20) - 1)) >> 10);
    if (g.proc_cfg.trace_alloc) fprintf(stderr, "[rafft] t=%.3f device buffer -> %.1f MB in %.3f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), (double)want / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count());
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(RAFFT_ERR_HIP, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    return 0;
}

int Workspace::match(const Workspace &o)
{
    Buf *mine[] = {&codes, &seq_off, &seq_len, &beam, &beam_n, &done, &nsteps, &ch_parent, &ch_combo, &ch_dcal, &ch_h, &seen,
                   &seen_off, &seen_cap, &seen_cnt, &st, &prod, &nd, &nlist, &nd_slot, &cslot, &pos, &br, &sp, &cand, &looptab, &trec, &tsid,
                   &work0, &work1, &work2, &work3, &work4, &work5, &mat, &counters, &row_sid, &row_off, &out_db, &out_dcal, &row_off2, &out_db2, &out_dcal2};
    const Buf *theirs[] = {&o.codes, &o.seq_off, &o.seq_len, &o.beam, &o.beam_n, &o.done, &o.nsteps, &o.ch_parent, &o.ch_combo, &o.ch_dcal, &o.ch_h, &o.seen,
                           &o.seen_off, &o.seen_cap, &o.seen_cnt, &o.st, &o.prod, &o.nd, &o.nlist, &o.nd_slot, &o.cslot, &o.pos, &o.br, &o.sp, &o.cand, &o.looptab, &o.trec, &o.tsid,
                           &o.work0, &o.work1, &o.work2, &o.work3, &o.work4, &o.work5, &o.mat, &o.counters, &o.row_sid, &o.row_off, &o.out_db, &o.out_dcal, &o.row_off2, &o.out_db2, &o.out_dcal2};
    for (size_t i = 0; i < sizeof(mine) / sizeof(mine[0]); i++)
        if (int rc = ensure(*mine[i], theirs[i]->cap, true)) return rc;
    return 0;
}

rafft_par::ParamSet &param_set()
{
    if (!g.P) { g.P = new rafft_par::ParamSet(); rafft_par::builtin(*g.P); }
    return *g.P;
}

// Device energy tables for `temp`: the current parameter set rescaled as ViennaRNA does for md.temperature
// (rafft/utils.py:17-21).  Submission is asynchronous: the caller (submit_locked) drains the batches in flight first, so
// the device is idle when the tables are replaced.
int ensure_tables(double temp)
{
    if (!g.T_dirty && g.T_temp == temp) return 0;
    std::unique_ptr<EnergyTables> h(new EnergyTables());
    std::string err;
    if (!rafft_par::scaled_tables(param_set(), temp, h.get(), err)) return fail(RAFFT_ERR_TEMP, err);
    HIPCHK(hipMemcpy(g.T, h.get(), sizeof(EnergyTables), hipMemcpyHostToDevice));
    g.T_temp = temp; g.T_dirty = false;
    return 0;
}

int init_ws(Workspace &w)
{
    if (w.ready) return 0;
    // Stream priorities: streams of another priority have HW queues of their own, so the kernels of one wave do not
    // queue behind those of another.  Workspaces 1 and 3 serve the bulk lane (of consecutive batches): high and low;
    // 0 and 2 the long-tail lane: normal.  (Measured with two waves: bulk high or low 13.4 ms, no priorities 17.3 ms,
    // the long tail high 15.4 ms.)  RAFFT_PRIO=0 switches priorities off.
    int plo = 0, phi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    const int prio_mode = g.proc_cfg.prio;
    const int idx = (int)(&w - g.ws);
    const int prio = prio_mode == 0 ? 0 : idx == 1 ? (prio_mode > 0 ? phi : plo) : idx == 3 ? (prio_mode > 0 ? plo : phi) : 0;
    HIPCHK(hipStreamCreateWithPriority(&w.stream, hipStreamNonBlocking, prio));
    for (int c = 0; c < NCLS; c++) {
        HIPCHK(hipStreamCreateWithPriority(&w.cls_stream[c], hipStreamNonBlocking, prio));
        HIPCHK(hipEventCreateWithFlags(&w.ev_join[c], hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&w.copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w.ev_copy, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) HIPCHK(hipEventCreateWithFlags(&w.ev_hot[k], hipEventDisableTiming | hipEventBlockingSync));   // (the scheduler sleeps on it when it has spun long enough)
    static_assert(offsetof(Counters, node) <= 1024, "hot counters must fit the pinned read-back slot");
    HIPCHK(hipHostMalloc(&w.hot, 2048, hipHostMallocDefault));
    w.ready = true;
    return 0;
}

int init_ctx(int device)
{
    if (g.ready && (device < 0 || device == g.device)) {
        HIPCHK(hipSetDevice(g.device));    // HIP's current device is per host thread: bind it on every entry
        return 0;
    }
    g.proc_cfg = read_config();
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(RAFFT_ERR_NO_DEVICE, "no HIP device: libraffthip.so has no CPU fallback");
    if (device < 0) device = 0;
    if (device >= ndev) return fail(RAFFT_ERR_NO_DEVICE, "device ordinal out of range");
    if (g.ready) return fail(RAFFT_ERR_PARAM, "library already initialised on another device in this process");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    g.hbm_total = prop.totalGlobalMem;
    g.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipMalloc((void **)&g.T, sizeof(EnergyTables)));
    g.T_dirty = true;
    std::vector<float2> tw(MAX_P / 2);
    for (int m = 0; m < MAX_P / 2; m++) {
        double a = -2.0 * M_PI * (double)m / (double)MAX_P;
        tw[m] = make_float2((float)cos(a), (float)sin(a));
    }
    HIPCHK(hipMalloc((void **)&g.tw, sizeof(float2) * tw.size()));
    HIPCHK(hipMemcpy(g.tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    g.device = device;
    g.ready = true;
    return init_ws(g.ws[0]);
}

struct ClsCfg { int nt, Pmax, Lmax, nmax, brmax, Kmax, lds, grid; bool tab; int wpb; bool nofft; bool direct3 = false; };   // grid: teams (wavefronts of the packed one-wavefront class, workgroups otherwise)

// limits of the one-wavefront class: its LDS per wavefront (hence its occupancy) follows from them
// (not below 256: the kernel addresses the staged bases through a pointer shifted back by up to 4095 positions, which must stay
//  inside the LDS - the 16 * P bytes in front of that area see to it)
static int cls1_P(const Config &cfg) { return std::max(256, std::min(next_pow2_ge(cfg.cls1_p), CLS1_P)); }
static int cls1_br(const Config &cfg) { return std::min(CLS1_BR, (8 * cls1_P(cfg) - 16) / 10 - 1); }

// `nofft1`: the one-wavefront class correlates every region by popcounts (production mode: no seam, no forced FFT, no negative
// weights, Dev::direct_n >= 256): its FFT buffers and twiddles go, its branch lists shrink to 128 entries, and a workgroup of twelve
// wavefronts leaves ~32 KiB of a CU's LDS - room for a workgroup of the small-region kernel beside it.
// `nofft2`: the same for the 256-thread class when Dev::direct_n covers all of its regions (<= 1024 positions): 46 -> 39 KiB, four
// workgroups per CU instead of three.
int class_cfg(const Config &cfg, int K, int maxL, ClsCfg out[NGEN + 1], bool nofft1 = false, bool nofft2 = false, bool direct3_ok = false)
{
    // sequences longer than LDS_SEQ: classes 2 and 3 read the bases of a loop from HBM/L2 (no LDS copy), class 0 takes the
    // regions whose FFT would not fit
    const bool longseq = maxL > LDS_SEQ;
    const int P[NGEN] = {CLS0_P, cls1_P(cfg), CLS2_P, MAX_P}, LM[NGEN] = {0, CLS01_L, longseq ? 0 : LDS_SEQ, longseq ? 0 : LDS_SEQ};
    const int NT[NGEN] = {512, 64, cfg.nt2, 512}, BR[NGEN] = {BIG_BR + 1, nofft1 ? 128 : cls1_br(cfg), MAX_BR, MAX_BR};
    // class 0 (tiny regions in their own kernel) is kept compiled for experiments but receives no work (see node_class)
    const int tabm = cfg.tab;      // bit c: energy tables of class c in LDS
    // The one-wavefront class packs 12 wavefronts - what a CU holds of them anyway - into one workgroup that shares ONE LDS
    // copy of the energy tables and twiddles: the table look-ups of the dE phase stop being dependent L2 round trips
    // (measured: 5.9 -> 5.4 ms per benchmark batch in this kernel; with 4 or 8 per workgroup a CU holds fewer wavefronts
    // and loses more than it gains).  Falls back to one wavefront per workgroup, tables in L2, when nb_mode makes the
    // per-wavefront arrays too big for 12 to fit.  RAFFT_WPB=1/4/12 overrides.
    int wpb1 = cfg.wpb ? cfg.wpb : (nofft1 ? 16 : 12);
    if (!(wpb1 == 4 || wpb1 == 12 || (wpb1 == 16 && nofft1))) wpb1 = 1;
    if (wpb1 > 1) {
        const int Kmax1 = std::max(1, std::min(K, cls1_P(cfg) - 1));
        if (wpb1 == 16 && expand_lds(cls1_P(cfg), CLS01_L, cls1_P(cfg) / 2, BR[1], Kmax1, true, wpb1, nofft1, 64).total > 160 * 1024) wpb1 = 12;
        if (expand_lds(cls1_P(cfg), CLS01_L, cls1_P(cfg) / 2, BR[1], Kmax1, true, wpb1, nofft1, 64).total > 160 * 1024) wpb1 = 1;
    }
    const int WPB[NGEN] = {1, wpb1, 1, 1};
    const bool TAB[NGEN] = {false, (tabm & 2) != 0 || WPB[1] > 1, (tabm & 4) != 0, false};    // (LDS tables come with LDS twiddles: FFT sizes <= CLS2_P only)
    for (int c = 0; c < NGEN; c++) {
        // (class 0 is planned for 16 384 positions unless a sequence of the wave is longer: the 64 KiB of 32 768 positions leave
        //  its scratch room for nb_mode <= 106 only, where the plan for 16 384 takes ~400)
        const int big_n = maxL > 16384 ? BIG_N : 16384;
        int nmax = c == 0 ? big_n : P[c] / 2;
        int Kmax = std::max(1, std::min(K, c == 0 ? 2 * big_n - 1 : P[c] - 1));
        const bool nf = (c == 1 && nofft1 && WPB[1] > 1) || (c == 2 && nofft2);
        ExpandLds l = expand_lds(P[c], LM[c], nmax, BR[c], Kmax, TAB[c], WPB[c], nf, NT[c], c != 0);      // (class 0: no LDS copy of the base codes - expand_kernel's CODE_LDS)
        // region A is time-shared: behind the fp64 lag values (8 P bytes) it must still hold the branch prefix sums
        // (10 bytes per branch), the select histogram and the window_slide scratch of this class
        if (c == 0 && longseq && (8 * MASK_WORDS * (big_n / 64) + 24 * 8 * std::max(Kmax, 1) + 4096 > 16 * P[c] || 10 * (BR[c] + 1) + 16 + 24 * Kmax + 2048 > 16 * P[c]))
            return fail(RAFFT_ERR_PARAM, std::string("nb_mode too large for the LDS scratch of the class for regions beyond 4096 positions: with a sequence of ") +
                                         (maxL > 16384 ? "more than 16384 nt it must stay at or below 106" : "more than 4096 nt it must stay below ~400") +
                                         " (this wave: nb_mode " + std::to_string(K) + ", longest sequence " + std::to_string(maxL) + " nt)");
        if (c >= 1 && (10 * (BR[c] + 1) + 16 > 8 * P[c] || 2 * P[c] + 1152 + 16 > 8 * P[c] || (NT[c] > 64 && NT[c] * 24 > 8 * P[c])))
            return fail(RAFFT_ERR_PARAM, "internal: expand LDS plan does not fit its size class");
        if (c >= 1 && LM[c] > 0 && l.off_S < LDS_SEQ)      // (expand_kernel's Sl: the staged bases are addressed by sequence position)
            return fail(RAFFT_ERR_PARAM, "internal: the LDS copy of the bases sits too low for its shifted pointer");
        int per_cu = std::max(1, std::min(32 / (NT[c] / 64), WPB[c] * ((160 * 1024) / l.total)));      // teams per CU
        if (c == 1 && cfg.c1_per_cu > 0) per_cu = std::max(1, std::min(per_cu, cfg.c1_per_cu));
        out[c] = {NT[c], P[c], LM[c], nmax, BR[c], Kmax, l.total, g.n_cu * per_cu, TAB[c], WPB[c], nf};
        // a size class that no region of this batch can reach need not fit (class 3 needs n > 1024, class 0 n > 4096)
        const bool reachable = c == 0 ? longseq : (c < 3 || maxL > CLS2_P / 2);
        if (l.total > 160 * 1024 && reachable)
            return fail(RAFFT_ERR_PARAM, "nb_mode too large for the LDS-resident expand kernel: it must stay below 2048 (below ~400 when a "
                                         "sequence is longer than 4096 nt)");
        if (l.total > 160 * 1024) out[c].lds = 160 * 1024, out[c].Kmax = 1;    // never launched with work
    }
    // Regions of 1025-4096 positions (class 3) without the 128-KiB FFT buffers: the kernel of the class for regions beyond 4096
    // positions - exact direct correlation on multi-word bit masks, lag values in a per-workgroup HBM scratch - with an LDS plan
    // sized for 4096 positions: ~50 KiB, so a CU holds two or three workgroups of it (four wavefronts per SIMD) instead of one.
    // Same integer pair counts, same fp64 values (tests/test_gpu_parity.py::test_gpu_fft_and_direct_correlation_agree).
    // (measured on the configs[3] shard: 202 -> 177 ms per call, the class itself 93 -> 66 ms - its regions cost 330 kcycles each at
    //  three workgroups per CU against 182 at one; the benchmark set, whose two 23S sequences are all it has of such regions, is
    //  unchanged.  RAFFT_C3_DIRECT=0: the FFT plan; read at every call, tests switch it.)
    const int c3_direct = cfg.c3_direct;
    if (c3_direct && direct3_ok && !longseq) {
        const int Kmax = std::max(1, std::min(K, MAX_P - 1)), nmax = MAX_P / 2, Pd = 2048;
        // (256 threads: at the 168 VGPRs the kernel needs without spilling a SIMD holds three wavefronts - three 256-thread
        //  workgroups per CU; a 512-thread workgroup is two wavefronts per SIMD, and two of those would need 128 VGPRs: 44 spilled)
        const int Cc = std::max(1, std::min(8, 256 / std::max(Kmax, 1)));
        const bool fits = 8 * MASK_WORDS * (nmax / 64) + 24 * Cc * Kmax + 2048 + 64 <= 16 * Pd && 10 * (MAX_BR + 1) + 16 + 8 * Kmax + 2048 <= 16 * Pd;
        ExpandLds l = expand_lds(Pd, 0, nmax, MAX_BR, Kmax, false, 1, false, 256);
        if (fits && l.total <= 80 * 1024) {
            const int per_cu = std::max(1, std::min(3, (160 * 1024) / l.total));
            out[NGEN] = out[3];          // the FFT plan stays for the steps with few such regions (launch_expand_cls)
            out[3] = {256, Pd, 0, nmax, MAX_BR, Kmax, l.total, g.n_cu * per_cu, false, 1, false, true};
        }
    }
    return 0;
}

template <int NT, bool TAB, int WPB = 1, int LONGSEQ = 0, int PROD = 0>
int launch_expand(const Dev &d, int cls, const ClsCfg &cf, unsigned n_teams, hipStream_t st)
{
    static int lds_set = 0;
    if (cf.lds > lds_set) {
        HIPCHK(hipFuncSetAttribute((const void *)expand_kernel<NT, TAB, WPB, LONGSEQ, PROD>, hipFuncAttributeMaxDynamicSharedMemorySize, cf.lds));
        lds_set = cf.lds;
    }
    const unsigned n_blocks = (n_teams + WPB - 1) / WPB;
    hipLaunchKernelGGL((expand_kernel<NT, TAB, WPB, LONGSEQ, PROD>), dim3(n_blocks), dim3(NT * WPB), cf.lds, st, d, cls, cf.Pmax, cf.Lmax, cf.nmax, cf.brmax, cf.Kmax);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_expand_cls(const Config &cfg, const Dev &d, int cls, const ClsCfg cf[NGEN + 1], unsigned n_blocks, hipStream_t st, bool dry = false)
{
    if (cls >= NGEN) {        // small regions: teams of 16 / 32 lanes, four wavefronts per workgroup (n_blocks = workgroups)
        const int arg = cls | (cfg.small_diag << 8);
        const bool prod_ok = cfg.prod != 0;
        const bool prod = prod_ok && arg == cls && d.prof_e == nullptr && d.dbg.lag == nullptr;      // no diagnostics asked for: the production build
        if (cls == 4 && prod) hipLaunchKernelGGL((expand_small_kernel<16, true>), dim3(n_blocks), dim3(64 * SM_WG_WAVES), small_lds_bytes<16>(), st, d, arg);
        else if (cls == 4) hipLaunchKernelGGL((expand_small_kernel<16, false>), dim3(n_blocks), dim3(64 * SM_WG_WAVES), small_lds_bytes<16>(), st, d, arg);
        else if (prod) hipLaunchKernelGGL((expand_small_kernel<32, true>), dim3(n_blocks), dim3(64 * SM_WG_WAVES), small_lds_bytes<32>(), st, d, arg);
        else hipLaunchKernelGGL((expand_small_kernel<32, false>), dim3(n_blocks), dim3(64 * SM_WG_WAVES), small_lds_bytes<32>(), st, d, arg);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const bool longseq = cf[2].Lmax == 0;          // (class_cfg: no LDS copy of the bases)
    const int nf = cls < NGEN && cf[cls].nofft ? 0x2000 : 0;
    // the production build of the classes without FFT buffers: no diagnostics of any kind asked for (RAFFT_PROD=0: the general build)
    const bool prod_ok = cfg.prod != 0;
    const bool nodiag = prod_ok && !dry && d.prof_e == nullptr && d.rep == 0 && d.dbg.lag == nullptr && !d.force_fft && d.gc >= 0.0 && d.au >= 0.0 && d.gu >= 0.0;
    const bool prod = nodiag && nf;
    if (cls == 0) return nodiag ? launch_expand<512, false, 1, 2, 2>(d, 0, cf[0], n_blocks, st) : launch_expand<512, false, 1, 2>(d, 0, cf[0], n_blocks, st);
    if (longseq && cls == 2 && cf[2].nt == 256) return prod ? launch_expand<256, false, 1, 1, 1>(d, 2 | nf, cf[2], n_blocks, st) : launch_expand<256, false, 1, 1>(d, 2 | nf, cf[2], n_blocks, st);
    if (longseq && cls >= 2) return nodiag && !nf ? launch_expand<512, false, 1, 1, 2>(d, cls, cf[cls], n_blocks, st) : launch_expand<512, false, 1, 1>(d, cls | nf, cf[cls], n_blocks, st);
    if (cls == 1) {
        if (cf[1].wpb == 4) return launch_expand<64, true, 4>(d, 1, cf[1], n_blocks, st);
        if (cf[1].wpb == 16 && prod) return launch_expand<64, true, 16, 0, 1>(d, 1 | nf, cf[1], n_blocks, st);
        if (cf[1].wpb == 16) return launch_expand<64, true, 16>(d, 1 | (cf[1].nofft ? 0x2000 : 0), cf[1], n_blocks, st);
        if (cf[1].wpb == 12) return launch_expand<64, true, 12>(d, (dry ? (0x101 | (std::max(0, cfg.twice - 2) << 9)) : 1) | (cf[1].nofft ? 0x2000 : 0), cf[1], n_blocks, st);
        return cf[1].tab ? launch_expand<64, true>(d, 1, cf[1], n_blocks, st) : launch_expand<64, false>(d, 1, cf[1], n_blocks, st);
    }
    if (cls == 2) {
        if (cf[2].nt == 512) return cf[2].tab ? launch_expand<512, true>(d, 2 | nf, cf[2], n_blocks, st) : launch_expand<512, false>(d, 2 | nf, cf[2], n_blocks, st);
        if (prod && !cf[2].tab) return launch_expand<256, false, 1, 0, 1>(d, 2 | nf, cf[2], n_blocks, st);
        return cf[2].tab ? launch_expand<256, true>(d, 2 | nf, cf[2], n_blocks, st) : launch_expand<256, false>(d, 2 | nf, cf[2], n_blocks, st);
    }
    if (cf[3].direct3) {
        // Two kernels share the class's work list; the length of the list decides ON THE DEVICE which of them works (the other one's
        // workgroups leave at once): up to Dev::c3_switch regions - every region has a CU to itself either way - the FFT plan, whose
        // region takes 76 us (182 kcycles) against 137 us without the FFT buffers; beyond that the FFT-free kernel, three workgroups
        // per CU.  (With the FFT-free kernel alone one synchronous call on the benchmark batch took 11.5 ms instead of 10.0: its
        // long-tail wave expands a handful of such regions per step, 24 steps in a row.)
        if (int rc = nodiag ? launch_expand<256, false, 1, 2, 3>(d, 3 | 0x8000, cf[3], n_blocks, st) : launch_expand<256, false, 1, 2>(d, 3 | 0x8000, cf[3], n_blocks, st)) return rc;
        const unsigned nb_fft = std::min<unsigned>(n_blocks, (unsigned)cf[NGEN].grid);
        return nodiag ? launch_expand<512, false, 1, 0, 2>(d, 3 | 0x4000, cf[NGEN], nb_fft, st) : launch_expand<512, false>(d, 3 | 0x4000, cf[NGEN], nb_fft, st);
    }
    if (nodiag && !cf[3].tab) return launch_expand<512, false, 1, 0, 2>(d, 3, cf[3], n_blocks, st);
    return cf[3].tab ? launch_expand<512, true>(d, 3, cf[3], n_blocks, st) : launch_expand<512, false>(d, 3, cf[3], n_blocks, st);
}

// timing events: handed out from a free list and returned when their batch has been finalised, so the spans of a
// wave stay valid while later waves (of the same or of another batch) use the same workspace
hipEvent_t next_event(std::vector<hipEvent_t> &used)
{
    hipEvent_t e = nullptr;
    if (!g.ev_free.empty()) { e = g.ev_free.back(); g.ev_free.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) return nullptr;
    used.push_back(e);
    return e;
}

struct Span { hipEvent_t a, b; int kind; };
// Timing events are not free: a pair around every kernel costs ~1.3 ms of the 17 ms benchmark batch (the markers
// serialise the queues).  Level 1 (default) times only the dominant kernel - the one-wavefront expand class, what
// the roofline is computed from; level 2 (RAFFT_SPANS=2 or RAFFT_TRACE) times every stage; level 0 none.
static int g_span_level = 1;
static inline bool span_on(int kind) { return g_span_level >= 2 || (g_span_level == 1 && kind == 11); }
#define SPAN_REC(ev, st, kind) do { if (span_on(kind)) HIPCHK(hipEventRecord((ev), (st))); } while (0)

// base codes of rafft/utils.py:73-80 (N=0 A=1 C=2 G=3 U=4); bit 3 marks a character outside "AGCUN"
struct BaseCodeTable {
    uint8_t v[256];
    BaseCodeTable() { for (int i = 0; i < 256; i++) v[i] = 8; v['N'] = 0; v['A'] = 1; v['C'] = 2; v['G'] = 3; v['U'] = 4; }
    uint8_t operator[](unsigned char c) const { return v[c]; }
};
static const BaseCodeTable kBaseCode;

struct Caps {
    size_t st, nd, pos, br, sp, cand, seen, trec, tsid, work, mat, looptab;
    int ch_cap, sort_cap;
    size_t bytes;
    bool capped;      // a table hit the limit of its 31-bit ids: the job is folded in halves when it has more than one sequence
};

// Initial slots of a sequence's `seen` set (round 5).  A set that outgrows its table is rehashed into one of twice the size inside
// beam_step_kernel - ~100 us of ONE workgroup (dependent compare-and-swap round trips), i.e. of the whole launch when it is the
// slowest: on the benchmark set every sequence beyond 200 nt grew once or twice (RAFFT_TRACE=2 prints the fill by length) and a
// fixed 65 536 slots took a sixth off both beam-step kernels.  Sized from the length instead: the upper envelope of the entries at
// the end of a fold (measured at max_stack 50, max_branch 1000: 1.9 k at 80 nt, 3.3 k at 130, 5.5 k at 300, 7.6 k at 500, 8.8 k at
// 1000, 20 k at 3000), scaled by the children a step accepts, for a table that is at most half full (the kernel's own growth rule).
// A set that still outgrows it grows as before.
static uint32_t seen_slots0(int L, const rafft_params &p, const Config &cfg)
{
    if (cfg.seen_fixed) return SEEN0;
    const double l = (double)L;
    const double e = l <= 130 ? 26.0 * l : l <= 300 ? 3380.0 + 13.0 * (l - 130) : l <= 500 ? 5590.0 + 10.5 * (l - 300) : l <= 1000 ? 7690.0 + 2.4 * (l - 500) : 8890.0 + 5.6 * (l - 1000);
    const double per_step = p.max_branch > 0 ? std::min((double)p.max_branch, 8.2 * (double)p.max_stack) : (double)p.max_stack;
    const double need = 2.0 * (e * std::max(per_step / 410.0, 0.1) + 384.0);
    uint32_t cap = 2048;
    while ((double)cap < need && cap < (1u << 22)) cap <<= 1;
    return cap;
}
// ... for a wave: per-sequence slots with the big tables halved until the initial tables fit `budget_slots` (the growth path does the rest)
static size_t seen_slots0_wave(const int *len, size_t S, const rafft_params &p, const Config &cfg, size_t budget_slots, uint32_t *out)
{
    uint32_t limit = 1u << 22;
    for (;;) {
        size_t tot = 0;
        for (size_t i = 0; i < S; i++) { const uint32_t c = std::min(seen_slots0(len[i], p, cfg), limit); if (out) out[i] = c; tot += c; }
        if (tot <= budget_slots || limit <= SEEN0) return tot;
        limit >>= 1;
    }
}
#define SEEN0_BUDGET ((size_t)192 << 20)        // slots: 3 GB of initial tables per wave at most

Caps plan_caps(const Config &cfg, size_t S, size_t sumL, const rafft_params &p, double est, double seen0_per_seq)
{
    // Arena sizes from measured usage on the BASELINE workloads (benchmark set, L 28..2968, ms 50;
    // random L 100..3000, ms 200): per surviving structure about 2 + L/100 regions, 0.6 L region
    // positions, one branch per region, ~5 candidates per region, the pairs a structure adds to its parent's
    // (one stem in every productive region - measured on L 100..3000: 30 pairs per structure, 0.02 L; structures of
    // long sequences are over-represented: they fold for more steps).  Factors below carry ~1.5x slack;
    // an overflow is detected on the device and the wave is re-run with doubled arenas.
    Caps c;
    const size_t B = (size_t)p.max_stack;
    double avgL = S ? (double)sumL / (double)S : 1.0;
    double nstruct = (double)S * (1.0 + (double)B * est);
    c.st = (size_t)std::min(nstruct, 2.0e9) + 64;
    double nodes_per = avgL / 60.0 + 4.0;
    c.nd = (size_t)std::min((double)c.st * nodes_per, 2.0e9) + 64;
    c.capped = nstruct > 2.0e9 || (double)c.st * nodes_per > 2.0e9;
    c.pos = (size_t)((double)sumL + (double)(c.st - S) * avgL * 0.9) + 4096;
    c.br = c.nd * 3 + 4096;
    c.sp = (size_t)((double)(c.st - S) * (avgL * 0.05 + 24.0)) + 4096;
    c.cand = c.nd * (size_t)std::min(std::max(p.nb_mode, 1), 8) + 4096;
    // (with memoization a region is created once per wave, whoever picks the stem that makes it: what a structure adds to the candidate
    //  table stops growing with the regions it HAS.  Measured, tools/arena_probe.py, candidates per structure: 10-20 on the benchmark
    //  set, 200-nt and 40-nt random sequences, ms 50 and 400; 15 on L 100..3000 at ms 200 - where the line above plans 238 -; 46 at
    //  ms 1; 50 on G/C-only sequences of 600 nt; 59 and 95 on 2.9-knt and 8-knt sequences at ms 50 and 20)
    const bool memo_on = p.min_nrj == 0.0 && !cfg.no_memo;
    if (memo_on) c.cand = std::min(c.cand, (size_t)((double)c.st * (60.0 + avgL / 40.0)) + 4096);
    c.cand = std::max<size_t>(c.cand, (size_t)NSHARD * 16384);
    // (a child slot is named by 2 x candidate + side in 31 bits - the node lists hold -(slot + 1), rafft_kernels.h - so a wave has
    //  at most 2^30 candidate records; round 4's structure rows used to keep the byte budget of a wave below that by themselves)
    size_t cand_limit = ((size_t)1 << 30) - 4096;
    if (cfg.test_cand_limit > 0) cand_limit = std::min<size_t>(cand_limit, std::max<size_t>((size_t)cfg.test_cand_limit, (size_t)NSHARD * 16384));   // (tests: the split path on small jobs)
    if (c.cand > cand_limit) { c.cand = cand_limit; c.capped = true; }
    // accepted children per sequence ~ steps * min(max_branch, ...); regions double and old ones are dropped
    // (measured, ms 50, max_branch 1000, regions abandoned by rehashing included: the benchmark set's bulk uses 4.6 x est x
    //  (B + max_branch / 4) slots per sequence, its two 2.9-knt sequences 11.6 x.  A factor of 24 used to reserve 1 MB per
    //  sequence - 12 GB for a merged wave of five batches, and a hipMalloc of that size now and then took seconds.)
    // (round 5: the initial tables are sized from the lengths - seen_slots0 - and a table rarely grows any more: the reserve for growth
    //  went from 14 x to 5 x est x (B + max_branch / 4), at least as much again as the initial tables.  At 14 x the arena was 10.6 GB of
    //  a merge-cap plan of 33.7 GB - above the tenth of the HBM up to which a first wave reserves its workspace for the merge cap, so
    //  every bigger wave of a stream re-allocated two dozen buffers; at 5 x the plan is 28.5 GB and the first wave's workspace holds them all.)
    double per_seq_seen = std::min(std::max(std::max(5.0 * est * ((double)B + (double)p.max_branch / 4.0), seen0_per_seq), 16384.0), 16777216.0);
    // (a floor of 4 M slots - 64 MB - whatever the batch: a few long sequences among sixty short ones double their tables twice)
    c.seen = (size_t)((double)S * seen0_per_seq) + std::max((size_t)((double)S * per_seq_seen), (size_t)4 << 20);
    c.trec = p.traj ? S * (size_t)(est * 3 + 16) : S + 16;
    c.tsid = c.trec * B + 16;
    c.mat = S * B + 16;
    // every arena is split into NSHARD sub-arenas: keep a floor per shard so that small batches,
    // whose few structures land on few shards, do not overflow a starved shard
    c.nd = std::max<size_t>(c.nd, S + (size_t)NSHARD * 2048);
    c.pos = std::max<size_t>(c.pos, sumL + (size_t)NSHARD * (16 * (size_t)avgL + 4096));
    c.sp = std::max<size_t>(c.sp, (size_t)NSHARD * (2 * (size_t)avgL + 4096));
    c.br = std::max<size_t>(c.br, (size_t)NSHARD * 8192);
    c.work = c.nd;
    // (the loop table holds the regions CREATED - one per child slot that a beam member picked, ~0.4 of the (structure, region) pairs
    //  `nd` is planned for: a power of two >= nd keeps it at most half full; it is zero-filled for every wave, and a full table is an
    //  overflow like any other - the wave is folded again with doubled arenas)
    c.looptab = 1024; while (c.looptab < c.nd) c.looptab <<= 1;
    c.ch_cap = p.max_branch + p.max_stack + 8;
    int need = p.max_branch + 2 * p.max_stack + 8;
    // keys of one step: children + old beam; only the max_stack selected ones are sorted (padded to a power of two)
    int m2 = 2; while (m2 < p.max_stack) m2 <<= 1;
    c.sort_cap = std::max((need + 1) & ~1, m2);
    c.bytes = c.st * 128 + c.nd * (64 + 4 + 4 + 6 * 4 + 16) + c.pos * 2 + c.br * 4 + c.sp * 4 + c.cand * (32 + 8) + c.seen * 16 +
              c.looptab * 8 + c.trec * 16 + c.tsid * 4 + c.mat * 48 + S * (size_t)c.ch_cap * 32 + S * B * 4;
    return c;
}

static size_t merge_cap();
struct SeqIn { const char *s; int len; int idx; int bi; const uint8_t *c = nullptr; };   // bi: which member batch of the job the sequence belongs to; c: the bases as codes (encoded at submit, on the caller's thread), or null

struct HostOut {   // owner of a rafft_result
    std::vector<rafft_seq_result> seq;
    std::vector<std::vector<int>> step_size, step_off;
    std::vector<int> one_size, one_off;     // ... of a sequence with a single step (every sequence without --traj)
    std::vector<const char *> db_ptr;       // rows live in pinned chunks (one per wave): the D2H copy lands
    std::vector<const int *> dcal_ptr;      // directly in the memory the caller reads
    std::vector<std::shared_ptr<struct PinChunk>> chunks;   // a chunk may hold rows of several batches folded as one wave
    rafft_result res;
};

// Pool of pinned host buffers, recycled across calls: hipHostMalloc / hipHostFree cost milliseconds for a result chunk
// of tens of MB and stall the queues while they run (measured: a steady stream of them turned 10 ms batches into
// 45-70 ms ones).  Sizes are rounded up to powers of two so that chunks of merged waves of different sizes reuse each
// other's buffers; the pool gives memory back only above 4 GB.
PinBuf pin_acquire(size_t bytes)
{
    std::lock_guard<std::mutex> lk(g.pin_mu);
    size_t want = 256 * 1024;
    while (want < bytes) want <<= 1;
    int best = -1;
    for (size_t i = 0; i < g.pin_free.size(); i++)
        if (g.pin_free[i].cap >= bytes && g.pin_free[i].cap <= 4 * want && (best < 0 || g.pin_free[i].cap < g.pin_free[best].cap)) best = (int)i;
    if (best >= 0) { PinBuf b = g.pin_free[best]; g.pin_free.erase(g.pin_free.begin() + best); return b; }
    PinBuf b;
    const auto t0_ = std::chrono::steady_clock::now();
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return b; }
    g_pin_allocs++; g_pin_bytes += want;
    if (g.proc_cfg.trace_alloc) fprintf(stderr, "[rafft] t=%.3f pinned chunk %.1f MB in %.3f ms (pool %zu)\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), (double)want / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(), g.pin_free.size());
    b.cap = want;
    return b;
}
void pin_release(PinBuf b)
{
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(g.pin_mu);
    size_t held = 0;
    for (auto &x : g.pin_free) held += x.cap;
    if (g.pin_free.size() < 64 && held + b.cap <= ((size_t)4 << 30)) g.pin_free.push_back(b);
    else {
        const auto t0_ = std::chrono::steady_clock::now();
        hipError_t e = hipHostFree(b.p); (void)e;
        if (g.proc_cfg.trace_alloc) fprintf(stderr, "[rafft] pinned chunk freed in %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count());
    }
}

// a pinned result chunk, returned to the pool when the last result that points into it is freed
struct PinChunk {
    PinBuf b;
    explicit PinChunk(PinBuf x) : b(x) {}
    ~PinChunk() { pin_release(b); }
    PinChunk(const PinChunk &) = delete;
    PinChunk &operator=(const PinChunk &) = delete;
};

struct Batch;
// One wave's worth of work.  `members`: the batches its sequences come from - queued jobs with identical parameters
// are merged (continuous batching), so one wave may serve several batches; seqs[i].bi indexes this list.
struct Job { std::vector<SeqIn> seqs; double est; int depth; std::vector<std::shared_ptr<Batch>> members; bool no_merge = false;
             bool big_prod = false; };    // re-run after a structure had more productive regions than the short lists hold

// One rafft_fold_submit(): its sequences (copied), its result under construction, its jobs (lane 0: the long tail of
// the batch, lane 1: the bulk - see rafft_fold_submit) and what the scheduler needs to finish it.
struct Batch {
    rafft_params p;
    Config cfg;                               // the environment switches as they were when the batch was submitted (rafft_config.h)
    int n_seq = 0;
    std::vector<char> seqbuf;                 // the caller's sequences, copied at submit
    std::vector<uint8_t> codebuf;             // ... and as base codes, same offsets
    HostOut *ho = nullptr;
    std::deque<Job> lane[2];                  // as submitted; the scheduler moves them to its own queues
    int pending = 0;                          // jobs (queued or running) that still hold sequences of this batch
    int rc = 0;
    std::string err;
    std::vector<Span> spans;
    std::vector<hipEvent_t> events;           // timing events in use by `spans`
    rafft_stats stats{};
    std::chrono::steady_clock::time_point t0;
    bool done = false;                        // under g.qmu
};

struct SeamIn {     // rafft_expand_node: one region of one given structure
    DebugOut dbg;
    std::vector<uint16_t> pos;
    std::vector<uint32_t> br;
    int ci, cj, pdcal;
};

// One wave = one batch of sequences folded in lock-step folding steps on one workspace.  It is a small
// state machine so that a single host thread can drive two waves at once (two pipelines): while it waits
// for one wave's 152-byte read-back the kernels of the other keep the GPU busy.
struct Wave {
    Workspace &g;                 // NB: named `g` on purpose - the buffers used to live in the global context
    const Config &cfg;            // of the first member batch (batches with other snapshots are never merged into the wave)
    rafft_params p;
    std::vector<SeqIn> seqs;
    double est;
    std::vector<std::shared_ptr<Batch>> members;   // whose sequences this wave folds (seqs[i].bi)
    Batch &bt;                                      // the first of them: carries the wave's timing spans and statistics
    std::vector<Span> &spans;
    const SeamIn *seam;
    size_t S = 0, sumL = 0, B = 0, bs_lds[2] = {0, 0}, mat_lds = RAFFT_MAX_LEN, out_row_lds = RAFFT_MAX_LEN;
    double reserve = 1.0;         // buffers are allocated for a wave this many times bigger (merged batches to come)
    bool longseq = false;         // a sequence longer than LDS_SEQ: its loops' bases are read from HBM, regions beyond 4096 positions exist
    unsigned dedupe_per_cu = 1024 / DEDUPE_NT;
    std::vector<int> off, len;
    std::vector<uint32_t> seen_cap0;      // initial slots of every sequence's `seen` set (seen_slots0)
    size_t seen0_total = 0;
    ClsCfg cf[NGEN + 1];          // (cf[NGEN]: the FFT plan of class 3 beside its FFT-free kernel)
    Caps c;
    Dev d;
    Counters hc;
    unsigned n_active = 0, ovf = 0, last_mat = 0;
    int merged_now = 0, merge_target = 0;   // size class that receives every region of the coming expand step (0: by size)
    int steps = 0;
    int depth = 0;                // regrowths of this job so far
    bool big_prod = false, want_big_prod = false;   // long productive-region lists (1024 per structure) for this run / asked for by it
    bool finished = false;
    // (round 5) read-backs issued / looked at.  In step-ahead mode the host queues the materialize step and the next folding step
    // BEFORE it has seen the counters of the running one (grids and class choices from the step before, counts from the device's own
    // counters): the 24 lock-step steps of a lone batch no longer wait 150 us each for the host's round trip
    int rb_issued = 0, rb_seen = 0;
    bool step_ahead = false;
    long long last_rows_bytes = 0;
    std::vector<OutRec> early_recs, late_recs;
    PinBuf stage{};               // pinned staging of the wave's inputs (setup)
    bool draining = false;        // every step is done, the last rows are on their way to the host (finish): ready() tells when they have landed
    double tl_stats_ = 0, tl_gather_ = 0, ms_loop_ = 0;
    std::chrono::steady_clock::time_point tw2_;
    ~Wave() { pin_release(stage); }
    size_t harvested = 0;         // trajectory records whose rows already left through the copy stream (early harvest)
    int emit_rows(size_t first, size_t count, bool early, double *t_gather);
    int result = 0;               // valid when finished: 0, RAFFT_ERR_CAPACITY (regrow) or a hard error
    std::chrono::steady_clock::time_point tw0, tw1;
    double ms_setup = 0, ms_issue = 0, ms_after = 0;   // host time inside issue_step / after_beam (trace)

    Wave(Workspace &w, std::vector<std::shared_ptr<Batch>> m, std::vector<SeqIn> s, double e, const SeamIn *sm = nullptr)
        : g(w), cfg(m[0]->cfg), p(m[0]->p), seqs(std::move(s)), est(e), members(std::move(m)), bt(*members[0]), spans(members[0]->spans), seam(sm) {}
    HostOut &out_of(int local_seq) { return *members[seqs[local_seq].bi]->ho; }

    double since(std::chrono::steady_clock::time_point t) const
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    }
    hipEvent_t next_event() { return ::next_event(bt.events); }
    // a wave whose steps still create many structures keeps the whole GPU busy; afterwards it is latency-bound
    bool heavy(unsigned below) const { return below != 0x7fffffffu && S >= 256 && !finished && (steps < 3 || last_mat >= below); }
    int setup();
    int issue_step();
    // 1: the step's read-back has landed, 0: not yet, -1: the device reported an error (sticky: the wave is failed, not polled forever)
    int ready()
    {
        const hipError_t e = hipEventQuery(g.ev_hot[rb_seen & 1]);
        if (e == hipSuccess) return 1;
        if (e == hipErrorNotReady) return 0;
        fail(RAFFT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        return -1;
    }
    std::chrono::steady_clock::time_point t_issued;      // when the running step was issued (the scheduler blocks on the oldest)
    int after_beam();
    int issue_materialize(unsigned n_mat_known, unsigned n_mat_guess);
    // (every error exit of the two leaves `finished` and `result` set: the scheduler reads `result` of a finished wave, and a failure
    //  while the rows are gathered - a device error, a pinned allocation - must not be released as a batch-level success)
    int finish() { const int rc = finish_body(); if (rc) { finished = true; draining = false; if (!result) result = rc; } return rc; }
    int finish_done() { const int rc = finish_done_body(); if (rc) { finished = true; if (!result) result = rc; } return rc; }
    int finish_body();
    int finish_done_body();
};

int Wave::setup()
{
    S = seqs.size();
    tw0 = std::chrono::steady_clock::now();
    // test hook: a wave of exactly this many sequences fails hard (what a structure beyond the kernels' limits does)
    if (cfg.test_hard_fail >= 0 && (int)S == cfg.test_hard_fail) return fail(RAFFT_ERR_PARAM, "test hook: hard failure of this wave");
    off.resize(S); len.resize(S);
    sumL = 0;
    for (size_t i = 0; i < S; i++) { off[i] = (int)sumL; len[i] = seqs[i].len; sumL += seqs[i].len; }
    // the wave's inputs are staged in a pinned chunk (codes | offsets | lengths | counters image): the uploads below are truly
    // asynchronous and the scheduler thread goes on to the other waves' steps at once (it used to wait here, 0.6-0.7 ms per
    // wave of five batches); the chunk goes back to the pool with the wave
    const size_t st_codes = 0, st_off = (sumL + 16 + 63) & ~(size_t)63, st_len = st_off + ((S * 4 + 63) & ~(size_t)63),
                 st_ctr = st_len + ((S * 4 + 63) & ~(size_t)63), st_soff = st_ctr + ((sizeof(Counters) + 63) & ~(size_t)63),
                 st_scap = st_soff + ((S * 8 + 63) & ~(size_t)63), st_bytes = st_scap + S * 4;
    stage = pin_acquire(st_bytes);
    if (!stage.p) return fail(RAFFT_ERR_HIP, "hipHostMalloc failed for the input staging buffer");
    uint8_t *codes = (uint8_t *)stage.p + st_codes;
    for (size_t i = 0; i < S; i++) {
        uint8_t *dst = codes + off[i];
        if (seqs[i].c) memcpy(dst, seqs[i].c, (size_t)seqs[i].len);          // (encoded at submit, on the caller's thread)
        else {
            const unsigned char *src = (const unsigned char *)seqs[i].s;
            for (int x = 0; x < seqs[i].len; x++) dst[x] = kBaseCode[src[x]] & 7;
        }
    }
    memset(codes + sumL, 0, 16);
    memcpy((char *)stage.p + st_off, off.data(), S * 4);
    memcpy((char *)stage.p + st_len, len.data(), S * 4);
    const double ms_enc = since(tw0);
    int maxL = 0;
    for (size_t i = 0; i < S; i++) maxL = std::max(maxL, len[i]);
    const int direct_n_ = cfg.direct_n;
    const bool force_fft_ = cfg.force_fft != 0;
    const bool nofft1 = !seam && !force_fft_ && direct_n_ >= cls1_P(cfg) / 2 && p.gc_wei >= 0.0 && p.au_wei >= 0.0 && p.gu_wei >= 0.0 &&
                        !cfg.c1_fft;
    const bool nofft2 = nofft1 && direct_n_ >= CLS2_P / 2 && !cfg.c2_fft;
    if (int rc = class_cfg(cfg, p.nb_mode, maxL, cf, nofft1, nofft2, nofft1)) return rc;      // (direct class 3: the same conditions as the other FFT-free plans)
    if (cfg.trace)
        for (int c = 0; c < NGEN; c++)
            fprintf(stderr, "[rafft] expand class %d: %d threads x %d regions per workgroup, P <= %d, branches <= %d, lags <= %d, LDS %d B%s\n", c, cf[c].nt, cf[c].wpb,
                    cf[c].Pmax, cf[c].brmax, cf[c].Kmax, cf[c].lds, cf[c].nofft ? " (no FFT buffers)" : "");
    merge_target = maxL > CLS2_P / 2 ? 3 : 2;
    seen_cap0.resize(S);
    seen0_total = seen_slots0_wave(len.data(), S, p, cfg, SEEN0_BUDGET, seen_cap0.data());
    const double seen0_avg = (double)seen0_total / (double)std::max<size_t>(S, 1);
    c = plan_caps(cfg, S, sumL, p, est, seen0_avg);
    if (std::max((size_t)c.sort_cap * 8, (size_t)24 * 1024) + RL_CAP * 12 + (size_t)(p.max_stack + 4) * (sizeof(ParentInfo) + 16) + 1024 > 150 * 1024)
        return fail(RAFFT_ERR_PARAM, "max_branch + 2*max_stack too large for the LDS-resident beam sort");
    B = (size_t)p.max_stack;

    // Buffers are allocated for the wave that queued batches could be merged into (the scheduler folds up to merge_cap()
    // sequences of equal-parameter batches as one wave), not just for this one: a hipFree + hipMalloc of a multi-GB arena
    // in the middle of a stream of batches stalls every queue for tens of ms - and now and then for SECONDS (measured: one
    // hipMalloc of 7 GB took 2.2 s while another wave's kernels were running; a bench run that met it fell from 290 k to 10 k
    // sequences/s).  Only when that reserve is small against the HBM (the benchmark set: 4.4 GB -> 21.6 GB per workspace, two
    // workspaces for bulk waves: 15 % of the card).
    size_t Sr = S;
    Caps cr = c;
    if (S < merge_cap() && !seam) {
        // (a long-tail job has a few sequences per batch: sized once for 64 of them, whatever gets merged later)
        // bulk batches: for the merge cap itself (a stream of small batches is merged up to it whatever their size), or for five of
        // them when that is too much
        const double reserve_frac = cfg.reserve_frac;
        const size_t tries[2] = {S >= 256 ? merge_cap() : std::max<size_t>(S, std::min<size_t>(64, 32 * S)), S >= 256 ? std::min(merge_cap(), 5 * S) : S};
        Sr = S;
        for (size_t want : tries) {
            if (want <= S) continue;
            Caps big = plan_caps(cfg, want, (size_t)((double)sumL * (double)want / (double)S), p, est, seen0_avg);
            if (big.bytes <= (size_t)((double)::g.hbm_total * reserve_frac)) { cr = big; Sr = want; break; }
        }
    }
    reserve = (double)Sr / (double)S;
    const size_t sumLr = Sr == S ? sumL : (size_t)((double)sumL * (double)Sr / (double)S);
#define ENS(buf, bytes) do { if (int rc_ = ensure(g.buf, (bytes))) return rc_; } while (0)
    ENS(codes, sumLr + 16); ENS(seq_off, Sr * 4); ENS(seq_len, Sr * 4);
    ENS(beam, Sr * B * 4); ENS(beam_n, Sr * 4); ENS(done, Sr * 4); ENS(nsteps, Sr * 4);
    ENS(ch_parent, Sr * c.ch_cap * 2); ENS(ch_combo, Sr * c.ch_cap * 8); ENS(ch_dcal, Sr * c.ch_cap * 4); ENS(ch_h, Sr * c.ch_cap * 16);
    ENS(seen, cr.seen * 16); ENS(seen_off, Sr * 8); ENS(seen_cap, Sr * 4); ENS(seen_cnt, Sr * 4);
    ENS(st, cr.st * sizeof(StRec)); ENS(prod, cr.nd * 16);
    ENS(nd, cr.nd * sizeof(NodeRec)); ENS(nlist, cr.nd * 4); ENS(nd_slot, cr.nd * 4); ENS(cslot, cr.cand * 8);
    ENS(pos, cr.pos * 2); ENS(br, cr.br * 4); ENS(sp, cr.sp * 4); ENS(cand, cr.cand * 32);
    ENS(looptab, cr.looptab * 8);
    ENS(trec, cr.trec * 16); ENS(tsid, cr.tsid * 4);
    ENS(work0, cr.work * 4); ENS(work1, cr.work * 4); ENS(work2, cr.work * 4); ENS(work3, cr.work * 4); ENS(work4, cr.work * 4); ENS(work5, cr.work * 4);
    ENS(mat, cr.mat * sizeof(MatRec));
    ENS(counters, sizeof(Counters));
#undef ENS


    memset(&d, 0, sizeof d);
    d.T = ::g.T; d.tw = ::g.tw; d.S = (int)S;
    d.codes = (const uint8_t *)g.codes.p; d.seq_off = (const int *)g.seq_off.p; d.seq_len = (const int *)g.seq_len.p;
    d.K = p.nb_mode; d.B = p.max_stack; d.max_branch = p.max_branch; d.min_hp = p.min_hp; d.traj = p.traj;
    d.min_nrj = p.min_nrj; d.gc = p.gc_wei; d.au = p.au_wei; d.gu = p.gu_wei;
    // identical loops share one expansion only when the energy filter cannot depend on the
    // parent's absolute energy through float32 rounding, i.e. for the default min_nrj == 0
    d.memo = (p.min_nrj == 0.0) ? 1 : 0;
    if (cfg.no_memo) d.memo = 0;
    if (cfg.force_fft) d.force_fft = 1;   // tests: FFT path for short regions too
    d.rl_cap = RL_CAP;
    longseq = maxL > LDS_SEQ;
    d.pos_packed = longseq ? 0 : 1;          // 12 bits of position leave room for the base code (Dev::pos_packed)
    // 64 productive regions per structure (no BASELINE workload has more: the configs[3] shard - 3000 nt, ms=200 - folds without a
    // regrowth); the lists live in materialize_kernel's LDS: with 256 entries a CU holds 19 of its workgroups instead of 20 at
    // 86 VGPRs (measured: 68.4 -> 62.2 ms per 36 benchmark batches), with 1024 entries 8 (1.5 -> 2.3 ms per batch) - so the long
    // lists are for sequences beyond 4096 nt and for a wave that overflowed the short ones and is being folded again
    d.max_prod = (longseq || big_prod) ? MAX_PROD_LONG : MAX_PROD;
    if (cfg.test_max_prod > 0 && !big_prod && !longseq) d.max_prod = std::max(1, std::min(cfg.test_max_prod, MAX_PROD));   // test hook: short lists overflow early
    if (longseq) {       // scratch of the class for regions beyond 4096 positions: lag values (fp64) + lag column, per workgroup
        d.big_stride = (size_t)2 * cf[0].nmax + (size_t)2 * cf[0].nmax / 4;
        if (int rc = ensure(g.big, (size_t)cf[0].grid * d.big_stride * 8)) return rc;
        d.big_keyv = (double *)g.big.p;
    } else if (cf[3].direct3) {      // ... and of class 3 when it runs without FFT buffers: FFT size 8192 at most
        d.big_stride = (size_t)MAX_P + (size_t)MAX_P / 4;
        if (int rc = ensure(g.big, (size_t)cf[3].grid * d.big_stride * 8)) return rc;
        d.big_keyv = (double *)g.big.p;
    }
    d.cls1_P = cls1_P(cfg); d.cls1_br = cf[1].brmax;
    d.c3_switch = cfg.c3_switch >= 0 ? cfg.c3_switch : ::g.n_cu;
    d.cand_slab = std::max(16, cfg.slab);
    d.fetch_bulk = std::max(1, cfg.fetch);
    d.taper_pct = std::max(0, std::min(100, cfg.taper));
    // wide classes: regions of up to 1024 positions are correlated by the exact direct form on multi-word bit masks, longer ones
    // by the LDS FFT (measured on the configs[3] shard: n <= 1024 direct 219.7 ms against 222.8 with the FFT everywhere, 236.4
    // with the direct form up to 4096 - scipy itself switches at 2381, rafft/utils.py:121).  RAFFT_DIRECT_N moves the limit.
    d.direct_n = cfg.direct_n;
    // small-region classes (expand_small_kernel): packed positions (no sequence beyond 4096 nt), the bit-mask form of
    // window_slide (non-negative weights, no forced FFT).  RAFFT_SMALL="n4,n5" moves the limits ("0,0": off).
    d.sm_n4 = std::max(0, std::min(cfg.small_n4, 16)); d.sm_n5 = std::max(d.sm_n4, std::min(cfg.small_n5, 32));
    if (!d.pos_packed || d.force_fft || !(p.gc_wei >= 0.0 && p.au_wei >= 0.0 && p.gu_wei >= 0.0)) d.sm_n4 = d.sm_n5 = 0;
    d.mat_tile = 64;
    d.mat_tile = std::max(1, std::min(cfg.mat_tile, 64));                  // tests: several tiles per structure
    if (cfg.rl_cap >= 0) d.rl_cap = std::min(cfg.rl_cap, RL_CAP);         // tests: region lists not resident in LDS
    d.beam = (int *)g.beam.p; d.beam_n = (int *)g.beam_n.p; d.done = (int *)g.done.p; d.nsteps = (int *)g.nsteps.p;
    d.ch_cap = c.ch_cap;
    d.ch_parent = (uint16_t *)g.ch_parent.p; d.ch_combo = (uint64_t *)g.ch_combo.p; d.ch_dcal = (int *)g.ch_dcal.p; d.ch_h = (uint64_t *)g.ch_h.p;
    d.seen = (uint64_t *)g.seen.p; d.seen_cap_total = c.seen;
    d.seen_off = (uint64_t *)g.seen_off.p; d.seen_cap = (uint32_t *)g.seen_cap.p; d.seen_cnt = (uint32_t *)g.seen_cnt.p;
    d.st_cap = (uint32_t)c.st;
    d.st = (StRec *)g.st.p;
    d.prod = (ProdEnt *)g.prod.p; d.prod_shard_cap = c.nd / NSHARD;
    d.nd_cap = (uint32_t)c.nd;
    d.nd = (NodeRec *)g.nd.p; d.nlist = (int *)g.nlist.p; d.nd_slot = (uint32_t *)g.nd_slot.p; d.cslot = (unsigned long long *)g.cslot.p;
    d.looptab = (unsigned long long *)g.looptab.p; d.looptab_cap = c.looptab;
    d.pos = (uint16_t *)g.pos.p; d.pos_cap = c.pos; d.br = (uint32_t *)g.br.p; d.br_cap = c.br;
    d.sp = (uint32_t *)g.sp.p; d.sp_cap = c.sp;
    d.cand = (Cand *)g.cand.p; d.cand_cap = c.cand;
    d.trec = (int4 *)g.trec.p; d.trec_cap = (uint32_t)c.trec; d.tsid = (int *)g.tsid.p; d.tsid_cap = c.tsid;
    d.work[0] = (int *)g.work0.p; d.work[1] = (int *)g.work1.p; d.work[2] = (int *)g.work2.p; d.work[3] = (int *)g.work3.p;
    d.work[4] = (int *)g.work4.p; d.work[5] = (int *)g.work5.p; d.work_cap = (uint32_t)c.work;
    d.mat = (MatRec *)g.mat.p; d.mat_cap = (uint32_t)c.mat;
    d.c = (Counters *)g.counters.p;
    d.nd_base = S; d.nd_shard_cap = (c.nd - S) / NSHARD;
    d.pos_base = sumL; d.pos_shard_cap = (c.pos - sumL) / NSHARD;
    d.sp_shard_cap = c.sp / NSHARD;
    d.br_shard_cap = c.br / NSHARD; d.cand_shard_cap = c.cand / NSHARD;
    if (seam) d.dbg = seam->dbg;
    d.rep = cfg.rep;
    static unsigned long long *prof_buf = nullptr;
    if (cfg.trace >= 3) {
        if (!prof_buf) HIPCHK(hipMalloc((void **)&prof_buf, 128));
        HIPCHK(hipMemset(prof_buf, 0, 128));
        d.prof = prof_buf;
        int best = 0;
        for (size_t i = 0; i < S; i++) if (len[i] > len[best]) best = (int)i;
        d.prof_seq = best;
        if (cfg.prof_seq != INT_MIN) d.prof_seq = cfg.prof_seq;
        static unsigned long long *ws_buf = nullptr; static size_t ws_cap = 0;
        if (ws_cap < S) { if (ws_buf) HIPCHK(hipFree(ws_buf)); HIPCHK(hipMalloc((void **)&ws_buf, S * 24)); ws_cap = S; }
        HIPCHK(hipMemset(ws_buf, 0, S * 24));
        d.prof_ws = ws_buf;
        static unsigned long long *pe_buf = nullptr;
        if (!pe_buf) HIPCHK(hipMalloc((void **)&pe_buf, NCLS * PROF_E * 8));
        HIPCHK(hipMemset(pe_buf, 0, NCLS * PROF_E * 8));
        d.prof_e = pe_buf;
    }


    const double ms_plan = since(tw0);
    hipStream_t st = g.stream;
    // (RAFFT_TRACE: a call of this section that keeps the scheduler thread for more than a millisecond is named - every wave in flight waits)
    double t_mark = ms_plan;
    auto slow_call = [&](const char *what) {
        if (!cfg.trace) return;
        const double now = since(tw0);
        if (now - t_mark > 1.0) fprintf(stderr, "[rafft] setup of a wave of %zu sequences: %s kept the host for %.3f ms\n", S, what, now - t_mark);
        t_mark = now;
    };
    memset(&hc, 0, sizeof hc);
    hc.n_struct = S; hc.seen_top = seen0_total;
    memcpy((char *)stage.p + st_ctr, &hc, sizeof hc);
    {
        // the sequences' initial `seen` tables, back to back (seen_slots0)
        uint64_t *so = (uint64_t *)((char *)stage.p + st_soff);
        uint32_t *sc = (uint32_t *)((char *)stage.p + st_scap);
        size_t o = 0;
        for (size_t i = 0; i < S; i++) { so[i] = o; sc[i] = seen_cap0[i]; o += seen_cap0[i]; }
        // base codes | offsets | lengths | seen-table offsets | sizes | counters image: one kernel reads them out of the pinned chunk
        StageIn si;
        memset(&si, 0, sizeof si);
        auto seg = [&](size_t off, void *dst, size_t bytes) {
            si.src[si.n] = (const uint32_t *)((const char *)stage.p + off); si.dst[si.n] = (uint32_t *)dst; si.words[si.n] = (bytes + 3) / 4; si.n++;
        };
        seg(st_codes, g.codes.p, sumL + 16); seg(st_off, g.seq_off.p, S * 4); seg(st_len, g.seq_len.p, S * 4);
        seg(st_soff, g.seen_off.p, S * 8); seg(st_scap, g.seen_cap.p, S * 4); seg(st_ctr, g.counters.p, sizeof hc);
        const size_t words = (sumL + 16 + 3) / 4;
        hipLaunchKernelGGL(stage_in_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 1024)), dim3(256), 0, st, si);
        HIPCHK(hipGetLastError());
    }
    slow_call("the launch of stage_in_kernel");
    if (d.memo) HIPCHK(hipMemsetAsync(g.looptab.p, 0, c.looptab * 8, st));
    slow_call("the memset of the loop table");
    HIPCHK(hipMemsetAsync(g.seen.p, 0, seen0_total * 16, st));   // first region of every sequence; later regions are zeroed on allocation
    slow_call("the memset of the seen tables");
    hipLaunchKernelGGL(init_roots_kernel, dim3((unsigned)S), dim3(64), 0, st, d);
    HIPCHK(hipGetLastError());
    slow_call("the launch of init_roots_kernel");
    if (seam) HIPCHK(hipStreamSynchronize(st));      // (the seam overwrites the root region with synchronous copies right after)
    // beam_step_kernel LDS: time-shared region 0 (walk scratch 24 B/thread, then sort keys), region list, per-member records
    for (int v = 0; v < 2; v++) {
        const size_t nt = v ? 1024 : 256;
        bs_lds[v] = std::max((size_t)c.sort_cap * 8, 24 * nt) + RL_CAP * 12 + B * sizeof(ParentInfo) + (B + 1) * 8 + ((B + 3) & ~(size_t)3) * 4 + 128 + B * 4;      // (+ B ints: the prepass's first node-list entries)
    }
    mat_lds = 20 * (size_t)d.max_prod;
    out_row_lds = ((size_t)maxL + 15) & ~(size_t)15;      // output_kernel builds a dot-bracket row in LDS: the longest sequence of the wave
    if (cfg.dedupe_per_cu > 0) dedupe_per_cu = (unsigned)cfg.dedupe_per_cu;
    n_active = (unsigned)S;
    {
        // step-ahead (RAFFT_STEP_AHEAD=0: lock-step as in rounds 1-4): needs the kernel that takes the device's own count
        // (materialize_team_kernel: short productive-region lists, no phase stamps), and no per-step diagnostics that read counters back
        // MEASURED AND LEFT OFF (round 5, tools/ab_stepahead.sh, tools/single_probe.py): the host's round trip is not what a folding step
        // waits for.  One synchronous call on the benchmark batch takes 9.8-10.9 ms either way (it is bound by its kernels), the
        // pipelined rate falls 3 % (an empty step per wave, guessed grids), and a lone 76-nt sequence takes 1.07 ms instead of 0.91
        // (its 6 steps cost ~150 us each on the DEVICE - eight dependent kernels at ~20 us of dispatch latency - and step-ahead
        // adds a seventh).  RAFFT_STEP_AHEAD=1 switches it on.
        step_ahead = cfg.step_ahead != 0 && cfg.mat4 != 0 && !seam && cfg.trace < 2 && d.prof_e == nullptr && d.max_prod <= MAT4_PROD && cfg.test_ovf_at < 0;
    }
    ms_setup = since(tw0);
    if (cfg.trace) fprintf(stderr, "[rafft] setup: encode %.3f ms, plan+buffers %.3f ms, copies+init %.3f ms\n", ms_enc, ms_plan - ms_enc, ms_setup - ms_plan);
    tw1 = std::chrono::steady_clock::now();
    return 0;
}

// expand (three size classes on their own streams) -> beam step -> asynchronous read-back of the hot counters
int Wave::issue_step()
{
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double &a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } acc_{ms_issue, t_in};
    hipStream_t st = g.stream;
    const unsigned wide_below = (unsigned)std::max(0, cfg.wide_below);
    const bool serial = cfg.serial != 0;
    const unsigned small_wg_per_cu = (unsigned)std::max(1, cfg.small_wg);   // 4 wavefronts each
    const size_t hot_len = offsetof(Counters, node);
    HIPCHK(hipEventRecord(g.ev_fork, st));
    Span wall{next_event(), next_event(), 4};
    SPAN_REC(wall.a, st, 4);
    static const int order_big_first[NCLS] = {3, 0, 2, 1, 5, 4}, order_small_first[NCLS] = {5, 4, 3, 0, 2, 1};    // big-LDS classes first
    const int *order = cfg.small_first ? order_small_first : order_big_first;
    for (int oi = 0; oi < NCLS; oi++) {
        const int cls = order[oi];
        if (cls == 0 && !longseq) continue;                      // regions beyond 4096 positions: only sequences longer than that have them
        if (cls == 3 && merge_target == 2) continue;             // no sequence long enough for a region of that class
        if (merged_now == 3 && cls != 3 && cls != 0) continue;   // the dedupe of the last step sent everything to one class
        if (merged_now == 2 && cls == 1) continue;               // ... or the one-wavefront class to the 256-thread one
        const bool small_step0 = cfg.small_step0 != 0;      // diagnostic: empty launches (their fixed cost)
        if (cls >= NGEN && (merged_now != 0 || (steps == 0 && !small_step0) || (cls == 4 ? d.sm_n4 : d.sm_n5) == 0 || (cls == 5 && d.sm_n5 == d.sm_n4))) continue;   // small-region classes: off, or nothing was sent there
        const bool inline_ = serial || (merged_now == 3 && !longseq);   // a single kernel: no fork/join through another stream
        hipStream_t cs = inline_ ? st : g.cls_stream[cls];
        if (!inline_) HIPCHK(hipStreamWaitEvent(cs, g.ev_fork, 0));
        Span sp{next_event(), next_event(), 10 + cls};
        SPAN_REC(sp.a, cs, sp.kind);
        // persistent workgroups loop over the work list, so any grid is correct: when few structures were
        // materialized (the tail of a batch) a small grid avoids dispatching thousands of empty workgroups
        unsigned grid = cls >= NGEN ? (unsigned)::g.n_cu * small_wg_per_cu : (unsigned)cf[cls].grid;
        const unsigned c1_wgs = (unsigned)std::max(0, cfg.c1_wgs);     // A/B: fewer workgroups of the one-wavefront class
        if (cls == 1 && c1_wgs && cf[1].wpb > 1) grid = std::min(grid, c1_wgs * (unsigned)cf[1].wpb);
        if (steps > 0 && !(step_ahead && steps < 3)) {       // (step-ahead: `last_mat` is the step before's - doubled; the first steps grow faster)
            const unsigned long long bound = (unsigned long long)last_mat * (step_ahead ? 2ULL : 1ULL) * (cls == 1 ? 8ULL : 4ULL) + 32ULL;
            if (bound < grid) grid = (unsigned)bound;
        }
        if (int rc = launch_expand_cls(cfg, d, cls, cf, grid, cs)) return rc;
        const int twice = cfg.twice;   // diagnostic: the same work again, caches warm
        if (twice && cls == 1) {
            HIPCHK(hipMemsetAsync((char *)g.counters.p + offsetof(Counters, wcur) + sizeof(ShardCtr) * NSHARD * cls, 0, sizeof(ShardCtr) * NSHARD, cs));
            HIPCHK(hipMemsetAsync((char *)g.counters.p + offsetof(Counters, wdone) + 8 * cls, 0, 8, cs));
            if (int rc = launch_expand_cls(cfg, d, cls, cf, grid, cs, twice >= 2)) return rc;
        }
        if (cls == 1) bt.stats.n_expand_launches++;       // launches of the dominant kernel (ms_expand is their sum)
        SPAN_REC(sp.b, cs, sp.kind);
        spans.push_back(sp);
        if (!inline_) {
            HIPCHK(hipEventRecord(g.ev_join[cls], cs));
            HIPCHK(hipStreamWaitEvent(st, g.ev_join[cls], 0));
        }
    }
    SPAN_REC(wall.b, st, 4);
    spans.push_back(wall);
    {
        Span sp{next_event(), next_event(), 1};
        SPAN_REC(sp.a, st, sp.kind);
        // few sequences left (the long ones): a 1024-thread workgroup per sequence shortens the serial
        // chains (16 wavefronts for the prepass, 1024 combos per chunk); many sequences: 256 threads
        const bool bs_prod = d.prof == nullptr && d.prof_ws == nullptr;          // no diagnostic stamps asked for: the production builds
        if (n_active < wide_below) {
            if (bs_prod) hipLaunchKernelGGL((beam_step_kernel<1024, true>), dim3((unsigned)S), dim3(1024), bs_lds[1], st, d, c.sort_cap);
            else hipLaunchKernelGGL((beam_step_kernel<1024, false>), dim3((unsigned)S), dim3(1024), bs_lds[1], st, d, c.sort_cap);
        } else {
            if (bs_prod) hipLaunchKernelGGL((beam_step_kernel<256, true>), dim3((unsigned)S), dim3(256), bs_lds[0], st, d, c.sort_cap);
            else hipLaunchKernelGGL((beam_step_kernel<256, false>), dim3((unsigned)S), dim3(256), bs_lds[0], st, d, c.sort_cap);
        }
        HIPCHK(hipGetLastError());
        SPAN_REC(sp.b, st, sp.kind);
        spans.push_back(sp);
    }
    steps++;
    HIPCHK(hipMemcpyAsync((char *)g.hot + 1024 * (rb_issued & 1), g.counters.p, hot_len, hipMemcpyDeviceToHost, st));   // pinned: truly asynchronous
    HIPCHK(hipEventRecord(g.ev_hot[rb_issued & 1], st));
    rb_issued++;
    t_issued = std::chrono::steady_clock::now();
    return 0;
}

// materialize the new beam members of the step whose beam step is queued or done, and find the loops among their regions that
// are known already.  `n_mat_known`: the step's count when its counters have been read back, 0 when they have not (step-ahead mode:
// the kernel takes the device's own counter and `n_mat_guess` only sizes its grid and chooses the size classes of the next step)
int Wave::issue_materialize(unsigned n_mat_known, unsigned n_mat_guess)
{
    hipStream_t st = g.stream;
    const unsigned nm = n_mat_known ? n_mat_known : n_mat_guess;
    Span sp{next_event(), next_event(), 2};
    SPAN_REC(sp.a, st, sp.kind);
    // (four structures per wavefront, teams of 16 lanes, when the short productive-region lists are in use -
    //  materialize_team_kernel; RAFFT_MAT4=0: one structure per wavefront)
    const bool mat4_on = cfg.mat4 != 0;
    if (mat4_on && d.prof_e == nullptr && d.max_prod <= MAT4_PROD) {
        const unsigned grid = n_mat_known ? (n_mat_known + MAT4_TEAMS - 1) / MAT4_TEAMS
                                          : std::min<unsigned>((unsigned)((c.mat + MAT4_TEAMS - 1) / MAT4_TEAMS), std::max<unsigned>(64u, nm / 2u + 64u));
        hipLaunchKernelGGL(materialize_team_kernel, dim3(grid), dim3(64), 0, st, d, n_mat_known ? (int)n_mat_known : -1);
    }
    else if (d.prof_e == nullptr) hipLaunchKernelGGL(materialize_kernel<true>, dim3(n_mat_known), dim3(MAT_NT), mat_lds, st, d);
    else hipLaunchKernelGGL(materialize_kernel<false>, dim3(n_mat_known), dim3(MAT_NT), mat_lds, st, d);
    HIPCHK(hipGetLastError());
    // tail of the batch: so few new structures that their regions fit one wave of workgroups of the widest class
    // (measured on the benchmark batch: 18.8 -> 17.3 ms; thresholds in new structures per step, per CU)
    // (round 3, after the 256-thread class got its production build and four workgroups per CU: everything goes to the widest
    //  class only below 2 structures per CU - 16 per CU before; a burst of 20 shard batches 18.5 -> 17.8 ms, the rest unchanged)
    const unsigned merge_below = cfg.merge_below >= 0 ? (unsigned)cfg.merge_below : 2u * (unsigned)::g.n_cu;
    const unsigned merge2_below = cfg.merge2_below >= 0 ? (unsigned)cfg.merge2_below : 128u * (unsigned)::g.n_cu;
    d.merge_cls = seam ? 0 : nm < merge_below ? merge_target : nm < merge2_below ? 2 : 0;
    merged_now = d.merge_cls;
    // (a thread per region created in this step - two or three per new structure: light steps launch a handful of workgroups instead
    //  of two per CU, which used to queue behind the expand kernels of the other waves only to find nothing)
    const unsigned dd_grid = std::min<unsigned>((unsigned)::g.n_cu * dedupe_per_cu, (unsigned)std::min<unsigned long long>(0x7fffffffULL, (unsigned long long)nm * 4ULL / DEDUPE_NT + 2ULL));
    hipLaunchKernelGGL(dedupe_kernel, dim3(dd_grid), dim3(DEDUPE_NT), 0, st, d);
    HIPCHK(hipGetLastError());
    SPAN_REC(sp.b, st, sp.kind);
    spans.push_back(sp);
    return 0;
}

// the beam step of this wave has finished: stop, or materialize the new beam members and go on
int Wave::after_beam()
{
    const auto t_in = std::chrono::steady_clock::now();
    struct Acc { double &a; std::chrono::steady_clock::time_point t; ~Acc() { a += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } acc_{ms_after, t_in};
    hipStream_t st = g.stream;
    if (draining) return finish_done();            // the last rows have landed
    const size_t hot_len = offsetof(Counters, node);
    memcpy(&hc, (const char *)g.hot + 1024 * (rb_seen & 1), hot_len);
    rb_seen++;
    if (hc.overflow) { ovf = hc.overflow; return finish(); }
    // test hook: pretend an arena overflowed at this step of the first attempt (regrowth late in a wave)
    if (cfg.test_ovf_at >= 0 && depth == 0 && steps == cfg.test_ovf_at) { ovf = OVF_STRUCT; return finish(); }
    if (hc.n_mat == 0) return finish();
    n_active = (unsigned)S - hc.n_done;
    last_mat = hc.n_mat;
    // most sequences of the wave have finished: their rows leave beside the folding steps of the others - once this step's kernels
    // are queued (below): the host's share of it, a millisecond or two for a wave of 16 k sequences, is off the wave's own path
    const bool harvest_now = !p.traj && !seam && !harvested && S >= 256 && (size_t)hc.trec_n * 10 >= S * 7 && !cfg.no_harvest;
    const size_t harvest_n = (size_t)hc.trec_n;
    // step-ahead mode: this step's materialize and the next folding step are queued already (when the step before was looked at);
    // what is issued now is the materialize of the step in flight and the folding step after it, sized by this step's counters
    if (int rc = issue_materialize(step_ahead ? 0u : hc.n_mat, hc.n_mat)) return rc;
    if (cfg.trace >= 2) {
        Counters h2;
        HIPCHK(hipMemcpyAsync(&h2, g.counters.p, hot_len, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        fprintf(stderr, "[rafft] step %d: n_mat %u -> work %u %u %u %u | small %u %u\n", steps, hc.n_mat, h2.n_work[0].v, h2.n_work[1].v, h2.n_work[2].v, h2.n_work[3].v, h2.n_work[4].v, h2.n_work[5].v);
    }
    const int step_of_harvest = steps;
    if (int rc = issue_step()) return rc;
    if (harvest_now && !finished) {
        if (int rc = emit_rows(0, harvest_n, true, nullptr)) return rc;
        harvested = harvest_n;
        if (cfg.trace) fprintf(stderr, "[rafft] early harvest after step %d: %zu of %zu sequences\n", step_of_harvest, harvested, S);
    }
    return 0;
}

// Format the beams of trajectory records [first, first + count) as result rows on the device and copy them to a
// pinned chunk of their own.  `early`: on the copy stream, while the wave goes on folding (only the records of
// sequences that have finished are final, so this is used without --traj, where a sequence has one record).
int Wave::emit_rows(size_t first, size_t count, bool early, double *t_gather)
{
    const auto t0_ = std::chrono::steady_clock::now();
    hipStream_t st = early ? g.copy_stream : g.stream;
    Buf &b_rec = early ? g.row_off2 : g.row_off, &b_db = early ? g.out_db2 : g.out_db, &b_dc = early ? g.out_dcal2 : g.out_dcal;
    std::vector<int4> trec(count);
    if (count) HIPCHK(hipMemcpy(trec.data(), (const int4 *)g.trec.p + first, count * sizeof(int4), hipMemcpyDeviceToHost));
    // records in (sequence, step) order; rows are laid out record after record.  (Without --traj a sequence has ONE record: any order
    // of the records will do, and sorting 16 k of them is a millisecond of the scheduler thread.)
    if (p.traj) std::sort(trec.begin(), trec.end(), [](const int4 &a, const int4 &b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    std::vector<OutRec> &recs = early ? early_recs : late_recs;     // (members: they outlive the asynchronous upload)
    recs.assign(trec.size(), OutRec{});
    long long tot_bytes = 0;
    size_t nrows = 0;
    for (size_t ri = 0; ri < trec.size(); ri++) {
        const int4 &r = trec[ri];
        recs[ri] = OutRec{tot_bytes, (int)nrows, r.w, r.z, len[r.x]};
        tot_bytes += (long long)r.z * (len[r.x] + 1);
        nrows += (size_t)r.z;
    }
    if (t_gather) *t_gather = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
    last_rows_bytes = tot_bytes;
    if (!nrows) return 0;
    const size_t dcal_off = ((size_t)tot_bytes + 63) & ~(size_t)63;
    PinBuf chunk;
    chunk = pin_acquire(dcal_off + nrows * 4 + 64);
    if (!chunk.p) return fail(RAFFT_ERR_HIP, "hipHostMalloc failed for the result buffer");
    {
        auto shared = std::make_shared<PinChunk>(chunk);
        for (auto &m : members) m->ho->chunks.push_back(shared);
    }
    char *all_db = (char *)chunk.p;
    int *all_dcal = (int *)((char *)chunk.p + dcal_off);
    if (int rc = ensure(b_rec, (size_t)((double)(recs.size() * sizeof(OutRec)) * reserve))) return rc;
    if (int rc = ensure(b_db, (size_t)((double)tot_bytes * reserve))) return rc;
    if (int rc = ensure(b_dc, (size_t)((double)(nrows * 4) * reserve))) return rc;
    HIPCHK(hipMemcpyAsync(b_rec.p, recs.data(), recs.size() * sizeof(OutRec), hipMemcpyHostToDevice, st));
    Span sp{next_event(), next_event(), 3};
    SPAN_REC(sp.a, st, sp.kind);
    unsigned grid = (unsigned)std::min<size_t>(nrows, 65536);
    hipLaunchKernelGGL(output_kernel, dim3(grid), dim3(64), out_row_lds, st, d, (int)nrows, (int)recs.size(), (const OutRec *)b_rec.p,
                       (char *)b_db.p, (int *)b_dc.p);
    HIPCHK(hipGetLastError());
    SPAN_REC(sp.b, st, sp.kind);
    spans.push_back(sp);
    HIPCHK(hipMemcpyAsync(all_db, b_db.p, (size_t)tot_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(all_dcal, b_dc.p, nrows * 4, hipMemcpyDeviceToHost, st));
    // `recs` was handed to an asynchronous copy from pageable memory: HIP stages such copies before returning.  Nobody waits
    // here: the early rows are waited for at the end of the wave, the late ones by the scheduler's poll of Workspace::ev_hot (finish)
    // per-sequence views into the chunk (the pointers are only read by the caller after the call has returned)
    for (size_t r0 = 0; r0 < recs.size();) {
        const int i = trec[r0].x;
        size_t r1 = r0;
        while (r1 < recs.size() && trec[r1].x == i) r1++;
        const int gi = seqs[i].idx;
        HostOut &out = out_of(i);
        int o = 0;
        if (r1 - r0 == 1) { out.one_size[gi] = o = recs[r0].cnt; out.one_off[gi] = 0; }      // (no heap allocation per sequence)
        else {
            auto &ss = out.step_size[gi];
            auto &so = out.step_off[gi];
            ss.resize(r1 - r0); so.resize(r1 - r0);
            for (size_t r = r0; r < r1; r++) { ss[r - r0] = recs[r].cnt; so[r - r0] = o; o += recs[r].cnt; }
        }
        out.dcal_ptr[gi] = all_dcal + recs[r0].row0;
        out.db_ptr[gi] = all_db + recs[r0].off;
        rafft_seq_result &sr = out.seq[gi];
        sr.status = RAFFT_OK; sr.length = len[i]; sr.n_steps = (int)(r1 - r0); sr.n_structs = o;
        r0 = r1;
    }
    return 0;
}

// Every step is done.  The statistics are read back, the rows that have not left yet are formatted and sent to the host - and
// the scheduler thread goes back to the other waves: it used to sit in a stream synchronize here for the 1-3 ms the rows of a
// wave of five batches take (32 MB D2H), during which no other wave's step was read back or issued.  ready() / after_beam()
// see the end of that copy through Workspace::ev_hot (finish_done).
int Wave::finish_body()
{
    hipStream_t st = g.stream;
    finished = true;
    const double ms_loop = ms_loop_ = since(tw1);
    auto tw2 = tw2_ = std::chrono::steady_clock::now();
    bt.stats.n_steps = std::max<int64_t>(bt.stats.n_steps, step_ahead ? rb_seen : steps);
    // (step-ahead: the step queued behind the last read-back has nothing to do - every sequence is done, no work list holds anything -
    //  and touches none of the counters and records read below)
    if (ovf && cfg.trace) fprintf(stderr, "[rafft] wave S=%zu est %.1f overflowed (bits %u) after %d steps, %.1f ms\n", S, est, ovf, steps, since(tw0));
    if (ovf) {
        HIPCHK(hipStreamSynchronize(st));
        if (harvested) HIPCHK(hipStreamSynchronize(g.copy_stream));
        if ((ovf & OVF_PROD) && !(ovf & OVF_SORT) && d.max_prod < MAX_PROD_LONG) {
            want_big_prod = true;          // not a limit yet: the wave is folded again with the long lists (MAX_PROD_LONG)
            return result = RAFFT_ERR_CAPACITY;
        }
        if (ovf & (OVF_PROD | OVF_SORT))
            return result = fail(RAFFT_ERR_PARAM, "structure with more than 1024 productive regions, or sort capacity exceeded");
        return result = RAFFT_ERR_CAPACITY;
    }
    // statistics (SURVEY.md 8d algorithmic bytes; only expansions the kernels really executed)
    HIPCHK(hipMemcpy(&hc, g.counters.p, sizeof hc, hipMemcpyDeviceToHost));
    for (int c = 0; c < NCLS; c++)            // the sharded statistics lines (Counters::xstat)
        for (int i = 0; i < NSHARD; i++) {
            const Counters::StatLine &x = hc.xstat[c][i];
            hc.n_expand += x.items; hc.sum_n += x.n; hc.sum_lags += x.lags; hc.sum_nbr += x.nbr;
            hc.cls_items[c] += x.items; hc.cls_sum_n[c] += x.n; hc.cls_sum_lags[c] += x.lags;
            hc.n_alias += x.alias; hc.n_children += x.children; hc.sum_struct_len += x.struct_len;
            bt.stats.n_dE_evals += (int64_t)x.evals; bt.stats.n_dE_guessed += (int64_t)x.guessed; bt.stats.n_kept_guessed += (int64_t)x.kept_guessed;
        }
    {
        unsigned long long nn = S, ni = S;
        for (int i = 0; i < NSHARD; i++) { nn += hc.node[i].v; ni += hc.nlist[i].v; }
        bt.stats.n_nodes_created += (int64_t)nn;
        bt.stats.n_node_instances += (int64_t)ni;
    }
    bt.stats.n_node_expansions += hc.n_expand;
    bt.stats.n_nodes_aliased += (int64_t)hc.n_alias;
    bt.stats.sum_node_len += hc.sum_n;
    bt.stats.sum_lags += hc.sum_lags;
    bt.stats.n_structs += (int64_t)hc.n_struct;
    bt.stats.n_children += hc.n_children;
    bt.stats.sum_struct_len += (int64_t)(hc.sum_struct_len + sumL);
    {
        int64_t ex = 3 * (int64_t)hc.sum_n + 16 * (int64_t)hc.sum_lags + 3 * (int64_t)(hc.sum_struct_len + sumL);
        // the dominant kernel (size class 1, P <= 512): its own regions, and the per-structure term in
        // proportion to the regions it expanded
        double share = hc.n_expand ? (double)hc.cls_items[1] / (double)hc.n_expand : 0.0;
        bt.stats.alg_bytes_expand += 3 * (int64_t)hc.cls_sum_n[1] + 16 * (int64_t)hc.cls_sum_lags[1] +
                                    (int64_t)(share * 3.0 * (double)(hc.sum_struct_len + sumL));
        for (int c = 0; c < NCLS; c++) {       // the same figure per size class (2, 3 + regions beyond 4096 positions, small-region classes)
            const double sh = hc.n_expand ? (double)hc.cls_items[c] / (double)hc.n_expand : 0.0;
            const int64_t v = 3 * (int64_t)hc.cls_sum_n[c] + 16 * (int64_t)hc.cls_sum_lags[c] + (int64_t)(sh * 3.0 * (double)(hc.sum_struct_len + sumL));
            if (c == 2) bt.stats.alg_bytes_expand_c2 += v;
            else if (c == 3 || c == 0) bt.stats.alg_bytes_expand_c3 += v;
            else if (c >= NGEN) bt.stats.alg_bytes_expand_small += v;
        }
        bt.stats.alg_bytes_beam += 2 * (int64_t)hc.sum_struct_len + 8 * (int64_t)(hc.n_struct - S);
        bt.stats.alg_bytes_expand_all += ex;
        bt.stats.alg_bytes += ex + 2 * (int64_t)hc.sum_struct_len + 8 * (int64_t)(hc.n_struct - S);
    }


    const double tl_stats = since(tw2);
    // ---- the rows that have not left yet: records of sequences that finished after the early harvest (or all)
    double tl_gather = 0;
    if (int rc = emit_rows(harvested, (size_t)hc.trec_n - harvested, false, &tl_gather)) return rc;
    tl_gather += tl_stats;
    tl_stats_ = tl_stats; tl_gather_ = tl_gather;
    if (harvested) {                                  // the early rows went through the copy stream: one event covers both
        HIPCHK(hipEventRecord(g.ev_copy, g.copy_stream));
        HIPCHK(hipStreamWaitEvent(st, g.ev_copy, 0));
    }
    // (a step queued ahead of the last read-back has nothing to do - every sequence is done - and its read-back is never looked at)
    rb_seen = rb_issued;
    HIPCHK(hipEventRecord(g.ev_hot[rb_issued & 1], st));
    rb_issued++;
    finished = false; draining = true;
    (void)ms_loop;
    return 0;
}

int Wave::finish_done_body()
{
    finished = true;
    const auto tw2 = tw2_;
    const double ms_loop = ms_loop_, tl_stats = tl_stats_, tl_gather = tl_gather_;
    const double tl_copy = since(tw2);
    const long long tot_bytes = last_rows_bytes;
    if (d.prof_e) {
        unsigned long long pe[NCLS * PROF_E];
        HIPCHK(hipMemcpy(pe, d.prof_e, sizeof pe, hipMemcpyDeviceToHost));
        static const char *nm[8] = {"fetch+header", "LDS fill", "FFT", "lag values", "ranking", "window_slide", "dE", "emit"};
        {
            static const char *mn[7] = {"header+list+digits", "pass 1 (sizes)", "allocation", "parent row", "pass 2 descriptors", "region copies", "row out"};
            unsigned long long t = 0;
            for (int k = 0; k < 7; k++) t += pe[k];
            fprintf(stderr, "[rafft] materialize phase shares (%llu Mcycles):", t / 1000000);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f%%", mn[k], t ? 100.0 * (double)pe[k] / (double)t : 0.0);
            fprintf(stderr, "\n");
        }
        for (int c = NGEN; c < NCLS; c++) {
            static const char *sn[7] = {"fetch", "header+fill+masks", "window_slide", "branch prefix sums", "dE", "values+compaction+slots", "order+emit"};
            unsigned long long t = 0;
            for (int k = 0; k < 7; k++) t += pe[c * PROF_E + k];
            fprintf(stderr, "[rafft] small-region class %d (lane 0 of every wavefront, %llu Mcycles, %llu rounds by %llu wavefronts = %.1f kcycles per round):", c, t / 1000000,
                    pe[c * PROF_E + 8], pe[c * PROF_E + 9], pe[c * PROF_E + 8] ? (double)t / (double)pe[c * PROF_E + 8] / 1e3 : 0.0);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f%%", sn[k], t ? 100.0 * (double)pe[c * PROF_E + k] / (double)t : 0.0);
            fprintf(stderr, "\n");
        }
        for (int c = 1; c < NGEN; c++) {
            unsigned long long t = 0;
            for (int k = 0; k < 8; k++) t += pe[c * PROF_E + k];
            fprintf(stderr, "[rafft] expand class %d phase shares (lane 0 of every wavefront, %llu Mcycles):", c, t / 1000000);
            for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f%%", nm[k], t ? 100.0 * (double)pe[c * PROF_E + k] / (double)t : 0.0);
            fprintf(stderr, "\n");
            unsigned long long hn = 0, hc_ = 0;
            for (int k = 0; k < 6; k++) { hn += pe[c * PROF_E + 8 + k]; hc_ += pe[c * PROF_E + 16 + k]; }
            static const char *bn[6] = {"n<=8", "<=16", "<=32", "<=64", "<=128", ">128"};
            fprintf(stderr, "[rafft] expand class %d by region size (share of regions / share of cycles / kcycles per region):", c);
            for (int k = 0; k < 6; k++)
                fprintf(stderr, "  %s %.1f%% / %.1f%% / %.1f", bn[k], hn ? 100.0 * pe[c * PROF_E + 8 + k] / hn : 0.0, hc_ ? 100.0 * pe[c * PROF_E + 16 + k] / hc_ : 0.0,
                        pe[c * PROF_E + 8 + k] ? (double)pe[c * PROF_E + 16 + k] / (double)pe[c * PROF_E + 8 + k] / 1e3 : 0.0);
            fprintf(stderr, "\n");
            fprintf(stderr, "[rafft]   class %d: inside fetch+header: claiming items %llu Mcycles, work-list entry %llu Mcycles (of %llu)\n", c, pe[c * PROF_E + 40] / 1000000, pe[c * PROF_E + 41] / 1000000, pe[c * PROF_E] / 1000000);
            fprintf(stderr, "[rafft]   class %d: inside dE: branch prefix sums %llu, candidates %llu Mcycles; inside emit: compaction %llu, candidate slots %llu, keys+rank %llu, hashes+cuts+stores %llu Mcycles\n", c,
                    pe[c * PROF_E + 42] / 1000000, pe[c * PROF_E + 43] / 1000000, pe[c * PROF_E + 44] / 1000000, pe[c * PROF_E + 45] / 1000000, pe[c * PROF_E + 46] / 1000000, pe[c * PROF_E + 47] / 1000000);
            if (pe[c * PROF_E + 32]) fprintf(stderr, "[rafft]   class %d: draining the previous region's stores (RAFFT_REP=256): %llu Mcycles\n", c, pe[c * PROF_E + 32] / 1000000);
            if (c == 1) {
                fprintf(stderr, "[rafft]   class 1, regions without any stem / without a kept candidate (share of the size class):");
                for (int k = 0; k < 6; k++) {
                    const double nreg = (double)pe[c * PROF_E + 8 + k];
                    fprintf(stderr, "  %s %.1f%% / %.1f%%", bn[k], nreg ? 100.0 * pe[c * PROF_E + 80 + k] / nreg : 0.0, nreg ? 100.0 * pe[c * PROF_E + 88 + k] / nreg : 0.0);
                }
                fprintf(stderr, "\n");
            }
        }
    }
    if (d.prof_ws) {
        std::vector<unsigned long long> wsv(S * 3);
        HIPCHK(hipMemcpy(wsv.data(), d.prof_ws, S * 24, hipMemcpyDeviceToHost));
        std::vector<int> ord(S);
        for (size_t i = 0; i < S; i++) ord[i] = (int)i;
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return wsv[3 * a] > wsv[3 * b]; });
        unsigned long long tot = 0, totc = 0;
        unsigned long long totp = 0, totk = 0;
        for (size_t i = 0; i < S; i++) { tot += wsv[3 * i]; totc += wsv[3 * i + 1] & 0xFFFFFF; totk += wsv[3 * i + 1] >> 24; totp += wsv[3 * i + 2]; }
        fprintf(stderr, "[rafft] beam_step per sequence: total cycles %llu, chunks %llu, combos %llu, parents walked %llu over %zu sequences\n", tot, totc, totk, totp, S);
        for (size_t k = 0; k < std::min<size_t>(S, 12); k++) {
            int i = ord[k];
            fprintf(stderr, "[rafft]   #%zu local seq %d (L=%d): cycles %llu, chunks %llu, combos %llu, parents walked %llu\n", k, i, len[i], wsv[3 * i], wsv[3 * i + 1] & 0xFFFFFF, wsv[3 * i + 1] >> 24, wsv[3 * i + 2]);
        }
    }
    if (d.prof) {
        unsigned long long pv[16];
        HIPCHK(hipMemcpy(pv, d.prof, 128, hipMemcpyDeviceToHost));
        fprintf(stderr, "[rafft]   product loop detail: head %llu, decode+lookup %llu, scans %llu, write+insert %llu; %llu growths of `seen` in the walk: before %llu, allocation %llu, zero fill %llu, rehash %llu\n",
                pv[6], pv[8], pv[9], pv[10], pv[14], pv[7], pv[11], pv[12], pv[13]);
        fprintf(stderr, "[rafft] beam_step stamps of the longest sequence (cycles): prepass %llu, product loop %llu, single phase %llu, sort %llu, survivors %llu over %llu steps\n",
                pv[0], pv[1], pv[2], pv[3], pv[4], pv[5]);
    }
    if (cfg.trace) {
        auto mx = [&](const ShardCtr *sc) { unsigned long long m = 0, t = 0; for (int i = 0; i < NSHARD; i++) { m = std::max(m, sc[i].v); t += sc[i].v; } return std::make_pair(m, t); };
        auto nd = mx(hc.node), po = mx(hc.pos), br = mx(hc.br), spr = mx(hc.sp), ca = mx(hc.cand), pr = mx(hc.prod), nl = mx(hc.nlist);
        fprintf(stderr, "[rafft] max productive regions per structure: %u (limit %d)\n", hc.max_nprod, d.max_prod);
        fprintf(stderr, "[rafft] arenas used/cap (max shard | total): st %llu/%zu  nd %llu/%llu|%llu  pos %llu/%llu|%llu  br %llu/%llu|%llu  sp %llu/%llu|%llu  cand %llu/%llu|%llu  prod %llu/%llu|%llu  nlist %llu/%llu|%llu  seen %llu/%zu  est %.1f\n",
                hc.n_struct, c.st, nd.first, (unsigned long long)d.nd_shard_cap, nd.second, po.first, (unsigned long long)d.pos_shard_cap, po.second,
                br.first, (unsigned long long)d.br_shard_cap, br.second, spr.first, (unsigned long long)d.sp_shard_cap, spr.second,
                ca.first, (unsigned long long)d.cand_shard_cap, ca.second, pr.first, (unsigned long long)d.prod_shard_cap, pr.second, nl.first, (unsigned long long)d.nd_shard_cap, nl.second, hc.seen_top, c.seen, est);
    }
    if (cfg.trace >= 2) {       // how full the `seen` sets got, by sequence length (sizes the initial tables: a growth is a rehash)
        std::vector<uint32_t> cnt(S), cap(S);
        HIPCHK(hipMemcpy(cnt.data(), d.seen_cnt, S * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cap.data(), d.seen_cap, S * 4, hipMemcpyDeviceToHost));
        const int edges[] = {0, 80, 100, 130, 200, 300, 500, 1000, 2000, 1 << 30};
        for (int e = 0; e + 1 < (int)(sizeof(edges) / sizeof(edges[0])); e++) {
            unsigned long long n = 0, sum = 0, mxc = 0, grown = 0, slots = 0;
            for (size_t i = 0; i < S; i++) if (len[i] > edges[e] && len[i] <= edges[e + 1]) { n++; sum += cnt[i]; mxc = std::max<unsigned long long>(mxc, cnt[i]); grown += cap[i] > seen_cap0[i] ? 1 : 0; slots += cap[i]; }
            if (n) fprintf(stderr, "[rafft] seen sets, %d < L <= %d: %llu sequences, mean %llu entries, max %llu, %llu grew, %llu slots at the end\n", edges[e], edges[e + 1], n, sum / n, mxc, grown, slots);
        }
    }
    if (cfg.trace) fprintf(stderr, "[rafft] host time inside issue_step %.3f ms, inside after_beam (incl. nested issue_step and this tail) %.3f ms\n", ms_issue, ms_after);
    if (cfg.trace) fprintf(stderr, "[rafft] wave S=%zu setup %.2f ms, loop %.2f ms (%d steps), tail %.2f ms (counters %.3f, records+gather %.3f, rows out %.3f incl. %.1f MB D2H)\n",
                                       S, ms_setup, ms_loop, steps, since(tw2), tl_stats, tl_gather - tl_stats, tl_copy - tl_gather, (double)tot_bytes / 1e6);
    return result = 0;
}

// rafft_expand_node: one region of one given structure through the expand kernel
int run_seam(Batch &bt, const std::vector<SeqIn> &one, const SeamIn &sm)
{
    Wave w(g.ws[0], {std::shared_ptr<Batch>(&bt, [](Batch *) {})}, one, 4.0, &sm);
    if (int rc = w.setup()) return rc;
    Workspace &W = g.ws[0];
    // overwrite the root region of sequence 0 with the given loop of the given structure
    {
        std::vector<uint16_t> packed(sm.pos);
        if (w.d.pos_packed) {
            for (auto &v : packed) v = (uint16_t)(v | (kBaseCode[(unsigned char)one[0].s[v]] << 12));
        }
        HIPCHK(hipMemcpy(W.pos.p, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    }
    if (!sm.br.empty()) {
        std::vector<uint32_t> pb(sm.br);
        if (w.d.pos_packed)      // (the base codes of a helix's outermost pair ride in bits 12-15 and 28-31)
            for (auto &v : pb) v |= ((uint32_t)kBaseCode[(unsigned char)one[0].s[v & 0xFFFFu]] << 12) | ((uint32_t)kBaseCode[(unsigned char)one[0].s[v >> 16]] << 28);
        HIPCHK(hipMemcpy(W.br.p, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
    }
    int n = (int)sm.pos.size(), nbr = (int)sm.br.size();
    {
        NodeRec root;                                       // region 0 as init_roots_kernel left it, with the given loop
        HIPCHK(hipMemcpy(&root, W.nd.p, sizeof root, hipMemcpyDeviceToHost));
        root.n = n; root.nbr = nbr; root.ci = sm.ci; root.cj = sm.cj; root.pdcal = sm.pdcal;
        HIPCHK(hipMemcpy(W.nd.p, &root, sizeof root, hipMemcpyHostToDevice));
    }
    int cls = node_class(n, (sm.ci < 0 || one[0].len > LDS_SEQ) ? one[0].len : sm.cj + 1 - sm.ci, nbr, 0, w.d.cls1_P, w.d.cls1_br, w.d.K, w.d.sm_n4, w.d.sm_n5);
    int zero = 0;
    memset(&w.hc.n_work, 0, sizeof w.hc.n_work);
    w.hc.n_work[cls].v = 1;
    HIPCHK(hipMemcpy(W.counters.p, &w.hc, sizeof w.hc, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w.d.work[cls], &zero, 4, hipMemcpyHostToDevice));
    if (int rc = launch_expand_cls(w.cfg, w.d, cls, w.cf, 1, W.stream)) return rc;
    HIPCHK(hipStreamSynchronize(W.stream));
    return 0;
}

void free_out(HostOut *o);

// ---------------------------------------------------------------------------------------------------- scheduler
// ONE thread drives every wave of every batch in flight.  A wave is a small state machine (issue_step / after_beam
// above): while the thread waits for one wave's 152-byte read-back the kernels of the others keep the GPU busy.
// Batches queue up (rafft_fold_submit); a batch is cut in a long-tail job and a bulk job (lanes 0 and 1).  What runs
// when - continuous batching:
//   * every queued job is admitted as soon as a workspace is free: the bulk of the next batch starts while the running
//     one is still folding, so the tail of a batch - and its long-tail wave - run beside the next batch's busy steps
//     (measured on the benchmark batch, three batches in flight: 11.9 ms per batch against 13.1 for synchronous
//     calls; holding the next bulk wave back until the running one has turned light - RAFFT_ADMIT_BELOW=<structures
//     per step> - was 2-4 % slower);
//   * a job that does not fit the HBM still free is split (or waits for running waves to release theirs).
struct Slot { std::unique_ptr<Wave> wave; Job job; int lane = 0; };

static unsigned admit_below()
{
    const unsigned v = (unsigned)std::max(0, g.sched_cfg.admit_below);
    return v ? v : 128u * (unsigned)g.n_cu;     // = the step size below which the one-wavefront expand class is merged away
}
// sequences one merged wave may hold (a wave of the whole benchmark set four times over folds 25 % faster per sequence
// than the set alone: fewer, fuller launches; five times over - with ten batches in flight, so that two such waves run side by
// side - another 4 % in round 3: 288 k sequences/s against 277-282 k over 30 steps; seven times over, three such waves side by side,
// 432-444 k -> 461-464 k in round 4, tools/ab_waves2.sh; beyond that the arenas of a wave pass a tenth of the HBM, where waves run
// one at a time)
static size_t merge_cap()
{
    return (size_t)std::max(1L, g.sched_cfg.merge_seqs);
}

static bool same_params(const rafft_params &a, const rafft_params &b)
{
    return a.nb_mode == b.nb_mode && a.max_stack == b.max_stack && a.max_branch == b.max_branch && a.min_hp == b.min_hp &&
           a.min_nrj == b.min_nrj && a.traj == b.traj && a.temp == b.temp && a.gc_wei == b.gc_wei && a.au_wei == b.au_wei && a.gu_wei == b.gu_wei;
}

static void finalize_batch(const std::shared_ptr<Batch> &bp)
{
    Batch &b = *bp;
    if (b.rc) {
        // early-harvest copies or kernels of a sibling wave may still be in flight: the pinned result chunks return
        // to the pool only once the device is idle
        { hipError_t e_ = hipDeviceSynchronize(); (void)e_; }
        free_out(b.ho);
        b.ho = nullptr;
    } else {
        if (b.cfg.trace) {       // per-step timeline: spans are recorded in step order
            float acc[16] = {0};
            int stepno = 0;
            for (auto &sp : b.spans) {
                float ms = 0;
                if (!span_on(sp.kind) || hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) continue;
                if (sp.kind < 16) acc[sp.kind] += ms;
                if (sp.kind == 1) {        // the beam step closes a folding step (materialize of it follows)
                    fprintf(stderr, "[rafft] t-step %2d: expand wall %.3f (c1 %.3f c2 %.3f c3 %.3f) beam %.3f  prev-materialize %.3f\n",
                            ++stepno, acc[4], acc[11], acc[12], acc[13], acc[1], acc[2]);
                    for (float &x : acc) x = 0;
                }
            }
        }
        for (auto &sp : b.spans) {
            float ms = 0;
            if (span_on(sp.kind) && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
                if (sp.kind == 14 || sp.kind == 15) b.stats.ms_expand_c1 += ms;   /* small-region classes 4 and 5 (expand_small_kernel) */
                else if (sp.kind == 10) b.stats.ms_expand_c3 += ms;   /* class 0 (regions beyond 4096 positions) rides with the widest class */
                else if (sp.kind == 11) b.stats.ms_expand += ms;   /* dominant kernel: regions with P <= 512 */
                else if (sp.kind == 12) b.stats.ms_expand_c2 += ms;
                else if (sp.kind == 13) b.stats.ms_expand_c3 += ms;
                else if (sp.kind == 4) b.stats.ms_expand_wall += ms;
                else if (sp.kind == 1) b.stats.ms_beam += ms;
                else if (sp.kind == 2) b.stats.ms_materialize += ms;
                else b.stats.ms_output += ms;
            }
        }
        HostOut *ho = b.ho;
        for (int i = 0; i < b.n_seq; i++) {
            rafft_seq_result &sr = ho->seq[i];
            const bool one = ho->step_size[i].empty();
            sr.step_size = one ? &ho->one_size[i] : ho->step_size[i].data(); sr.step_off = one ? &ho->one_off[i] : ho->step_off[i].data();
            sr.db = ho->db_ptr[i]; sr.dcal = ho->dcal_ptr[i];
        }
        ho->res.n_seq = b.n_seq; ho->res.seq = ho->seq.data(); ho->res._owner = ho;
        ho->res.n_failed = 0;
        for (int i = 0; i < b.n_seq; i++) ho->res.n_failed += ho->seq[i].status != RAFFT_OK;
        b.stats.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b.t0).count();
    }
    for (hipEvent_t e : b.events) g.ev_free.push_back(e);
    b.events.clear(); b.spans.clear();
    {
        std::lock_guard<std::mutex> lk(g.qmu);
        b.done = true;
        g.n_inflight--;
    }
    g.qcv_done.notify_all();
}

static void free_garbage()
{
    std::vector<void *> junk;
    { std::lock_guard<std::mutex> lk(g.gc_mu); junk.swap(g.garbage); }
    for (void *p : junk) { hipError_t e_ = hipFree(p); (void)e_; }
}

// Nothing in flight: workspaces that grew beyond two fifths of the HBM for some huge batch are given back (other processes
// may share the card; the next batch allocates what it needs).
static void trim_workspaces()
{
    std::unique_lock<std::mutex> lk(g.ws_mu, std::try_to_lock);
    if (!lk.owns_lock()) return;              // a seam call is using workspace 0 right now
    size_t held = 0;
    for (int i = 0; i < MAX_PIPES; i++) held += g.ws[i].bytes();
    if (held <= g.hbm_total / 5 * 2) return;       // (round 5: 2/5 of the card, a quarter until then - a stream's three bulk workspaces, sized for the merge cap at once, are 96 GB and stay)
    for (int i = 0; i < MAX_PIPES; i++) g.ws[i].release_buffers();
}

static void scheduler_main()
{
    { hipError_t e_ = hipSetDevice(g.device); (void)e_; }
    Slot slot[MAX_PIPES];
    std::deque<Job> queue[2];                 // lane 0: long-tail jobs, lane 1: bulk jobs; submission order
    int n_active_batches = 0;
    // waves in flight: two (typically the long-tail wave of one batch beside a bulk wave) - more only split the work
    // into smaller, less efficient waves (measured with 6-12 batches in flight: 2 waves 9.5-9.9 ms per benchmark batch, 3-4
    // waves 10.1-10.3 ms) and multiply the HBM held by workspaces
    // (round 4: THREE bulk waves.  A kernel trace of the pipelined loop with two - tools/concurrency.py - shows an expand kernel in flight
    //  for 70 % of the wall time; for the rest only a beam step or the small latency-bound kernels of the tails are.  A third wave fills
    //  part of that: steady state over 80 benchmark batches 367-377 k -> 384-388 k sequences/s with 15 in flight, a 20-step bench run
    //  366 -> 373 k; four waves of four batches 372-375 k - smaller waves, more launches.  Round 2 measured the opposite with waves of
    //  one or two batches and twice the kernel time per batch.)
    const Config &scfg = g.sched_cfg;          // (read by start_scheduler, before this thread was started)
    const int max_waves = std::max(1, std::min(scfg.max_waves, MAX_PIPES));
    // a member batch is finished when its last job is: finalise it
    auto release = [&](Job &job, int rc, const std::string &err) {
        for (auto &m : job.members) {
            if (rc && !m->rc) { m->rc = rc; m->err = err; }
            if (--m->pending == 0) { finalize_batch(m); n_active_batches--; }
        }
        job.members.clear();
    };
    // A job some of whose member batches have failed elsewhere keeps folding for the healthy ones: the failed members'
    // sequences are taken out (their batches are released from this job), the rest stays one job.  False: nothing left.
    auto strip_failed = [&](Job &job) -> bool {
        bool any = false;
        for (auto &m : job.members) any = any || m->rc != 0;
        if (!any) return true;
        std::vector<int> remap(job.members.size(), -1);
        std::vector<std::shared_ptr<Batch>> keep_m;
        for (size_t i = 0; i < job.members.size(); i++) {
            auto &m = job.members[i];
            if (m->rc == 0) { remap[i] = (int)keep_m.size(); keep_m.push_back(m); }
            else if (--m->pending == 0) { finalize_batch(m); n_active_batches--; }
        }
        std::vector<SeqIn> keep_s;
        for (SeqIn sq : job.seqs) if (remap[sq.bi] >= 0) { sq.bi = remap[sq.bi]; keep_s.push_back(sq); }
        job.members.swap(keep_m); job.seqs.swap(keep_s);
        return !job.members.empty() && !job.seqs.empty();
    };
    // A hard error of a wave that folds several batches (merged by the scheduler because their parameters were equal)
    // must not fail batches whose own sequences are fine: every member is folded again on its own; only a job of ONE
    // batch takes the error.
    auto fail_or_split = [&](Slot &sl, int rc, const std::string &err, std::deque<Job> *queue_) {
        if (sl.job.members.size() <= 1 || rc == RAFFT_ERR_NO_DEVICE) { release(sl.job, rc, err); return; }
        for (size_t i = 0; i < sl.job.members.size(); i++) {
            Job one;
            one.est = sl.job.est; one.depth = sl.job.depth; one.no_merge = true; one.big_prod = sl.job.big_prod;
            one.members.assign(1, sl.job.members[i]);
            for (SeqIn sq : sl.job.seqs) if (sq.bi == (int)i) { sq.bi = 0; one.seqs.push_back(sq); }
            if (one.seqs.empty()) { if (--sl.job.members[i]->pending == 0) { finalize_batch(sl.job.members[i]); n_active_batches--; } continue; }
            queue_[sl.lane].push_front(std::move(one));        // (the member's pending count moves with it)
        }
        sl.job.members.clear();
    };
    bool big_prod_seen = false;               // a wave with `big_prod_params` met a structure with more productive regions than the short lists hold
    rafft_params big_prod_params{};
    int big_prod_quiet = 0;                   // ... and this many waves in a row since then, folded with the long lists, never needed them:
                                              // after eight the short lists are back (one outlier batch does not slow the process for good)
    auto last_progress = std::chrono::steady_clock::now();
    for (;;) {
        {
            if (n_active_batches == 0) free_garbage();            // nothing in flight: the device is idle, hipFree is cheap
            std::unique_lock<std::mutex> lk(g.qmu);
            if (g.stop && n_active_batches == 0 && g.submitted.empty()) return;
            if (n_active_batches == 0 && g.submitted.empty()) {
                // idle for a second: give back workspaces that grew huge (re-allocating 80 GB costs seconds, so not between
                // back-to-back batches)
                if (!g.qcv_sched.wait_for(lk, std::chrono::seconds(1), [] { return !g.submitted.empty() || g.stop; })) {
                    lk.unlock();
                    trim_workspaces();
                    lk.lock();
                    g.qcv_sched.wait(lk, [] { return !g.submitted.empty() || g.stop; });
                }
                if (g.stop && g.submitted.empty()) return;
            }
            while (!g.submitted.empty()) {
                std::shared_ptr<Batch> bp = g.submitted.front();
                g.submitted.pop_front();
                n_active_batches++;
                bp->pending = 1;                                  // (held while its jobs are being queued)
                for (int ln = 0; ln < 2; ln++)
                    for (Job &j : bp->lane[ln]) {
                        if (j.seqs.empty()) continue;
                        j.members.assign(1, bp);
                        bp->pending++;
                        queue[ln].push_back(std::move(j));
                    }
                bp->lane[0].clear(); bp->lane[1].clear();
                if (--bp->pending == 0) { lk.unlock(); finalize_batch(bp); n_active_batches--; lk.lock(); }   // nothing foldable in it
            }
        }
        bool progressed = false;
        // ---- advance the running waves
        for (int i = 0; i < MAX_PIPES; i++) {
            Slot &sl = slot[i];
            if (!sl.wave) continue;
            const int rdy = sl.wave->ready();
            if (rdy == 0) continue;
            progressed = true;
            if (rdy < 0) {                                   // sticky device error: fail the job, free the slot
                const std::string err = g_err;
                { hipError_t e_ = hipDeviceSynchronize(); (void)e_; }
                release(sl.job, RAFFT_ERR_HIP, err);
                sl.wave.reset();
                continue;
            }
            int rc = sl.wave->after_beam();
            if (sl.wave->finished) {
                if (sl.wave->result || !rc) rc = sl.wave->result;      // (a failure on the way out that left no result keeps its own code)
                if (rc == RAFFT_ERR_CAPACITY) {
                    if (sl.job.depth >= 12)
                        release(sl.job, RAFFT_ERR_CAPACITY, "HBM arena overflow after 12 regrowths (bits " + std::to_string(sl.wave->ovf) + ")");
                    else {                                   // re-run with larger arenas, ahead of everything queued
                        sl.job.members[0]->stats.n_regrows++;
                        if (sl.wave->want_big_prod) {                                // (same arenas, longer lists)
                            sl.job.big_prod = true;
                            sl.job.members[0]->stats.n_regrows_prod++;
                            // sticky: later waves with these parameters start with the long lists instead of paying the double fold again
                            big_prod_seen = true; big_prod_params = sl.job.members[0]->p; big_prod_quiet = 0;
                        }
                        else sl.job.est *= (sl.job.depth >= 2 ? 4.0 : 2.0);
                        sl.job.depth++;
                        queue[sl.lane].push_front(std::move(sl.job));
                    }
                } else if (rc) {
                    const std::string err = g_err;
                    { hipError_t e_ = hipDeviceSynchronize(); (void)e_; }
                    sl.wave.reset();
                    fail_or_split(sl, rc, err, queue);
                } else {
                    if (sl.wave->big_prod && !sl.job.members.empty()) {
                        sl.job.members[0]->stats.n_waves_long_lists++;
                        if (!sl.job.big_prod && big_prod_seen) {       // the long lists came from the sticky flag, not from this job's own overflow
                            if (sl.wave->hc.max_nprod <= MAX_PROD) { if (++big_prod_quiet >= 8) { big_prod_seen = false; big_prod_quiet = 0; } }
                            else big_prod_quiet = 0;
                        }
                    }
                    release(sl.job, 0, "");
                }
                sl.wave.reset();
            } else if (rc) {
                const std::string err = g_err;
                { hipError_t e_ = hipDeviceSynchronize(); (void)e_; }
                sl.wave.reset();
                fail_or_split(sl, rc, err, queue);
            }
        }
        // ---- admit queued jobs: the long-tail lane first (light from the start), then the bulk lane
        int n_running = 0;
        bool heavy_running = false;
        for (int i = 0; i < MAX_PIPES; i++) if (slot[i].wave) { n_running++; heavy_running = heavy_running || slot[i].wave->heavy(admit_below()); }
        // The long-tail lane has a wave slot of its own: a long-tail wave (a handful of sequences, two dozen latency-bound steps)
        // never keeps a second bulk wave from starting (288 -> 294 k sequences/s with eight batches in flight, three interleaved
        // pairs of runs; RAFFT_TAIL_SLOT=0: the lanes share the `max_waves` slots as in round 2)
        const bool tail_slot = scfg.tail_slot != 0;
        int n_lane[2] = {0, 0};
        for (int i = 0; i < MAX_PIPES; i++) if (slot[i].wave) n_lane[slot[i].lane]++;
        // A caller that streams batches (two or more in flight) queues them microseconds apart: a bulk wave admitted the moment the
        // first one arrives would fold that one alone and the second wave whatever came in the meantime - three waves one after the
        // other (heavy phases do not overlap) where one merged wave would do.  So while submissions keep coming (the last one less
        // than RAFFT_LINGER_US = 600 us ago) and the queues are below the merge cap, both lanes wait for them.  A lone synchronous
        // call never lingers.
        const long linger_us = scfg.linger_us;
        bool linger = false;
        if (linger_us > 0 && (!queue[1].empty() || !queue[0].empty())) {
            size_t queued = 0;
            for (int ln = 0; ln < 2; ln++) for (auto &j : queue[ln]) queued = std::max(queued, j.seqs.size() * queue[ln].size());   // (a bound is enough)
            std::lock_guard<std::mutex> lk(g.qmu);
            // (the FIRST batch of a burst lingers too when it came through rafft_fold_submit - otherwise it is folded alone, in both
            //  lanes, and the long-tail lane, one wave at a time, needs two rounds: 20 shard batches 17.8 -> 13.4 ms.  rafft_fold_batch,
            //  whose caller is blocked and cannot be streaming, never lingers.)
            const long lim = (g.n_inflight >= 2 || g.last_submit_async) ? linger_us : 0;
            linger = queued < merge_cap() &&
                     std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g.t_last_submit).count() < lim;
        }
        for (int ln = 0; ln < 2; ln++) {
            if (linger) break;          // (both lanes: the long-tail jobs of a burst are merged into one wave too - the lane runs one at a time)
            while (!queue[ln].empty() && (tail_slot ? (ln == 0 ? n_lane[0] < 1 : n_lane[1] < max_waves && n_running < MAX_PIPES) : n_running < max_waves)) {
                Job &front = queue[ln].front();
                if (!strip_failed(front)) {                       // every member already failed elsewhere: nothing to fold
                    Job j = std::move(front); queue[ln].pop_front(); release(j, 0, ""); progressed = true; continue;
                }
                const bool job_heavy = front.seqs.size() >= 256;
                if (job_heavy && heavy_running) break;
                // a free workspace.  The long-tail lane keeps workspace 0 to itself and the bulk lane the others; a lane whose own are all
                // busy takes any free one - the biggest for a bulk wave, the smallest otherwise.  (Round 5: by slot parity - "this
                // lane's first" - a bulk wave now and then landed on the workspace the long-tail waves had used so far and grew all
                // its 43 buffers to bulk size: 23.6 GB of hipMalloc in the middle of a stream of batches, in one bench run out of four,
                // and one hipMalloc in a few hundred takes SECONDS - measured 3.4 s, a run of 13 k sequences/s instead of 500 k.)
                int w = -1;
                auto better = [&](int k) { return w < 0 || (job_heavy ? g.ws[k].bytes() > g.ws[w].bytes() : g.ws[k].bytes() < g.ws[w].bytes()); };
                for (int k = 0; k < MAX_PIPES; k++) if (!slot[k].wave && (tail_slot ? (k == 0) == (ln == 0) : true) && better(k)) w = k;
                if (w < 0) for (int k = 0; k < MAX_PIPES; k++) if (!slot[k].wave && better(k)) w = k;
                if (w < 0) break;
                // continuous batching: queued jobs with the same parameters join this one (first regrowths stay alone)
                Job job = std::move(front);
                queue[ln].pop_front();
                // (a sequence beyond 16 384 nt makes the class for the biggest regions plan for 32 768 positions, which lowers the
                //  wave's nb_mode limit from ~400 to 106 - class_cfg: jobs are not merged across that line)
                auto very_long = [](const Job &j) { for (auto &sq : j.seqs) if (sq.len > 16384) return true; return false; };
                const bool job_vl = very_long(job);
                while (job.depth == 0 && !job.no_merge && !queue[ln].empty()) {
                    Job &nx = queue[ln].front();
                    bool nx_failed = false;
                    for (auto &m : nx.members) nx_failed = nx_failed || m->rc != 0;
                    if (nx.depth != 0 || nx.no_merge || nx_failed || !same_params(nx.members[0]->p, job.members[0]->p) || !same_config(nx.members[0]->cfg, job.members[0]->cfg) ||
                        job.seqs.size() + nx.seqs.size() > merge_cap() || very_long(nx) != job_vl)
                        break;
                    const int off = (int)job.members.size();
                    for (SeqIn sq : nx.seqs) { sq.bi += off; job.seqs.push_back(sq); }
                    for (auto &m : nx.members) job.members.push_back(m);
                    job.est = std::max(job.est, nx.est);
                    queue[ln].pop_front();
                }
                size_t sl_ = 0;
                for (auto &sq : job.seqs) sl_ += sq.len;
                std::vector<int> lens_(job.seqs.size());
                for (size_t i = 0; i < job.seqs.size(); i++) lens_[i] = job.seqs[i].len;
                const double s0_ = (double)seen_slots0_wave(lens_.data(), lens_.size(), job.members[0]->p, job.members[0]->cfg, SEEN0_BUDGET, nullptr) / (double)std::max<size_t>(lens_.size(), 1);
                const Caps cc = plan_caps(job.members[0]->cfg, job.seqs.size(), sl_, job.members[0]->p, job.est, s0_);
                size_t others = 0;
                for (int k = 0; k < MAX_PIPES; k++) if (k != w) others += g.ws[k].bytes();
                const size_t budget = (size_t)((double)g.hbm_total * 0.55 / 2.0);
                bool fits_now = std::max(cc.bytes, g.ws[w].bytes()) + others <= (size_t)((double)g.hbm_total * 0.85);
                // waves whose arenas take more than a tenth of the HBM run one at a time (the halves of a split job would
                // otherwise fill two workspaces of that size)
                const double big_wave_frac = scfg.big_wave_frac;
                const size_t big_wave = (size_t)((double)g.hbm_total * big_wave_frac);
                if (cc.bytes > big_wave)
                    for (int k = 0; k < MAX_PIPES; k++) if (slot[k].wave && slot[k].wave->c.bytes > big_wave) fits_now = false;
                if ((cc.bytes > budget || cc.capped || (!fits_now && n_running == 0)) && job.seqs.size() > 1) {
                    const size_t h = job.seqs.size() / 2;          // too big for one wave: two jobs, one after the other
                    if (job.members[0]->cfg.trace)
                        fprintf(stderr, "[rafft] job of %zu sequences folded in halves: plan %.1f GB (budget %.1f), this workspace holds %.1f GB, the others %.1f GB, %s%s\n",
                                job.seqs.size(), (double)cc.bytes / 1e9, (double)budget / 1e9, (double)g.ws[w].bytes() / 1e9, (double)others / 1e9,
                                cc.capped ? "a table at the limit of its ids, " : "", fits_now ? "fits" : "does not fit beside what is held");
                    Job a{std::vector<SeqIn>(job.seqs.begin(), job.seqs.begin() + h), job.est, job.depth, job.members, true, job.big_prod};
                    Job c{std::vector<SeqIn>(job.seqs.begin() + h, job.seqs.end()), job.est, job.depth, job.members, true, job.big_prod};
                    for (auto &m : job.members) m->pending++;      // one job became two (halves of a split are not merged again)
                    queue[ln].push_front(std::move(c));
                    queue[ln].push_front(std::move(a));
                    progressed = true;
                    continue;
                }
                if (!fits_now && n_running > 0) { queue[ln].push_front(std::move(job)); break; }   // wait for running waves to finish
                Slot &sl = slot[w];
                sl.job = std::move(job);
                sl.lane = ln;
                int rc = init_ws(g.ws[w]);
                if (!rc) {
                    sl.wave.reset(new Wave(g.ws[w], sl.job.members, sl.job.seqs, sl.job.est));
                    sl.wave->depth = sl.job.depth;
                    sl.wave->big_prod = sl.job.big_prod || (big_prod_seen && same_params(big_prod_params, sl.job.members[0]->p));
                    if (sl.job.members[0]->cfg.trace) fprintf(stderr, "[rafft] wave of %zu sequences (lane %d) takes workspace %d (%.1f GB held)\n", sl.job.seqs.size(), ln, w, (double)g.ws[w].bytes() / 1e9);
                    rc = sl.wave->setup();
                    // A stream of batches (two or more in flight) will have `max_waves` bulk waves going at once: the bulk lane's other
                    // workspaces are brought to this one's sizes NOW, while the stream is young, instead of whenever a third wave first
                    // overlaps two others - 23.6 GB of hipMalloc at an arbitrary moment, and one hipMalloc in a few hundred takes
                    // seconds (bench.py: a run in six allocated its third workspace inside the timed region, 300 k instead of 500 k).
                    // Only workspaces that are not bulk-sized yet (less than a quarter of this one): one that merely lags behind a
                    // workspace that grew for some wave is left alone - following it would put 2 x 47 GB of hipMalloc into the stream.
                    if (!rc && job_heavy && tail_slot && g.n_inflight >= 2) {
                        size_t held = 0, add = 0;
                        for (int k = 0; k < MAX_PIPES; k++) held += g.ws[k].bytes();
                        for (int k = 1; k <= max_waves && k < MAX_PIPES; k++) if (k != w && !slot[k].wave && g.ws[k].bytes() < g.ws[w].bytes() / 4) add += g.ws[w].bytes() - g.ws[k].bytes();
                        size_t free_b = 0, total_b = 0;
                        if (add && hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
                        // (at most half of the card for this process, and at most half of what is free now: other processes share it)
                        if (add && held + add <= (size_t)((double)g.hbm_total * 0.5) && add <= free_b / 2)
                            for (int k = 1; k <= max_waves && k < MAX_PIPES && !rc; k++)
                                if (k != w && !slot[k].wave && g.ws[k].bytes() < g.ws[w].bytes() / 4) { rc = init_ws(g.ws[k]); if (!rc) rc = g.ws[k].match(g.ws[w]); }
                    }
                    if (!rc) rc = sl.wave->issue_step();
                    // step-ahead: the materialize of the first step and the second step are queued at once (counts from the device,
                    // no size class merged away: the first steps are the big ones)
                    if (!rc && sl.wave->step_ahead) {
                        rc = sl.wave->issue_materialize(0u, 0x7fffffffu);
                        if (!rc) rc = sl.wave->issue_step();
                    }
                }
                progressed = true;
                if (rc) {
                    const std::string err = g_err;
                    { hipError_t e_ = hipDeviceSynchronize(); (void)e_; }
                    sl.wave.reset();
                    fail_or_split(sl, rc, err, queue);
                    continue;
                }
                n_running++; n_lane[ln]++;
                heavy_running = heavy_running || (job_heavy && sl.wave->heavy(admit_below()));
            }
        }
        // Nothing moved.  A step's read-back lands within tens to hundreds of microseconds, so the thread polls for a
        // short while; after that it stops holding a core (below).
        const auto now = std::chrono::steady_clock::now();
        if (progressed) { last_progress = now; continue; }
        if (linger) { std::this_thread::yield(); continue; }          // (at most linger_us: keep looking)
        const long spin_us = scfg.spin_us;
        if (std::chrono::duration_cast<std::chrono::microseconds>(now - last_progress).count() < spin_us) { std::this_thread::yield(); continue; }
        bool any_wave = false;
        for (int i = 0; i < MAX_PIPES; i++) any_wave = any_wave || (bool)slot[i].wave;
        if (any_wave) {
            // Several waves run and ANY of them may finish its step next: sleeping on one wave's event made the others wait for it
            // (measured: the long-tail wave's steps - 14 sequences, 0.3 ms of kernels - sat 0.5 ms on average, up to 1.6 ms, behind
            // a bulk wave's step, and while the long-tail wave holds one of the two wave slots the bulk waves do not overlap).  So
            // the thread naps in short slices and looks at all of them; a nap costs no core to speak of.
            const long nap_us = scfg.nap_us;
            std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
            continue;
        } else {
            // nothing running yet something queued (a job waiting for HBM that running waves hold cannot happen here: no wave runs)
            std::unique_lock<std::mutex> lk(g.qmu);
            g.qcv_sched.wait_for(lk, std::chrono::milliseconds(1), [] { return !g.submitted.empty() || g.stop; });
        }
    }
}

extern "C" void rafft_shutdown(void);
// In the child of a fork() the scheduler thread does not exist (only the forking thread survives), yet the inherited Ctx says it
// was started and the inherited atexit(rafft_shutdown) would join it: the child forgets the thread object (never joined, never
// destructed as joinable) and starts a scheduler of its own if it ever submits.  (A HIP context does not survive a fork either:
// a child that folds must initialise the GPU itself - this only keeps a child that does NOT fold from hanging in exit().)
static void atfork_child()
{
    new (&g.sched_thread) std::thread();      // placement-new over the stale handle: ~thread() of a joinable thread would terminate()
    g.sched_started = false; g.stop = false; g.n_inflight = 0;
    new (&g.qmu) std::mutex(); new (&g.mu) std::mutex();       // (may have been held by another thread of the parent at fork time)
}
static void start_scheduler()
{
    std::lock_guard<std::mutex> lk(g.qmu);    // (sched_started / sched_thread: the same mutex as rafft_shutdown)
    if (g.sched_started) return;
    g.sched_started = true;
    g.sched_cfg = read_config();              // the scheduler's own settings: read when it starts (rafft_config.h)
    g.sched_thread = std::thread(scheduler_main);
    static bool registered = false;
    if (!registered) {
        registered = true;
        // stopped and joined at process exit BEFORE the HIP runtime tears down (atexit handlers run in reverse order of
        // registration and the runtime registered its own when it was initialised, earlier than this)
        atexit(rafft_shutdown);
        pthread_atfork(nullptr, nullptr, atfork_child);
    }
}

// no batch may be in flight when the device tables change or a seam call borrows workspace 0
static void drain()
{
    std::unique_lock<std::mutex> lk(g.qmu);
    g.qcv_done.wait(lk, [] { return g.n_inflight == 0; });
}

void free_out(HostOut *o) { delete o; }      // (its pinned chunks go back to the pool with their last reference)

} // namespace

extern "C" {

const char *rafft_last_error(void) { return g_err.c_str(); }

/* Drains the batches in flight, stops the scheduler thread and joins it.  Registered with atexit(); may be called by
 * hand before unloading the library.  Entry points called afterwards start a fresh scheduler. */
void rafft_shutdown(void)
{
    std::thread t;
    {
        std::unique_lock<std::mutex> lk(g.qmu);
        if (!g.sched_started) return;
        g.qcv_done.wait(lk, [] { return g.n_inflight == 0; });
        g.stop = true;
        t = std::move(g.sched_thread);
    }
    g.qcv_sched.notify_all();
    if (t.joinable()) t.join();
    std::lock_guard<std::mutex> lk(g.qmu);
    g.stop = false; g.sched_started = false;
}

/* out[0..4] = device buffers allocated so far (calls), their bytes, the slowest such call in microseconds, pinned host chunks
 * allocated (calls), their bytes.  Process-wide, monotonic: a caller that takes the difference around a region of its own
 * sees whether the library had to allocate inside it (a hipMalloc of gigabytes now and then takes seconds). */
void rafft_alloc_counters(unsigned long long out[5])
{
    out[0] = g_dev_allocs; out[1] = g_dev_bytes; out[2] = g_dev_worst_us; out[3] = g_pin_allocs; out[4] = g_pin_bytes;
}

const char *rafft_version(void) { return "raffthip 0.2 (gfx950, HIP; built-in Turner-2004 37C tables or ViennaRNA parameter files)"; }

int rafft_init(int device)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return init_ctx(device);
}

struct rafft_job { std::shared_ptr<Batch> b; };

// (holds g.mu)
static int submit_locked(const rafft_params *p, int n_seq, const char *const *seqs, const int *lens, int device, rafft_job **job_, bool async_call = true)
{
    if (!p || !job_ || n_seq < 0 || (n_seq > 0 && !seqs)) return fail(RAFFT_ERR_PARAM, "null argument");
    *job_ = nullptr;
    if (!(p->temp > -273.15 && p->temp < 1000.0)) return fail(RAFFT_ERR_TEMP, "temp out of range");
    if (p->temp != 37.0 && !param_set().has_dH)
        return fail(RAFFT_ERR_TEMP, "temp != 37 needs the enthalpy tables of a ViennaRNA parameter file (rafft_load_params); "
                                    "the built-in tables are 37 C only");
    if (p->max_stack < 1 || p->max_stack > 65535) return fail(RAFFT_ERR_PARAM, "max_stack must be in [1, 65535]");
    if (p->nb_mode < 0 || p->max_branch < 0) return fail(RAFFT_ERR_PARAM, "nb_mode/max_branch must be >= 0");
    if (int rc = init_ctx(device)) return rc;
    if (g.T_dirty || g.T_temp != p->temp) {       // other tables: the batches in flight finish with theirs first
        drain();
        if (int rc = ensure_tables(p->temp)) return rc;
    }
    std::shared_ptr<Batch> bp(new Batch());
    Batch &b = *bp;
    b.cfg = read_config();                    // the environment switches as they are NOW travel with the batch (rafft_config.h)
    g_span_level = b.cfg.trace ? 2 : b.cfg.spans >= 0 ? b.cfg.spans : 1;
    b.p = *p; b.n_seq = n_seq; b.t0 = std::chrono::steady_clock::now();
    HostOut *ho = b.ho = new HostOut();
    ho->seq.resize(n_seq); ho->step_size.resize(n_seq); ho->step_off.resize(n_seq); ho->one_size.assign(n_seq, 0); ho->one_off.assign(n_seq, 0);
    ho->dcal_ptr.assign(n_seq, nullptr); ho->db_ptr.assign(n_seq, nullptr);
    // the sequences are copied: the caller's buffers may go away before rafft_fold_wait
    std::vector<int> L(n_seq);
    size_t tot = 0;
    for (int i = 0; i < n_seq; i++) { L[i] = lens ? lens[i] : (int)strlen(seqs[i]); tot += (size_t)std::max(L[i], 0); }
    b.seqbuf.resize(tot + 1);
    b.codebuf.resize(tot + 1);
    std::vector<SeqIn> good;
    size_t o = 0;
    for (int i = 0; i < n_seq; i++) {
        rafft_seq_result &sr = ho->seq[i];
        memset(&sr, 0, sizeof sr);
        sr.length = L[i];
        sr.status = RAFFT_ERR_HIP;          // "never folded": only emit_rows sets RAFFT_OK, with the rows in place
        if (L[i] <= 0) { sr.status = RAFFT_ERR_EMPTY; continue; }
        char *dst = b.seqbuf.data() + o;
        memcpy(dst, seqs[i], (size_t)L[i]);
        o += (size_t)L[i];
        unsigned bad = 0;
        uint8_t *cdst = b.codebuf.data() + (dst - b.seqbuf.data());
        for (int x = 0; x < L[i]; x++) { const unsigned k = kBaseCode[(unsigned char)dst[x]]; bad |= k; cdst[x] = (uint8_t)(k & 7); }
        if (bad & 8) { sr.status = RAFFT_ERR_BAD_CHAR; continue; }
        if (L[i] > RAFFT_MAX_LEN) { sr.status = RAFFT_ERR_TOO_LONG; continue; }
        good.push_back({dst, L[i], i, 0, cdst});
    }
    // ---- lanes.  Folds are independent, so how the batch is cut cannot change any result.  The number of
    // folding steps of a wave is set by its longest sequence, and the steps that only the long ones still need
    // are latency-bound and nearly empty (the benchmark set: 24 steps for two 2.9-knt sequences, 12 for the rest).
    // So a batch whose few longest sequences stand far out is cut in two jobs: the long tail starts first and runs
    // beside the bulk (and beside the bulk of the next batch).  The workspaces of the bulk lane have a stream
    // priority of their own, which gives them HW queues of their own - with all streams at one priority the waves
    // share the process's four queues and the cut is a loss (17.3 ms against 15.2 for the benchmark batch; with it: 13.3 ms).
    // RAFFT_SPLIT: unset / -1 automatic, 0 never, > 0 cut at that length.
    int split_len = 0;
    {
        const int want = b.cfg.split;
        if (good.size() >= 32 && want != 0 && (want > 0 || good.size() < 16384)) {     // (very large batches amortise the tail anyway)
            if (want > 0) split_len = want;
            else {   // the sequences at least twice as long as the 99th percentile of the batch (leaving room for two)
                std::vector<int> ls;
                for (auto &sq : good) ls.push_back(sq.len);
                std::sort(ls.begin(), ls.end());
                const size_t top = std::max<size_t>(2, ls.size() / 100);
                const int ref = ls[ls.size() - top - 1];
                if (ls.back() >= 2 * ref) split_len = 2 * ref;
            }
        }
    }
    {
        auto est_of = [&](const std::vector<SeqIn> &v) {
            // expected survivors per beam slot (~ folding steps in which a slot is renewed): grows with length
            size_t sl = 0;
            for (auto &sq : v) sl += sq.len;
            double e0 = 6.0 + (v.empty() ? 0.0 : (double)sl / (double)v.size()) / 100.0;
            if (b.cfg.est > 0) e0 = b.cfg.est;
            return e0;
        };
        std::vector<SeqIn> shorts, longs;
        for (auto &sq : good) (split_len > 0 && sq.len >= split_len ? longs : shorts).push_back(sq);
        if (!longs.empty() && !shorts.empty()) {
            b.lane[0].push_back(Job{longs, est_of(longs), 0});      // the long tail starts first
            b.lane[1].push_back(Job{shorts, est_of(shorts), 0});
        } else if (!good.empty())
            b.lane[good.size() >= 256 ? 1 : 0].push_back(Job{good, est_of(good), 0});
    }
    start_scheduler();
    {
        std::lock_guard<std::mutex> lk(g.qmu);
        g.submitted.push_back(bp);
        g.n_inflight++;
        g.t_last_submit = std::chrono::steady_clock::now();
        g.last_submit_async = async_call;
    }
    g.qcv_sched.notify_one();
    *job_ = new rafft_job{bp};
    return 0;
}

static int wait_job(rafft_job *job, rafft_result **out_)
{
    if (out_) *out_ = nullptr;
    if (!job) return fail(RAFFT_ERR_PARAM, "null job");
    std::shared_ptr<Batch> bp = job->b;
    delete job;
    {
        std::unique_lock<std::mutex> lk(g.qmu);
        g.qcv_done.wait(lk, [&] { return bp->done; });
    }
    {
        std::lock_guard<std::mutex> lk(g.mu);
        g.stats = bp->stats;
    }
    if (bp->rc) return fail(bp->rc, bp->err);
    if (!out_) { free_out(bp->ho); bp->ho = nullptr; return fail(RAFFT_ERR_PARAM, "null result pointer"); }
    *out_ = &bp->ho->res;
    bp->ho = nullptr;          // the caller owns it now (rafft_free_result)
    return 0;
}

int rafft_fold_submit(const rafft_params *p, int n_seq, const char *const *seqs, const int *lens, int device, rafft_job **job)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return submit_locked(p, n_seq, seqs, lens, device, job);
}

int rafft_fold_wait(rafft_job *job, rafft_result **out) { return wait_job(job, out); }

int rafft_fold_batch(const rafft_params *p, int n_seq, const char *const *seqs, const int *lens, int device, rafft_result **out_)
{
    if (!out_) return fail(RAFFT_ERR_PARAM, "null argument");
    *out_ = nullptr;
    rafft_job *job = nullptr;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (int rc = submit_locked(p, n_seq, seqs, lens, device, &job, false)) return rc;      // (this caller cannot be streaming: no linger)
    }
    return wait_job(job, out_);
}

void rafft_free_result(rafft_result *r)
{
    if (r && r->_owner) free_out((HostOut *)r->_owner);
}

int rafft_get_stats(rafft_stats *o)
{
    if (!o) return RAFFT_ERR_PARAM;
    std::lock_guard<std::mutex> lk(g.mu);
    *o = g.stats;
    return 0;
}

static int parse_db(const char *seq, const char *db, int L, std::vector<int16_t> &pt)
{
    pt.assign(L, -1);
    std::vector<int> stk;
    for (int i = 0; i < L; i++) {
        if (db[i] == '(') stk.push_back(i);
        else if (db[i] == ')') {
            if (stk.empty()) return RAFFT_ERR_STRUCT;
            int j = stk.back(); stk.pop_back();
            pt[i] = (int16_t)j; pt[j] = (int16_t)i;
        } else if (db[i] != '.') return RAFFT_ERR_STRUCT;
    }
    return stk.empty() ? 0 : RAFFT_ERR_STRUCT;
}

static thread_local bool g_ws_locked_by_me = false;     // rafft_expand_node holds ws_mu across its nested evaluation

static int eval_structures_impl(int n, const char *const *seqs, const char *const *dbs, int *dcal_out, int *status_out, double temp = 37.0, int *guessed_out = nullptr)
{
    if (int rc = init_ctx(-1)) return rc;
    drain();                                   // (g.mu is held: nothing new is submitted meanwhile)
    std::unique_lock<std::mutex> ws_lk(g.ws_mu, std::defer_lock);
    if (!g_ws_locked_by_me) ws_lk.lock();
    if (int rc = ensure_tables(temp)) return rc;
    if (int rc = init_ws(g.ws[0])) return rc;
    std::vector<long long> off(n);
    std::vector<int> len(n), status(n, 0);
    long long tot = 0;
    for (int i = 0; i < n; i++) {
        len[i] = (int)strlen(seqs[i]);
        off[i] = tot;
        if ((int)strlen(dbs[i]) != len[i] || len[i] > RAFFT_MAX_LEN) { status[i] = RAFFT_ERR_STRUCT; len[i] = 0; }      // (16-bit pair tables: positions 0..32767)
        tot += len[i];
    }
    std::vector<uint8_t> codes(tot + 16, 0);
    std::vector<int16_t> pts(tot + 16, -1);
    for (int i = 0; i < n; i++) {
        if (status[i]) continue;
        std::vector<int16_t> pt;
        if (parse_db(seqs[i], dbs[i], len[i], pt)) { status[i] = RAFFT_ERR_STRUCT; len[i] = 0; continue; }
        for (int x = 0; x < len[i]; x++) {
            char ch = seqs[i][x];
            int c = ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'U' ? 4 : ch == 'N' ? 0 : -1;
            if (c < 0) { status[i] = RAFFT_ERR_BAD_CHAR; break; }
            codes[off[i] + x] = (uint8_t)c;
            pts[off[i] + x] = pt[x];
        }
        if (status[i]) len[i] = 0;
    }
    void *dc = nullptr, *dp = nullptr, *doff = nullptr, *dlen = nullptr, *dout = nullptr, *dst = nullptr, *dg = nullptr;
    HIPCHK(hipMalloc(&dc, tot + 16)); HIPCHK(hipMalloc(&dp, (tot + 16) * 2)); HIPCHK(hipMalloc(&doff, n * 8 + 8));
    HIPCHK(hipMalloc(&dlen, n * 4 + 4)); HIPCHK(hipMalloc(&dout, n * 4 + 4)); HIPCHK(hipMalloc(&dst, n * 4 + 4)); HIPCHK(hipMalloc(&dg, n * 4 + 4));
    HIPCHK(hipMemcpy(dc, codes.data(), tot + 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp, pts.data(), (tot + 16) * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doff, off.data(), n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dlen, len.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_kernel, dim3(n), dim3(64), 0, g.ws[0].stream, g.T, n, (const uint8_t *)dc, (const int16_t *)dp,
                       (const long long *)doff, (const int *)dlen, (int *)dout, (int *)dst, (int *)dg);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.ws[0].stream));
    std::vector<int> st2(n);
    HIPCHK(hipMemcpy(dcal_out, dout, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st2.data(), dst, n * 4, hipMemcpyDeviceToHost));
    if (guessed_out) HIPCHK(hipMemcpy(guessed_out, dg, n * 4, hipMemcpyDeviceToHost));
    for (void *q : {dc, dp, doff, dlen, dout, dst, dg}) { hipError_t fe = hipFree(q); (void)fe; }
    if (guessed_out) for (int i = 0; i < n; i++) if (status[i]) guessed_out[i] = 0;
    int worst = 0;
    for (int i = 0; i < n; i++) {
        int s = status[i] ? status[i] : st2[i];
        if (status_out) status_out[i] = s;
        if (s && !worst) worst = s;
    }
    if (worst && !status_out) return fail(worst, "malformed structure, bad character or non-canonical pair");
    return 0;
}

int rafft_eval_structures(int n, const char *const *seqs, const char *const *dbs, int *dcal_out, int *status_out)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return eval_structures_impl(n, seqs, dbs, dcal_out, status_out);
}

int rafft_eval_structure(const char *seq, const char *db, int *dcal_out)
{
    return rafft_eval_structures(1, &seq, &db, dcal_out, nullptr);
}

int rafft_eval_structures_at(double temp, int n, const char *const *seqs, const char *const *dbs, int *dcal_out, int *status_out)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return eval_structures_impl(n, seqs, dbs, dcal_out, status_out, temp);
}

int rafft_eval_structures_info(int n, const char *const *seqs, const char *const *dbs, int *dcal_out, int *status_out, int *guessed_out)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return eval_structures_impl(n, seqs, dbs, dcal_out, status_out, 37.0, guessed_out);
}

int rafft_params_unpinned(int counts[3])
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!counts) return fail(RAFFT_ERR_PARAM, "null argument");
    counts[0] = counts[1] = counts[2] = 0;
    if (param_set().builtin_set) rafft_par::builtin_unpinned_counts(counts);
    return 0;
}

// ---- energy parameters (no GPU needed to load, inspect or save a parameter set; the upload happens with the next fold)

static int set_params_from_text(const std::string &text, const std::string &source)
{
    std::unique_ptr<rafft_par::ParamSet> P(new rafft_par::ParamSet());
    std::string err;
    if (!rafft_par::parse(text, *P, err)) return fail(RAFFT_ERR_PARAM, "parameter file " + source + ": " + err);
    P->source = source;
    {   // every table must survive the conversion to the device layout at 37 C
        std::unique_ptr<EnergyTables> h(new EnergyTables());
        if (!rafft_par::scaled_tables(*P, 37.0, h.get(), err)) return fail(RAFFT_ERR_PARAM, "parameter file " + source + ": " + err);
    }
    delete g.P;
    g.P = P.release();
    g.T_dirty = true;
    return 0;
}

int rafft_load_params(const char *path)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!path) return fail(RAFFT_ERR_PARAM, "null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RAFFT_ERR_PARAM, std::string("cannot open parameter file ") + path);
    std::string text;
    char buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
    fclose(f);
    return set_params_from_text(text, path);
}

int rafft_load_params_text(const char *text, const char *source_name)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!text) return fail(RAFFT_ERR_PARAM, "null text");
    return set_params_from_text(text, source_name ? source_name : "<memory>");
}

int rafft_reset_params(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    delete g.P;
    g.P = nullptr;
    g.T_dirty = true;
    return 0;
}

int rafft_save_params(const char *path)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!path) return fail(RAFFT_ERR_PARAM, "null path");
    const std::string txt = rafft_par::format(param_set());
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RAFFT_ERR_PARAM, std::string("cannot write ") + path);
    const bool ok = fwrite(txt.data(), 1, txt.size(), f) == txt.size();
    fclose(f);
    return ok ? 0 : fail(RAFFT_ERR_PARAM, std::string("short write to ") + path);
}

int rafft_params_info(char *source, int source_cap, int *has_enthalpies)
{
    std::lock_guard<std::mutex> lk(g.mu);
    const rafft_par::ParamSet &P = param_set();
    if (source && source_cap > 0) { strncpy(source, P.source.c_str(), (size_t)source_cap - 1); source[source_cap - 1] = 0; }
    if (has_enthalpies) *has_enthalpies = P.has_dH ? 1 : 0;
    return 0;
}

// The 37 C value of one table entry of the current parameter set, by ViennaRNA table name and flat row-major index in
// ViennaRNA's own array shape (pairs 0..7, bases 0..4): lets a caller check what a file gave without a GPU.
int rafft_param_value(const char *table, int enthalpy, long index, int *value_out)
{
    std::lock_guard<std::mutex> lk(g.mu);
    const rafft_par::ParamSet &P = param_set();
    if (!table || !value_out) return fail(RAFFT_ERR_PARAM, "null argument");
    const int w = enthalpy ? 1 : 0;
    struct Ent { const char *n; const int *p; long cnt; };
    const Ent ents[] = {
        {"stack", &P.stack[w][0][0], 64}, {"hairpin", P.hairpin[w], 31}, {"bulge", P.bulge[w], 31}, {"interior", P.interior[w], 31},
        {"mismatch_hairpin", &P.mmH[w][0][0][0], 200}, {"mismatch_interior", &P.mmI[w][0][0][0], 200},
        {"mismatch_interior_1n", &P.mm1n[w][0][0][0], 200}, {"mismatch_interior_23", &P.mm23[w][0][0][0], 200},
        {"mismatch_multi", &P.mmM[w][0][0][0], 200}, {"mismatch_exterior", &P.mmE[w][0][0][0], 200},
        {"dangle5", &P.d5[w][0][0], 40}, {"dangle3", &P.d3[w][0][0], 40},
        {"int11", &P.int11[w][0][0][0][0], 8 * 8 * 25}, {"int21", &P.int21[w][0][0][0][0][0], 8 * 8 * 125},
        {"int22", &P.int22[w][0][0][0][0][0][0], 8 * 8 * 625},
        {"ninio", &P.ninio[w], 1}, {"ml_base", &P.ml_base[w], 1}, {"ml_closing", &P.ml_closing[w], 1}, {"ml_intern", &P.ml_intern[w], 1},
        {"terminal_au", &P.term_au[w], 1}, {"max_ninio", &P.max_ninio, 1}};
    for (const Ent &e : ents)
        if (!strcmp(e.n, table)) {
            if (index < 0 || index >= e.cnt) return fail(RAFFT_ERR_PARAM, "index out of range");
            *value_out = e.p[index];
            return 0;
        }
    return fail(RAFFT_ERR_PARAM, std::string("unknown table ") + table);
}

int rafft_expand_node(const rafft_params *p, const char *seq, const char *db, const int *pos, int n,
                      int *n_ranked, int *lag, double *corval, int *nb, int *mi, int *mj,
                      double *score, int *ddcal, int *n_kept, int *kept)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (int rc = init_ctx(-1)) return rc;
    drain();
    std::lock_guard<std::mutex> ws_lk(g.ws_mu);
    struct Flag { Flag() { g_ws_locked_by_me = true; } ~Flag() { g_ws_locked_by_me = false; } } flag_;
    const int L = (int)strlen(seq);
    if (L == 0 || L > RAFFT_MAX_LEN || n < 1 || n > L) return fail(RAFFT_ERR_PARAM, "bad node");
    std::vector<int16_t> pt;
    if (parse_db(seq, db, L, pt)) return fail(RAFFT_ERR_STRUCT, "malformed dot-bracket");
    // enclosing loop of the region: nearest pair (i,j) with i < pos[0] < j
    int ci = -1, cj = L;
    for (int x = pos[0] - 1, depth = 0; x >= 0; x--) {
        if (pt[x] < 0) continue;
        if (pt[x] < x) { depth++; continue; }
        if (depth > 0) { depth--; continue; }
        if (pt[x] > pos[0]) { ci = x; cj = pt[x]; break; }
    }
    SeamIn sm;
    sm.ci = ci; sm.cj = cj;
    for (int x = ci + 1; x < cj;) {            // branch helices hanging in that loop
        if (pt[x] < 0) { x++; continue; }
        sm.br.push_back((uint32_t)x | ((uint32_t)pt[x] << 16));
        x = pt[x] + 1;
    }
    sm.pos.assign(pos, pos + n);
    int par_dcal = 0;
    if (int rc = eval_structures_impl(1, &seq, &db, &par_dcal, nullptr, p->temp)) return rc;   // (also scales the tables for p->temp)
    sm.pdcal = par_dcal;
    const int K = std::max(1, std::min(p->nb_mode, 2 * n - 1));
    if (int rc = ensure(g.ws[0].dbg, (size_t)K * (4 * 7 + 8 * 2) + 64)) return rc;
    char *b = (char *)g.ws[0].dbg.p;
    DebugOut &dbg = sm.dbg;
    dbg.n_ranked = (int *)b; b += 16;
    dbg.lag = (int *)b; b += 4 * K; dbg.nb = (int *)b; b += 4 * K; dbg.mi = (int *)b; b += 4 * K; dbg.mj = (int *)b; b += 4 * K;
    dbg.ddcal = (int *)b; b += 4 * K; dbg.kept = (int *)b; b += 4 * K;
    b = (char *)(((uintptr_t)b + 15) & ~(uintptr_t)15);
    dbg.corval = (double *)b; b += 8 * K; dbg.score = (double *)b;
    std::vector<SeqIn> one{{seq, L, 0, 0}};
    HostOut ho;
    ho.seq.resize(1); ho.step_size.resize(1); ho.step_off.resize(1); ho.one_size.assign(1, 0); ho.one_off.assign(1, 0); ho.dcal_ptr.assign(1, nullptr); ho.db_ptr.assign(1, nullptr);
    Batch bt;                                  // a private batch: the scheduler is idle (drained above) and g.mu is held
    bt.p = *p;
    bt.cfg = read_config();
    bt.p.max_stack = std::max(1, bt.p.max_stack);
    bt.n_seq = 1; bt.ho = &ho;
    const int src = run_seam(bt, one, sm);
    for (hipEvent_t e : bt.events) g.ev_free.push_back(e);
    if (src) return src;
    int hdr[4];
    HIPCHK(hipMemcpy(hdr, dbg.n_ranked, 16, hipMemcpyDeviceToHost));
    *n_ranked = hdr[0]; *n_kept = hdr[1];
    int r = hdr[0];
    HIPCHK(hipMemcpy(lag, dbg.lag, 4 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nb, dbg.nb, 4 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mi, dbg.mi, 4 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mj, dbg.mj, 4 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ddcal, dbg.ddcal, 4 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(kept, dbg.kept, 4 * hdr[1], hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(corval, dbg.corval, 8 * r, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(score, dbg.score, 8 * r, hipMemcpyDeviceToHost));
    return 0;
}

// ---- kinetics on the fast-folding graph

int rafft_kin_rate_matrix(int n_steps, const int *step_size, int L, const char *rows, const int *uid, int n_unique,
                          const double *energy, double kt, double *rate_device)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!step_size || !rows || !uid || !energy || !rate_device || n_steps < 1 || L < 1 || L > 32767 || n_unique < 1 || !(kt > 0))
        return fail(RAFFT_ERR_PARAM, "bad argument");
    if (int rc = init_ctx(-1)) return rc;
    // like the other seam calls: no fold in flight (hipMalloc / hipFree below synchronise the device, and the matrix is
    // written on the library's stream - the caller hands over a buffer its own stream is done with), workspace 0 held
    drain();
    std::lock_guard<std::mutex> ws_lk(g.ws_mu);
    if (int rc = init_ws(g.ws[0])) return rc;
    long long n = 0;
    std::vector<int> row0(n_steps);
    for (int i = 0; i < n_steps; i++) { row0[i] = (int)n; n += step_size[i]; if (step_size[i] < 0) return fail(RAFFT_ERR_PARAM, "negative step size"); }
    if (n < 1 || n > 0x7fffffff) return fail(RAFFT_ERR_PARAM, "bad number of structures");
    for (long long r = 0; r < n; r++) if (uid[r] < 0 || uid[r] >= n_unique) return fail(RAFFT_ERR_PARAM, "uid out of range");
    hipStream_t st = g.ws[0].stream;
    void *d_rows = nullptr, *d_pt = nullptr, *d_stack = nullptr, *d_uid = nullptr, *d_en = nullptr, *d_bad = nullptr;
    auto cleanup = [&]() { for (void *q : {d_rows, d_pt, d_stack, d_uid, d_en, d_bad}) if (q) { hipError_t fe = hipFree(q); (void)fe; } };
#define KCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(RAFFT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    KCHK(hipMalloc(&d_rows, (size_t)n * L)); KCHK(hipMalloc(&d_pt, (size_t)n * L * 2)); KCHK(hipMalloc(&d_stack, (size_t)n * L * 2));
    KCHK(hipMalloc(&d_uid, (size_t)n * 4)); KCHK(hipMalloc(&d_en, (size_t)n_unique * 8)); KCHK(hipMalloc(&d_bad, 4));
    KCHK(hipMemcpyAsync(d_rows, rows, (size_t)n * L, hipMemcpyHostToDevice, st));
    KCHK(hipMemcpyAsync(d_uid, uid, (size_t)n * 4, hipMemcpyHostToDevice, st));
    KCHK(hipMemcpyAsync(d_en, energy, (size_t)n_unique * 8, hipMemcpyHostToDevice, st));
    KCHK(hipMemsetAsync(d_bad, 0, 4, st));
    KCHK(hipMemsetAsync(rate_device, 0, (size_t)n_unique * n_unique * 8, st));
    hipLaunchKernelGGL(kin_pair_table_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (int)n, L, (const char *)d_rows,
                       (int16_t *)d_pt, (int16_t *)d_stack, (int *)d_bad);
    KCHK(hipGetLastError());
    for (int i = 0; i < n_steps; i++) {
        const int pi = i == 0 ? n_steps - 1 : i - 1;      // the reference compares step 0 with the LAST step (fast_paths[-1], rafft_kin.py:75)
        if (!step_size[i] || !step_size[pi]) continue;
        hipLaunchKernelGGL(kin_rates_kernel, dim3((unsigned)step_size[i]), dim3(KIN_NT), (size_t)L * 2, st, L, (const int16_t *)d_pt,
                           row0[i], step_size[pi], row0[pi], (const int *)d_uid, (const double *)d_en, kt, n_unique, rate_device);
        KCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(kin_diag_kernel, dim3((unsigned)n_unique), dim3(256), 0, st, n_unique, rate_device);
    KCHK(hipGetLastError());
    int bad = 0;
    KCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    KCHK(hipStreamSynchronize(st));
#undef KCHK
    cleanup();
    if (bad) return fail(RAFFT_ERR_STRUCT, "malformed dot-bracket row");
    return 0;
}

} // extern "C"
