// rafft_kin.hip - the rate matrix of the fast-folding-graph kinetics on the GPU.
//
// Replaces the O(steps * ms^2 * L) Python set logic of the reference's post-processor:
//   get_connected_prev   rafft/rafft_kin.py:48-56   a structure of step i-1 is connected to a structure of step i
//                                                   when all of its pairs are pairs of the latter
//   get_transition_mat   rafft/rafft_kin.py:68-91   Metropolis rates min(1, exp(-+dE/KT)) between connected
//                                                   structures, diagonal = -(row sum)
// Structures arrive as dot-bracket rows (what the fold left in the fast-folding graph / side-car).  Pair-set
// inclusion on nested structures is a position-wise test on pair tables: every paired position of the earlier
// structure has the same partner in the later one.  One workgroup per structure of step i keeps its pair table in
// LDS; its wavefronts walk the structures of step i-1, 64 positions per wavefront instruction, and stop at the first
// violation (a ballot).  HBM-bound byte/integer work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

// pair tables from dot-bracket rows: one thread per row, explicit stack in a global scratch row
__global__ void kin_pair_table_kernel(int n, int L, const char *rows, int16_t *pt, int16_t *stack, int *bad)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const char *row = rows + (size_t)r * L;
    int16_t *p = pt + (size_t)r * L, *st = stack + (size_t)r * L;
    int sp = 0;
    for (int x = 0; x < L; x++) {
        const char c = row[x];
        p[x] = -1;
        if (c == '(') st[sp++] = (int16_t)x;
        else if (c == ')') {
            if (sp == 0) { *bad = 1; return; }
            const int j = st[--sp];
            p[x] = (int16_t)j; p[j] = (int16_t)x;
        } else if (c != '.') { *bad = 1; return; }
    }
    if (sp) *bad = 1;
}

// rates between connected structures.  cur_row0/prev_row0: first rows of the two steps; uid: row -> unique structure;
// energy: per unique structure (the energy of its first appearance, rafft_kin.py:115); rate: S x S row-major, zeroed.
// Several (step, pair) occurrences of the same two structures write the same values (benign).
#define KIN_NT 256
__global__ __launch_bounds__(KIN_NT) void kin_rates_kernel(int L, const int16_t *pt, int cur_row0, int n_prev, int prev_row0,
                                                           const int *uid, const double *energy, double kt, int S, double *rate)
{
    extern __shared__ int16_t cur[];
    const int c = cur_row0 + blockIdx.x;
    const int16_t *pc = pt + (size_t)c * L;
    for (int x = threadIdx.x; x < L; x += KIN_NT) cur[x] = pc[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int uc = uid[c];
    const double ec = energy[uc];
    for (int p = wv; p < n_prev; p += KIN_NT / 64) {
        const int16_t *pp = pt + (size_t)(prev_row0 + p) * L;
        bool ok = true;
        for (int x0 = 0; x0 < L && ok; x0 += 64) {
            const int x = x0 + lane;
            const int q = x < L ? (int)pp[x] : -1;
            const bool viol = q >= 0 && q != (int)cur[x < L ? x : 0];
            if (__ballot(viol)) ok = false;
        }
        if (ok && lane == 0) {
            const int up = uid[prev_row0 + p];
            if (up != uc) {
                const double d = ec - energy[up];       // delta_nrj = cur_nrj - prev_nrj
                rate[(size_t)up * S + uc] = fmin(1.0, exp(-d / kt));
                rate[(size_t)uc * S + up] = fmin(1.0, exp(d / kt));
            }
        }
    }
}

// transition_mat[si, si] = -transition_mat[si, :].sum()   (rafft_kin.py:87-88); one workgroup per row
__global__ __launch_bounds__(256) void kin_diag_kernel(int S, double *rate)
{
    __shared__ double part[256];
    const int r = blockIdx.x;
    double s = 0.0;
    for (int c = threadIdx.x; c < S; c += 256) s += rate[(size_t)r * S + c];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) rate[(size_t)r * S + r] = -part[0];
}
