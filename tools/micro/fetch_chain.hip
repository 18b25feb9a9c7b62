// micro-benchmark: a persistent kernel whose wavefronts claim items from a work list and read a 64-byte record per item
// (the fetch + header part of expand_kernel), by occupancy and by how the record is addressed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { int seq, pdcal, n, ci, cj, nbr, ncand, L; unsigned long long pos, br, cand, soff; };
template <int MODE>
__global__ void chain(const int *work, const Rec *recs, unsigned n_items, unsigned *cursor, unsigned long long *out, int fetch)
{
    extern __shared__ int lds[];
    unsigned long long acc = 0;
    const int lane = threadIdx.x & 63;
    unsigned base = 0, left = 0;
    for (;;) {
        if (left == 0) {
            unsigned b = 0;
            if (lane == 0) b = atomicAdd(cursor, (unsigned)fetch);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
            left = fetch;
        }
        const unsigned item = base; base++; left--;
        if (item >= n_items) break;
        const int nid = __builtin_amdgcn_readfirstlane(work[item]);
        if (MODE == 0) {            // as expand_kernel: uniform vector loads, L first
            const int L = recs[nid].L;
            acc += L + recs[nid].n + recs[nid].ci + recs[nid].cj + recs[nid].nbr + recs[nid].pdcal + recs[nid].pos + recs[nid].br + recs[nid].soff;
        } else if (MODE == 1) {     // one 64-byte read spread over 16 lanes, fields by shuffle
            const int *p = (const int *)&recs[nid];
            int v = lane < 16 ? p[lane] : 0;
            for (int k = 0; k < 16; k++) acc += __shfl(v, k, 64);
        } else {                    // per-lane different records (gather), for comparison
            const int nid2 = work[(item + lane * 64) % n_items];
            acc += recs[nid2].L;
        }
    }
    if (acc == 12345) out[0] = acc;
}
int main()
{
    const unsigned N = 150000;
    const size_t nrec = (size_t)24 << 20;        // 1.5 GB of records
    std::vector<int> work(N);
    std::mt19937_64 rng(1);
    // records of one step sit in 64 windows of the arena (one per allocation shard)
    for (unsigned i = 0; i < N; i++) work[i] = (int)((rng() % 64) * (nrec / 64) + 100000 + rng() % 2400);
    int *dw; Rec *dr; unsigned *cur; unsigned long long *out;
    CK(hipMalloc(&dw, N * 4)); CK(hipMalloc(&dr, nrec * sizeof(Rec))); CK(hipMalloc(&cur, 4)); CK(hipMalloc(&out, 8));
    CK(hipMemcpy(dw, work.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(dr, 0, nrec * sizeof(Rec)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](int mode, int grid, int block, int lds, int fetch) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemset(cur, 0, 4));
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(grid), dim3(block), lds, 0, dw, dr, N, cur, out, fetch);
            else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(grid), dim3(block), lds, 0, dw, dr, N, cur, out, fetch);
            else hipLaunchKernelGGL(chain<2>, dim3(grid), dim3(block), lds, 0, dw, dr, N, cur, out, fetch);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = std::min(best, ms);
        }
        printf("mode %d  grid %4d x %4d  lds %6d  fetch %d: %8.1f us  = %6.2f ns per item, %6.2f us per item and wavefront\n", mode, grid, block, lds, fetch, best * 1e3, best * 1e6 / N,
               best * 1e3 / N * (grid * block / 64));
    };
    CK(hipFuncSetAttribute((const void *)chain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int mode : {0, 1, 2})
        for (int fetch : {1, 4}) {
            run(mode, 256, 768, 159 * 1024, fetch);      // expand_kernel<64,true,12>: 12 wavefronts per CU
            run(mode, 1024, 256, 0, fetch);              // 16 wavefronts per CU
            run(mode, 2048, 256, 0, fetch);              // 32 wavefronts per CU
        }
    return 0;
}
