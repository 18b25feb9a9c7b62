// micro-benchmark: throughput of returning device-scope atomics by how many counters they are spread over and how far
// apart the counters lie (which unit of the memory system serialises them?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(unsigned *ctr, unsigned n_ctr, size_t stride_words, unsigned per_wave, unsigned long long *out)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned *c = ctr + (size_t)(wave % n_ctr) * stride_words;
    unsigned long long acc = 0;
    for (unsigned i = 0; i < per_wave; i++) {
        unsigned b = 0;
        if (lane == 0) b = atomicAdd(c, 1u);
        acc += (unsigned)__builtin_amdgcn_readfirstlane((int)b);       // the next claim depends on this one
    }
    if (acc == 12345) out[0] = acc;
}
int main()
{
    unsigned *ctr; unsigned long long *out;
    const size_t bytes = (size_t)1 << 30;
    CK(hipMalloc(&ctr, bytes)); CK(hipMalloc(&out, 8)); CK(hipMemset(ctr, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 1024, block = 256;            // 4096 wavefronts
    const unsigned per_wave = 16;                  // 65536 atomics in all
    for (unsigned n_ctr : {1u, 8u, 64u, 512u, 4096u})
        for (size_t stride : {(size_t)64, (size_t)256, (size_t)4096, (size_t)65536, (size_t)262144 + 256}) {
            if (n_ctr * stride > bytes) continue;
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, ctr, n_ctr, stride / 4, per_wave, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                best = std::min(best, ms);
            }
            printf("%5u counters, %7zu bytes apart: %8.1f us for %u atomics = %6.2f ns per atomic\n", n_ctr, stride, best * 1e3, 4096 * per_wave, best * 1e6 / (4096.0 * per_wave));
        }
    return 0;
}
