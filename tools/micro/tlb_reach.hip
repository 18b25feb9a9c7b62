// micro-benchmark: latency of dependent random reads by footprint (TLB reach) on gfx950, idle chip and loaded chip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// each lane walks its own chain: idx = (idx * A + C) mod n (LCG over line indices), reads 8 bytes of the line
__global__ void chase(const unsigned long long *buf, unsigned long long n_lines, int steps, unsigned long long *out, int active_lanes)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    if ((threadIdx.x & 63) >= active_lanes) return;
    unsigned long long idx = (gid * 0x9E3779B97F4A7C15ULL) % n_lines, acc = 0;
    for (int s = 0; s < steps; s++) {
        const unsigned long long v = buf[idx * 8];              // 64-byte lines
        acc += v;
        idx = (idx * 6364136223846793005ULL + 1442695040888963407ULL + v) % n_lines;   // depends on the loaded value (0)
    }
    if (acc == 12345) out[0] = acc;
}
int main()
{
    const size_t maxbytes = (size_t)48 << 30;
    unsigned long long *buf, *out;
    CK(hipMalloc(&buf, maxbytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, maxbytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mb : {16, 256, 1024, 4096, 16384, 49152}) {
        const unsigned long long n_lines = (mb << 20) / 64;
        for (int mode = 0; mode < 3; mode++) {
            const int grid = mode == 0 ? 1 : 1024, lanes = mode == 2 ? 64 : 1, steps = mode == 0 ? 2000 : 400;
            hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, buf, n_lines, 50, out, lanes);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, buf, n_lines, steps, out, lanes);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("footprint %6zu MB  %s: %8.1f ns per dependent read\n", mb, mode == 0 ? "1 wavefront, 1 lane   " : mode == 1 ? "4096 wavefronts, 1 lane" : "4096 wavefronts, 64 lanes", ms * 1e6 / steps);
        }
    }
    return 0;
}
