// micro-benchmark: what an (almost) empty launch costs on gfx950 by grid size, LDS size, register footprint and prologue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { unsigned long long a[90]; };   // ~720-byte kernarg like Dev
__global__ void k_empty(Big b, int *out) { if (b.a[3] == 12345 && threadIdx.x == 0) out[0] = 1; }
__global__ void k_lds(Big b, int *out) { extern __shared__ int l[]; if (b.a[3] == 12345) { l[threadIdx.x] = 1; out[0] = l[0]; } }
__global__ void k_copy(Big b, const int *tab, int *out) {
    extern __shared__ int l[];
    for (int i = threadIdx.x; i < 860; i += blockDim.x) l[i] = tab[i];
    __syncthreads();
    if (b.a[3] == 12345) out[0] = l[b.a[4] & 63];
}
__global__ void k_copy4(Big b, const int4 *tab, int *out) {
    extern __shared__ int l[];
    if (threadIdx.x < 215) ((int4 *)l)[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (b.a[3] == 12345) out[0] = l[b.a[4] & 63];
}
// register-heavy variant: launch bounds force 128 VGPRs
__global__ __launch_bounds__(256, 4) void k_regs(Big b, const int *tab, int *out) {
    extern __shared__ int l[];
    for (int i = threadIdx.x; i < 860; i += blockDim.x) l[i] = tab[i];
    __syncthreads();
    if (b.a[3] == 12345) {
        float acc[100];
        for (int i = 0; i < 100; i++) acc[i] = tab[i] * 1.5f + threadIdx.x;
        for (int r = 0; r < 100; r++) for (int i = 0; i < 100; i++) acc[i] = acc[i] * acc[(i + 1) % 100] + 1.0f;
        float s = 0; for (int i = 0; i < 100; i++) s += acc[i];
        out[threadIdx.x] = (int)s;
    }
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    Big b{}; int *out; int *tab; hipMalloc(&out, 4096); hipMalloc(&tab, 8192); hipMemset(tab, 0, 8192);
    for (int grid : {64, 256, 1024, 4096}) {
        for (int lds : {0, 24576, 65536}) {
            float t0 = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, b, out); }, 200);
            float t1 = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), lds, 0, b, out); }, 200);
            float t2 = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), lds + 4096, 0, b, tab, out); }, 200);
            float t3 = timeit([&] { hipLaunchKernelGGL(k_copy4, dim3(grid), dim3(256), lds + 4096, 0, b, (const int4 *)tab, out); }, 200);
            float t4 = timeit([&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), lds + 4096, 0, b, tab, out); }, 200);
            printf("grid %5d x256  lds %6d:  empty %6.1f us  lds %6.1f  copy-loop %6.1f  copy-uint4 %6.1f  128-vgpr %6.1f\n", grid, lds, t0, t1, t2, t3, t4);
        }
    }
    return 0;
}
