// micro-benchmark: what hipMalloc costs while kernels run on other streams, against an idle device, and after a
// hipDeviceSynchronize().  (The scheduler allocates workspaces between the steps of running waves; one such call in a few
// took 2.2 s.)   usage: malloc_busy.bin [rounds]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long cycles, int *out)
{
    const unsigned long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    const size_t sizes[] = {(size_t)360 << 20, (size_t)2 << 30, (size_t)7 << 30};
    hipStream_t st[3];
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *flag; hipMalloc(&flag, 64);
    for (int mode = 0; mode < 3; mode++) {          // 0: idle device, 1: kernels in flight, 2: kernels in flight, synchronize first
        std::atomic<bool> stop{false};
        std::thread feeder;
        if (mode) feeder = std::thread([&] {
            hipSetDevice(0);
            int k = 0;
            while (!stop) {       // ~0.3 ms kernels on three streams, a few queued ahead, host polling an event like the scheduler does
                for (int q = 0; q < 4; q++) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, st[k % 3], 30000ULL, flag);
                hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); hipEventRecord(e, st[k % 3]);
                while (hipEventQuery(e) == hipErrorNotReady && !stop) std::this_thread::yield();
                hipEventDestroy(e); k++;
            }
        });
        double worst = 0, sum = 0, worst_sync = 0; int n = 0;
        std::vector<void *> held;
        for (int r = 0; r < rounds; r++)
            for (size_t sz : sizes) {
                double ts = now();
                if (mode == 2) hipDeviceSynchronize();
                const double t0 = now();
                void *p = nullptr;
                if (hipMalloc(&p, sz) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
                const double dt = now() - t0;
                worst = dt > worst ? dt : worst; sum += dt; n++;
                worst_sync = t0 - ts > worst_sync ? t0 - ts : worst_sync;
                if (dt > 50) printf("  mode %d: hipMalloc(%zu MB) took %.1f ms (round %d)\n", mode, sz >> 20, dt, r);
                held.push_back(p);
                if (held.size() >= 9) {          // free in batches, idle (hipFree waits for the device anyway)
                    stop = mode ? stop.load() : false;
                    for (void *q : held) hipFree(q);
                    held.clear();
                }
            }
        stop = true;
        if (feeder.joinable()) feeder.join();
        hipDeviceSynchronize();
        for (void *q : held) hipFree(q);
        printf("%-44s %d calls: mean %.3f ms, worst %.3f ms%s\n", mode == 0 ? "idle device" : mode == 1 ? "kernels in flight" : "kernels in flight, hipDeviceSynchronize first",
               n, sum / n, worst, mode == 2 ? "" : "");
        if (mode == 2) printf("  (worst wait in hipDeviceSynchronize %.3f ms)\n", worst_sync);
    }
    return 0;
}
