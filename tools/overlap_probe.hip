// Do kernels of two HIP streams overlap on this device?  (1) two small grids that each leave most of the device idle;
// (2) two device-filling grids of short workgroups — does the second one fill the slots the first leaves while it drains?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(long long cycles, unsigned *sink) {
    const long long t0 = wall_clock64();
    unsigned v = threadIdx.x;
    while (wall_clock64() - t0 < cycles) v = v * 1664525u + 1013904223u;
    if (v == 0x12345678u) *sink = v;
}
// uneven workgroups: blockIdx-dependent duration, like segments of uneven bins
__global__ void uneven(long long base, unsigned *sink) {
    const long long t0 = wall_clock64();
    const long long cycles = base * (1 + (blockIdx.x * 2654435761u >> 29));   // 1..8 x base
    unsigned v = threadIdx.x;
    while (wall_clock64() - t0 < cycles) v = v * 1664525u + 1013904223u;
    if (v == 0x12345678u) *sink = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    const long long ms = 100000;   // wall_clock64 ticks at 100 MHz: 1 ms
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, ms, sink);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, ms, sink);
        CK(hipDeviceSynchronize());
        double same = now() - t0;
        t0 = now();
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, ms, sink);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, b, ms, sink);
        CK(hipDeviceSynchronize());
        double two = now() - t0;
        printf("small grids (64 WG x 1 ms): same stream %.3f ms, two streams %.3f ms\n", same * 1e3, two * 1e3);
    }
    for (int wgs : {4096, 16384}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int n = 20;
            CK(hipDeviceSynchronize());
            double t0 = now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(uneven, dim3(wgs), dim3(256), 0, a, 500, sink);
            CK(hipDeviceSynchronize());
            double same = (now() - t0) / n;
            t0 = now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(uneven, dim3(wgs), dim3(256), 0, (i & 1) ? b : a, 500, sink);
            CK(hipDeviceSynchronize());
            double two = (now() - t0) / n;
            printf("device-filling grids (%d uneven WGs of 5-40 us): same stream %.4f ms/launch, alternating streams %.4f ms/launch\n", wgs, same * 1e3, two * 1e3);
        }
    }
    return 0;
}
