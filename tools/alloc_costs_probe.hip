// What the fixed costs at the start of the two device parsers are made of: pinned allocations, stream creation, a stream's first use,
// device allocations and their first touch — each timed on a fresh process (tools/gpu_alloc_costs.sh).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, stmt) do { const double t0 = now(); hipError_t e = (stmt); printf("%-58s %8.3f ms%s\n", label, (now() - t0) * 1e3, e == hipSuccess ? "" : "  FAILED"); } while (0)
__global__ void touch(unsigned char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
static hipError_t run_touch(hipStream_t st, void *p, size_t n, unsigned blocks) {
    hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, st, (unsigned char *)p, n);
    return hipStreamSynchronize(st);
}
int main() {
    double t0 = now();
    hipSetDevice(0); hipFree(nullptr);
    printf("%-58s %8.3f ms\n", "runtime up (hipSetDevice + hipFree(0))", (now() - t0) * 1e3);
    hipStream_t s0, s1, s2, p[3];
    T("hipStreamCreateWithFlags #1", hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    T("hipStreamCreateWithFlags #2", hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    T("hipStreamCreateWithFlags #3", hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    void *h[4], *d[4];
    T("hipHostMalloc 32 MB #1", hipHostMalloc(&h[0], 32u << 20, hipHostMallocDefault));
    T("hipHostMalloc 32 MB #2", hipHostMalloc(&h[1], 32u << 20, hipHostMallocDefault));
    T("hipHostMalloc 32 MB #3", hipHostMalloc(&h[2], 32u << 20, hipHostMallocDefault));
    T("hipHostMalloc 8 MB", hipHostMalloc(&h[3], 8u << 20, hipHostMallocDefault));
    T("hipMalloc 3.2 GB #1", hipMalloc(&d[0], 3200ull << 20));
    T("hipMalloc 3.2 GB #2", hipMalloc(&d[1], 3200ull << 20));
    T("hipMalloc 2 GB", hipMalloc(&d[2], 2048ull << 20));
    T("hipMalloc 0.1 GB", hipMalloc(&d[3], 100ull << 20));
    memset(h[0], 1, 32u << 20);
    T("first hipMemcpyAsync on stream #1 (32 MB) + sync", (hipMemcpyAsync(d[3], h[0], 32u << 20, hipMemcpyHostToDevice, s0), hipStreamSynchronize(s0)));
    T("second hipMemcpyAsync on stream #1 (32 MB) + sync", (hipMemcpyAsync(d[3], h[0], 32u << 20, hipMemcpyHostToDevice, s0), hipStreamSynchronize(s0)));
    T("first hipMemcpyAsync on stream #2 (32 MB) + sync", (hipMemcpyAsync(d[3], h[1], 32u << 20, hipMemcpyHostToDevice, s1), hipStreamSynchronize(s1)));
    T("first kernel on stream #3 (touch 3.2 GB, 4 KB stride) + sync", run_touch(s2, d[0], 3200ull << 20, 3200));
    T("second kernel on stream #3 (same buffer) + sync", run_touch(s2, d[0], 3200ull << 20, 3200));
    T("kernel on stream #3 (the other 3.2 GB buffer) + sync", run_touch(s2, d[1], 3200ull << 20, 3200));
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    T("hipStreamCreateWithPriority (lowest) #1", hipStreamCreateWithPriority(&p[0], hipStreamNonBlocking, least));
    T("hipStreamCreateWithPriority (lowest) #2", hipStreamCreateWithPriority(&p[1], hipStreamNonBlocking, least));
    T("hipStreamCreateWithPriority (lowest) #3", hipStreamCreateWithPriority(&p[2], hipStreamNonBlocking, least));
    T("first kernel on priority stream #1 + sync", run_touch(p[0], d[3], 4096, 1));
    T("first kernel on priority stream #2 + sync", run_touch(p[1], d[3], 4096, 1));
    T("first kernel on priority stream #3 + sync", run_touch(p[2], d[3], 4096, 1));
    hipEvent_t ev;
    T("hipEventCreateWithFlags", hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipStream_t more[4];
    for (int i = 0; i < 4; ++i) { char l[64]; snprintf(l, sizeof l, "stream #%d create + first kernel + sync", 7 + i);
        T(l, (hipStreamCreateWithFlags(&more[i], hipStreamNonBlocking), run_touch(more[i], d[3], 4096, 1))); }
    T("hipHostMalloc 256 MB", hipHostMalloc(&h[3], 256u << 20, hipHostMallocDefault));
    T("hipHostFree 256 MB", hipHostFree(h[3]));
    T("hipFree 3.2 GB", hipFree(d[0]));
    return 0;
}
