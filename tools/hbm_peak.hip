// Achievable HBM read bandwidth on this device with a plain streaming kernel (dwordx4 loads, grid-stride), to put the
// scan kernel's real traffic rate next to something measured rather than the 8 TB/s spec number.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void stream_read(const uint4 *__restrict__ p, size_t n, unsigned int *out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
int main() {
    const size_t bytes = 8ull << 30, n = bytes / 16;
    uint4 *d; unsigned int *o;
    (void)hipMalloc(&d, bytes); (void)hipMalloc(&o, 4); (void)hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        stream_read<<<blocks, 256>>>(d, n, o);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) stream_read<<<blocks, 256>>>(d, n, o);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("stream read 8 GiB, %5d workgroups: %.2f TB/s\n", blocks, 5.0 * bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
