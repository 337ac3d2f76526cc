// Micro-benchmark: integer VALU issue rate on gfx950 for the ops the scoring kernel is made of.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_peak tools/valu_peak.hip && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t s = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = a[i] & (a[(i + 1) & 15] | s);                                  // v_and_or / bitop3
            if (OP == 1) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 15], s);          // v_alignbit_b32
            if (OP == 2) a[i] = a[i] & __builtin_amdgcn_alignbit(a[(i + 3) & 15], a[(i + 1) & 15], s);  // alignbit + and
            if (OP == 3) a[i] = __builtin_fmaf(__uint_as_float(a[i]), 1.0001f, __uint_as_float(a[(i + 1) & 15])) > 0 ? a[i] + 1 : a[i];
            if (OP == 4) a[i] = __popc(a[(i + 1) & 15]) + a[i];                               // v_bcnt_u32_b32
        }
        s = s * 3 + 1;
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int vops_per_iter) {
    uint32_t *d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int iters = 4096, blocks = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves / SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 12345, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 12345, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * vops_per_iter;   // wave-instructions
    printf("%-18s %8.3f ms  %.3e wave-instr/s  = %.2f wave-instr/cycle/SIMD at 2.4 GHz (%.2f cycles per wave64 op)\n", name, ms,
           winstr / (ms * 1e-3), winstr / (ms * 1e-3) / (1024 * 2.4e9), (1024 * 2.4e9) / (winstr / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    run<0>("and_or", 16); run<1>("alignbit", 16); run<2>("alignbit+and", 32); run<3>("fma+cmp+add", 48); run<4>("bcnt", 16);
    return 0;
}
