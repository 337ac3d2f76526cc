// Micro-benchmark 2: issue cost of candidate inner-loop instructions on gfx950 (cycles per wave64 op per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t s = (seed | 1) & 31;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t &x = a[i], y = a[(i + 1) & 15], z = a[(i + 5) & 15];
            if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(y));
            if (OP == 1) asm volatile("v_and_b32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if (OP == 2) asm volatile("v_and_b32_dpp %0, %1, %0 row_ror:5 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if (OP == 3) asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(x) : "v"(y), "v"(z), "s"(s));
            if (OP == 4) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(x) : "v"(y), "v"(z));
            if (OP == 5) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(y));
            if (OP == 6) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x) : "v"(y));
            if (OP == 7) asm volatile("v_or_b32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(z));
            if (OP == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(x) : "v"(y), "v"(z));
            if (OP == 10) asm volatile("v_and_b32 %0, %1, %0\n\tv_and_b32 %0, %2, %0" : "+v"(x) : "v"(y), "v"(z));
            if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int per) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 12345, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 12345, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 16 * per;
    printf("%-28s %8.3f ms  %.2f cycles per wave64 op per SIMD (at 2.4 GHz)\n", name, ms, (1024 * 2.4e9) / (winstr / (ms * 1e-3)));
    (void)hipFree(d);
}
int main() {
    run<0>("v_and_b32", 1); run<1>("v_and_b32_dpp row_shr", 1); run<2>("v_and_b32_dpp row_ror", 1); run<3>("v_alignbit (sgpr shift)", 1);
    run<4>("v_alignbit (imm shift)", 1); run<5>("v_lshrrev_b32 (sgpr)", 1); run<6>("v_bcnt_u32_b32", 1); run<7>("v_or_b32_dpp row_shl", 1);
    run<8>("v_cndmask_b32 vcc", 1); run<9>("v_bitop3_b32", 1); run<10>("2 x v_and_b32", 2); run<11>("v_mov_b32_dpp row_shr", 1);
    return 0;
}
