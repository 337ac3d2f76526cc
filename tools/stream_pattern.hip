// Which streaming-read rate does the scoring kernel's ACCESS PATTERN allow?  Same loads as score_kernel (per wave and
// chunk: one dwordx4 per lane from each of P planes = P x 1 KiB), no compute.  Variants:
//   sep      P separate plane arrays (the engine's layout), one workgroup per 16-chunk segment
//   sep-xcd  same with the XCD-aware blockIdx remap of score_kernel
//   il       planes interleaved per chunk: a wave reads one contiguous P KiB, a workgroup 16 P KiB
//   il-xcd   interleaved + remap
// build: hipcc -O3 --offload-arch=gfx950 -o stream_pattern tools/stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
constexpr int P = 6, CHUNK_WORDS = 256, SEG = 16;

template <bool IL, bool XCD>
__global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ base, size_t plane_words, uint32_t n_chunks, unsigned *out) {
    uint32_t seg = blockIdx.x;
    if (XCD) { const uint32_t per = (gridDim.x + 7) / 8; seg = (blockIdx.x % 8) * per + blockIdx.x / 8; }
    const uint32_t c0 = seg * SEG;
    if (c0 >= n_chunks) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t ck = c0 + wave; ck < min(c0 + SEG, n_chunks); ck += 4) {
        uint4 v[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const size_t w = IL ? ((size_t)ck * P + p) * CHUNK_WORDS + lane * 4 : (size_t)p * plane_words + (size_t)ck * CHUNK_WORDS + lane * 4;
            v[p] = *reinterpret_cast<const uint4 *>(base + w);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) { acc.x ^= v[p].x; acc.y ^= v[p].y; acc.z ^= v[p].z; acc.w ^= v[p].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <bool IL, bool XCD>
void run(const char *name, const uint32_t *d, size_t plane_words, uint32_t n_chunks, unsigned *o) {
    const uint32_t segs = (n_chunks + SEG - 1) / SEG, grid = (segs + 7) / 8 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<IL, XCD><<<grid, 256>>>(d, plane_words, n_chunks, o);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<IL, XCD><<<grid, 256>>>(d, plane_words, n_chunks, o);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n_chunks * CHUNK_WORDS * 4 * P;
    printf("%-8s %u chunks x %d planes (%.2f GB), %u workgroups: %.3f ms, %.2f TB/s\n", name, n_chunks, P, bytes / 1e9, grid, ms / 10, 10 * bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const uint32_t n_chunks = argc > 1 ? atoi(argv[1]) : 127160;          // 1.04 Gbp padded / 8192
    const size_t plane_words = (size_t)n_chunks * CHUNK_WORDS;
    uint32_t *d; unsigned *o;
    (void)hipMalloc(&d, plane_words * 4 * P); (void)hipMalloc(&o, 4); (void)hipMemset(d, 1, plane_words * 4 * P);
    for (int rep = 0; rep < 2; ++rep) {
        run<false, false>("sep", d, plane_words, n_chunks, o);
        run<false, true>("sep-xcd", d, plane_words, n_chunks, o);
        run<true, false>("il", d, plane_words, n_chunks, o);
        run<true, true>("il-xcd", d, plane_words, n_chunks, o);
    }
    return 0;
}
