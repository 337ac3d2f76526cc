// Stand-alone harness (round 6) for the device inflate kernels of nanomotif_amd/csrc/nmbedinflate.h — the SHIPPED device code, included
// as it is: bed_inflate_kernel (one lane per block, rounds 4 - 5) against the two-phase form bed_tokens_kernel + bed_resolve_kernel.
// Host: makes modkit-like bedMethyl text, deflates DISTINCT blocks of it with zlib (level 6, raw streams, 65 280 bytes of text each: what
// bgzip writes), lays copies of them out as one slab of the size the library uses (3 GiB of text by default), runs every kernel on the
// slab, checks the text byte for byte and prints each kernel's time.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o inflate2_proto tools/inflate2_proto.hip -lz -lpthread
// Run:   ./inflate2_proto [blocks in the slab = 49152] [distinct blocks = 1024] [repetitions = 3]
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace {
#include "../nanomotif_amd/csrc/nmbedinflate.h"
}

int main(int argc, char **argv) {
    const size_t n_blocks = argc > 1 ? (size_t)atol(argv[1]) : 49152;
    const size_t n_distinct = std::min(n_blocks, argc > 2 ? (size_t)atol(argv[2]) : 1024);
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const double fraction = argc > 4 ? atof(argv[4]) : 0.625;         // token region per block as a fraction of its text (the library's default)
    // what the text is: 0 = modkit-like rows; 1 = runs and periodic patterns (period 1 .. 300, run lengths up to 70 000: matches whose
    // distance is below their length, matches of 258 bytes one after the other); 2 = random bytes (nothing to compress: literal runs far
    // beyond 255, token regions overflow and the single kernel takes over); 3 = a mixture of all three, block by block
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    const size_t bs = 0xFF00;
    // modkit-like rows (the columns and value ranges of nanomotif_amd/synth.py's writer)
    std::string text;
    {
        std::mt19937_64 rng(7);
        text.reserve(n_distinct * bs + 256);
        unsigned long long pos = 0;
        int contig = 0;
        char line[256];
        auto want_mode = [&]() { return mode == 3 ? (int)((text.size() / bs) % 3) : mode; };
        while (text.size() < n_distinct * bs) {
            if (want_mode() == 1) {                                    // a run of one pattern
                const size_t period = 1 + rng() % (rng() % 4 == 0 ? 300 : 12), len = 1 + rng() % (rng() % 8 == 0 ? 70000 : 600);
                std::string pat(period, 'x');
                for (auto &ch : pat) ch = (char)('A' + rng() % 20);
                for (size_t k = 0; k < len; ++k) text.push_back(pat[k % period]);
                continue;
            }
            if (want_mode() == 2) {                                    // noise
                for (int k = 0; k < 4096; ++k) text.push_back((char)(rng() & 0xFF));
                continue;
            }
            pos += 1 + rng() % 3;
            if (pos > 2000000) { pos = rng() % 5; ++contig; }
            const int cov = 10 + (int)(rng() % 40), pct = (int)(rng() % 10000), nmod = cov * pct / 10000;
            const int k = snprintf(line, sizeof line, "contig_%04d\t%llu\t%llu\t%s\t%d\t%c\t%llu\t%llu\t255,0,0\t%d\t%d.%02d\t%d\t%d\t0\t0\t0\t0\t0\n", contig, pos, pos + 1,
                                   (rng() & 1) ? "a" : "m", cov, (rng() & 1) ? '+' : '-', pos, pos + 1, cov, pct / 100, pct % 100, nmod, cov - nmod);
            text.append(line, (size_t)k);
        }
        text.resize(n_distinct * bs);
    }
    struct Comp { std::string z; };
    std::vector<Comp> comp(n_distinct);
    {
        const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t] {
                std::vector<unsigned char> tmp(2 * bs + 1024);             // (noise in the FIXED code is 9/8 of its size: compressBound() is not enough)
                for (size_t i = t; i < n_distinct; i += nt) {
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    deflateInit2(&zs, i % 61 == 60 ? 1 : 6, Z_DEFLATED, -15, 8, i % 97 == 96 ? Z_FIXED : Z_DEFAULT_STRATEGY);      // a few fixed-code and level-1 blocks among them
                    zs.next_in = (Bytef *)text.data() + i * bs;
                    zs.avail_in = (uInt)bs;
                    zs.next_out = tmp.data();
                    zs.avail_out = (uInt)tmp.size();
                    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("deflate of block %zu did not finish\n", i); exit(2); }
                    comp[i].z.assign((const char *)tmp.data(), tmp.size() - zs.avail_out);
                    deflateEnd(&zs);
                }
            });
        for (auto &th : pool) th.join();
    }
    // the slab: block k is a copy of distinct block k % n_distinct; compressed streams packed back to back
    std::vector<InfPiece> pieces(n_blocks);
    std::string packed;
    unsigned long long tok = 0;
    for (size_t k = 0; k < n_blocks; ++k) {
        const std::string &z = comp[k % n_distinct].z;
        const unsigned int region = inf2_region_bytes((unsigned int)bs, fraction);
        pieces[k] = InfPiece{(unsigned long long)packed.size(), (unsigned int)z.size(), (unsigned int)bs, (unsigned long long)k * bs, 0u, (unsigned int)bs, 0ull, 0u, region, tok};
        tok += region;
        packed += z;
    }
    const size_t n = n_blocks * bs;
    printf("slab: %zu blocks (%zu distinct), %.2f GB of text, %.2f GB deflated (ratio %.2f), token buffer %.2f GB\n", n_blocks, n_distinct, n / 1e9, packed.size() / 1e9,
           (double)n / packed.size(), tok / 1e9);
    unsigned char *d_in, *d_out, *d_tok, *d_scratch;
    InfPiece *d_pieces;
    InfTokMeta *d_meta;
    unsigned int *d_status;
    CHK(hipMalloc(&d_in, packed.size() + INF_OVERRUN));
    CHK(hipMalloc(&d_out, n + 128));
    CHK(hipMalloc(&d_tok, tok + 64));
    CHK(hipMalloc(&d_scratch, 1 << 16));
    CHK(hipMalloc(&d_pieces, n_blocks * sizeof(InfPiece)));
    CHK(hipMalloc(&d_meta, n_blocks * sizeof(InfTokMeta)));
    CHK(hipMalloc(&d_status, 4));
    CHK(hipMemset(d_in, 0, packed.size() + INF_OVERRUN));
    CHK(hipMemcpy(d_in, packed.data(), packed.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_pieces, pieces.data(), n_blocks * sizeof(InfPiece), hipMemcpyHostToDevice));
    const unsigned int np = (unsigned int)n_blocks;
    std::vector<char> back(n);
    auto check = [&](const char *what) {
        unsigned int status = 0;
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
        size_t bad = 0, first_bad = (size_t)-1;
        for (size_t k = 0; k < n_blocks; ++k)
            if (memcmp(back.data() + k * bs, text.data() + (k % n_distinct) * bs, bs) != 0) { if (!bad) first_bad = k; ++bad; }
        printf("%s: status %u (code %u, block %u), %zu of %zu blocks differ%s\n", what, status, status & 255u, status >> 8, bad, n_blocks, bad ? "  <-- WRONG" : ": text IDENTICAL");
        if (bad) {
            const char *a = back.data() + first_bad * bs, *e = text.data() + (first_bad % n_distinct) * bs;
            size_t at = 0;
            while (at < bs && a[at] == e[at]) ++at;
            printf("   first difference: block %zu, byte %zu\n", first_bad, at);
            // the block itself, for a look at it on a host: the deflate stream and the text it must give
            const std::string &z = comp[first_bad % n_distinct].z;
            if (FILE *f = fopen("gpurun_out/inflate2_bad_block.deflate", "wb")) { fwrite(z.data(), 1, z.size(), f); fclose(f); }
            if (FILE *f = fopen("gpurun_out/inflate2_bad_block.text", "wb")) { fwrite(e, 1, bs, f); fclose(f); }
            if (FILE *f = fopen("gpurun_out/inflate2_bad_block.got", "wb")) { fwrite(a, 1, bs, f); fclose(f); }
        }
        return bad == 0 && status == 0;
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto timed = [&](const char *what, auto &&launch) {
        launch();
        CHK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int it = 0; it < reps; ++it) {
            CHK(hipEventRecord(e0));
            launch();
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            sum += ms;
        }
        printf("%-44s %8.2f ms (best %8.2f) = %7.1f GB/s of text\n", what, sum / reps, best, n / 1e9 / (sum / reps * 1e-3));
        return sum / reps;
    };
    bool ok = true;
    // (1) the single-kernel form
    CHK(hipMemset(d_status, 0, 4));
    CHK(hipMemset(d_out, 0, n));
    timed("bed_inflate_kernel (one lane per block)", [&] { hipLaunchKernelGGL(bed_inflate_kernel, dim3((np + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), 0, 0, d_in, d_pieces, np, d_out, d_scratch, d_status, (const InfTokMeta *)nullptr); });
    ok = check("bed_inflate_kernel") && ok;
    // (2) two phases
    CHK(hipMemset(d_status, 0, 4));
    CHK(hipMemset(d_out, 0, n));
    const float t1 = timed("bed_tokens_kernel (phase 1)", [&] { hipLaunchKernelGGL(bed_tokens_kernel, dim3((np + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), 0, 0, d_in, d_pieces, np, d_tok, d_meta, d_status); });
    const float t3 = timed("bed_inflate_kernel (blocks phase 1 gave up)", [&] { hipLaunchKernelGGL(bed_inflate_kernel, dim3((np + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), 0, 0, d_in, d_pieces, np, d_out, d_scratch, d_status, (const InfTokMeta *)d_meta); });
    const float t2 = timed("bed_resolve_kernel (phase 2)", [&] { hipLaunchKernelGGL(bed_resolve_kernel, dim3(np), dim3(64), 0, 0, d_pieces, np, d_tok, d_meta, d_out, d_scratch, d_status); });
    printf("two phases together: %.2f ms = %.1f GB/s of text\n", t1 + t2 + t3, n / 1e9 / ((t1 + t2 + t3) * 1e-3));
    ok = check("two-phase inflate") && ok;
    {
        std::vector<InfTokMeta> hm(n_blocks);
        CHK(hipMemcpy(hm.data(), d_meta, n_blocks * sizeof(InfTokMeta), hipMemcpyDeviceToHost));
        double s = 0, l = 0, mx = 0;
        size_t full = 0;
        for (const auto &m : hm) {
            if (m.err == INF2_FULL || m.err == INF2_WIDE) { ++full; continue; }
            s += m.n_seq; l += m.n_lit; mx = std::max(mx, 4.0 * m.n_seq + m.n_lit);
        }
        printf("tokens per block: %.0f sequence records, %.0f literals = %.1f KB (largest %.1f KB) of the %.1f KB region; %zu blocks did not fit and went through bed_inflate_kernel\n",
               s / (n_blocks - full), l / (n_blocks - full), (4 * s + l) / (n_blocks - full) / 1e3, mx / 1e3, inf2_region_bytes((unsigned int)bs, fraction) / 1e3, full);
    }
    return ok ? 0 : 1;
}
