// Prototype (round 4): DEFLATE decoding of BGZF blocks ON THE GPU, one lane per block (<= 64 KiB of text each), Huffman
// tables of the lane in LDS (canonical decoding from per-length counts, like zlib's contrib/puff), output straight to HBM.
// Host: makes bedMethyl-like text, deflates it in BGZF-sized blocks with zlib, checks the device's text byte for byte, times.
// Build: hipcc -O3 --offload-arch=gfx950 -o inflate_proto inflate_proto.hip -lz ; run: ./inflate_proto [MB of text]
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Piece {
    unsigned long long in_off;       // first byte of the raw deflate stream
    unsigned int in_len, out_len;
    unsigned long long out_off;
};

constexpr int LANES = 64;            // blocks per workgroup (one wave)
constexpr int MAXL = 288, MAXD = 30;

// per-lane tables in LDS, entry-major so that lane l of the wave touches word [entry][l]
struct Tables {
    unsigned short lcount[16][LANES], lsym[MAXL][LANES], dcount[16][LANES], dsym[MAXD][LANES];
};

struct Bits {
    const unsigned char *p, *end;
    unsigned long long buf;
    int cnt;
    __device__ __forceinline__ void refill() {              // (the input buffer has 8 readable bytes after its end)
        unsigned long long w;
        memcpy(&w, p, 8);                                      // one unaligned 8-byte load
        buf |= w << cnt;
        const int take = (63 - cnt) >> 3;                      // whole bytes that fit
        p += take;
        cnt += take * 8;
    }
    __device__ __forceinline__ unsigned int get(int n) {       // n <= 16
        if (cnt < 32) refill();
        const unsigned int v = (unsigned int)(buf & ((1ull << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
};

// canonical Huffman code from code lengths (puff.c: construct); returns > 0 for an incomplete, < 0 for an over-subscribed set
template <int NSYM>
__device__ int construct(unsigned short (*count)[LANES], unsigned short (*symbol)[LANES], const unsigned char *length, int n, int lane) {
    for (int len = 0; len <= 15; ++len) count[len][lane] = 0;
    for (int s = 0; s < n; ++s) count[length[s]][lane] += 1;
    if (count[0][lane] == n) return 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= count[len][lane];
        if (left < 0) return left;
    }
    unsigned short offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len][lane];
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) symbol[offs[length[s]]++][lane] = (unsigned short)s;
    return left;
}

struct Counts { unsigned int c[16]; };      // per-length code counts in registers (static indices in the unrolled walk)
__device__ __forceinline__ Counts load_counts(unsigned short (*count)[LANES], int lane) {
    Counts k;
#pragma unroll
    for (int len = 0; len < 16; ++len) k.c[len] = count[len][lane];
    return k;
}

__device__ __forceinline__ int decode(Bits &b, const Counts &k, unsigned short (*symbol)[LANES], int lane) {
    if (b.cnt < 32) b.refill();
    int code = 0, first = 0, index = 0;
    unsigned int bits = (unsigned int)b.buf;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int c = (int)k.c[len];
        if (code - c < first) {
            b.buf >>= len;
            b.cnt -= len;
            return symbol[index + (code - first)][lane];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__constant__ unsigned short LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ unsigned char LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ unsigned short DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ unsigned char DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ unsigned char CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// status: 0 ok, else the first thing that went wrong
__global__ __launch_bounds__(LANES) void inflate_kernel(const unsigned char *__restrict__ in, const Piece *__restrict__ pieces, unsigned int n_pieces,
                                                        unsigned char *__restrict__ out, unsigned int *__restrict__ status, int mode) {
    __shared__ Tables T;
    const int lane = threadIdx.x;
    const unsigned int i = blockIdx.x * LANES + lane;
    if (i >= n_pieces) return;
    const Piece pc = pieces[i];
    Bits b{in + pc.in_off, in + pc.in_off + pc.in_len, 0ull, 0};
    unsigned char *dst = out + pc.out_off;
    unsigned int o = 0;
    int err = 0, last = 0;
    unsigned char lengths[MAXL + MAXD];
    while (!last && !err) {
        last = (int)b.get(1);
        const int type = (int)b.get(2);
        if (type == 0) {                                         // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const unsigned int len = b.get(16), nlen = b.get(16);
            if ((len ^ 0xFFFFu) != nlen || o + len > pc.out_len) { err = 2; break; }
            for (unsigned int k = 0; k < len; ++k) dst[o++] = (unsigned char)b.get(8);
            continue;
        }
        if (type == 3) { err = 3; break; }
        if (type == 1) {                                         // fixed code
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            construct<MAXL>(T.lcount, T.lsym, lengths, 288, lane);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            construct<MAXD>(T.dcount, T.dsym, lengths, 30, lane);
        } else {                                                 // dynamic code
            const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
            if (nlen > 286 || ndist > 30) { err = 4; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[CLORDER[idx]] = (unsigned char)b.get(3);
            for (; idx < 19; ++idx) lengths[CLORDER[idx]] = 0;
            if (construct<MAXL>(T.lcount, T.lsym, lengths, 19, lane) != 0) { err = 5; break; }
            const Counts kc = load_counts(T.lcount, lane);
            idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(b, kc, T.lsym, lane);
                if (sym < 0) { err = 6; break; }
                if (sym < 16) lengths[idx++] = (unsigned char)sym;
                else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { err = 7; break; }
                        len = lengths[idx - 1];
                        rep = 3 + (int)b.get(2);
                    } else if (sym == 17) rep = 3 + (int)b.get(3);
                    else rep = 11 + (int)b.get(7);
                    if (idx + rep > nlen + ndist) { err = 8; break; }
                    while (rep--) lengths[idx++] = (unsigned char)len;
                }
            }
            if (err) break;
            if (lengths[256] == 0) { err = 9; break; }
            int r = construct<MAXL>(T.lcount, T.lsym, lengths, nlen, lane);
            if (r < 0 || (r > 0 && nlen - T.lcount[0][lane] != 1)) { err = 10; break; }
            r = construct<MAXD>(T.dcount, T.dsym, lengths + nlen, ndist, lane);
            if (r < 0 || (r > 0 && ndist - T.dcount[0][lane] != 1)) { err = 11; break; }
        }
        const Counts kl = load_counts(T.lcount, lane), kd = load_counts(T.dcount, lane);
        for (;;) {                                               // the block's symbols
            int sym = decode(b, kl, T.lsym, lane);
            if (sym < 0) { err = 12; break; }
            if (sym < 256) {
                if (o >= pc.out_len) { err = 13; break; }
                if (mode < 2) dst[o] = (unsigned char)sym;
                o++;
            } else if (sym == 256) break;
            else {
                sym -= 257;
                if (sym >= 29) { err = 14; break; }
                const unsigned int len = LBASE[sym] + b.get(LEXT[sym]);
                const int ds = decode(b, kd, T.dsym, lane);
                if (ds < 0 || ds >= 30) { err = 15; break; }
                const unsigned int dist = DBASE[ds] + b.get(DEXT[ds]);
                if (dist > o || o + len > pc.out_len) { err = 16; break; }
                if (mode >= 1) { o += len; continue; }                 // probe: the match is not copied
                const unsigned char *src = dst + o - dist;
                unsigned char *d = dst + o;
                // A match is copied in 8-byte words (unaligned global accesses are fine).  Words may write up to 7 bytes past
                // the match: those bytes are rewritten by what follows before anything can refer to them (a back-reference
                // only reaches positions below the current end of the output) — hence the room test.  With a distance of 32
                // or more, up to four words are LOADED before the first is stored: one round trip through memory instead of
                // one per word (and one per BYTE in the tail: the byte loop was 56 % of a lane's time, tools/inflate_proto.hip).
                if (dist >= 32 && o + len + 8 <= pc.out_len) {
                    for (unsigned int k = 0; k < len; k += 32) {
                        const unsigned int rem = len - k;
                        unsigned long long v0, v1 = 0, v2 = 0, v3 = 0;
                        memcpy(&v0, src + k, 8);
                        if (rem > 8) memcpy(&v1, src + k + 8, 8);
                        if (rem > 16) memcpy(&v2, src + k + 16, 8);
                        if (rem > 24) memcpy(&v3, src + k + 24, 8);
                        memcpy(d + k, &v0, 8);
                        if (rem > 8) memcpy(d + k + 8, &v1, 8);
                        if (rem > 16) memcpy(d + k + 16, &v2, 8);
                        if (rem > 24) memcpy(d + k + 24, &v3, 8);
                    }
                } else if (dist >= 8 && o + len + 8 <= pc.out_len) {
                    for (unsigned int k = 0; k < len; k += 8) {
                        unsigned long long v;
                        memcpy(&v, src + k, 8);
                        memcpy(d + k, &v, 8);
                    }
                } else {
                    for (unsigned int k = 0; k < len; ++k) d[k] = src[k];
                }
                o += len;
            }
        }
    }
    if (!err && o != pc.out_len) err = 17;
    if (err) atomicCAS(status, 0u, (unsigned int)err | (i << 8));
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const int level = argc > 2 ? atoi(argv[2]) : 6;
    std::mt19937_64 rng(7);
    std::string text;
    text.reserve(mb << 20);
    unsigned long long pos = 0;
    int contig = 0;
    char line[256];
    while (text.size() < (mb << 20)) {                            // modkit bedMethyl-like rows
        pos += 1 + rng() % 3;
        if (pos > 2000000) { pos = rng() % 5; ++contig; }
        const int cov = 10 + (int)(rng() % 40), pct = (int)(rng() % 10000), nmod = cov * pct / 10000;
        const int k = snprintf(line, sizeof line, "contig_%05d\t%llu\t%llu\t%s\t%d\t%c\t%llu\t%llu\t255,0,0\t%d\t%d.%02d\t%d\t%d\t0\t0\t0\t0\t0\n", contig, pos,
                               pos + 1, (rng() & 1) ? "a" : "m", cov, (rng() & 1) ? '+' : '-', pos, pos + 1, cov, pct / 100, pct % 100, nmod, cov - nmod);
        text.append(line, (size_t)k);
    }
    const size_t n = text.size(), bs = 0xFF00;
    std::vector<Piece> pieces;
    std::string comp;
    std::vector<unsigned char> tmp(compressBound(bs) + 64);
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t off = 0; off < n; off += bs) {
        const size_t len = std::min(bs, n - off);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef *)text.data() + off;
        zs.avail_in = (uInt)len;
        zs.next_out = tmp.data();
        zs.avail_out = (uInt)tmp.size();
        deflate(&zs, Z_FINISH);
        const size_t clen = tmp.size() - zs.avail_out;
        deflateEnd(&zs);
        pieces.push_back({comp.size(), (unsigned int)clen, (unsigned int)len, off});
        comp.append((const char *)tmp.data(), clen);
    }
    printf("text %.1f MB, deflated %.1f MB (level %d, %zu blocks) in %.1f s on one host thread\n", n / 1e6, comp.size() / 1e6, level, pieces.size(),
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    // host inflate rate for reference (one thread)
    {
        std::vector<char> o(bs);
        const auto t1 = std::chrono::steady_clock::now();
        const size_t nb = std::min<size_t>(pieces.size(), 2000);
        for (size_t i = 0; i < nb; ++i) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = (Bytef *)comp.data() + pieces[i].in_off;
            zs.avail_in = pieces[i].in_len;
            zs.next_out = (Bytef *)o.data();
            zs.avail_out = (uInt)bs;
            inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        printf("zlib inflate on one host thread: %.0f MB/s of text\n", nb * bs / 1e6 / dt);
    }
    unsigned char *d_in, *d_out;
    Piece *d_pieces;
    unsigned int *d_status;
    CHK(hipMalloc(&d_in, comp.size() + 16));
    CHK(hipMalloc(&d_out, n + 16));
    CHK(hipMalloc(&d_pieces, pieces.size() * sizeof(Piece)));
    CHK(hipMalloc(&d_status, 4));
    CHK(hipMemcpy(d_in, comp.data(), comp.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_pieces, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice));
    CHK(hipMemset(d_status, 0, 4));
    CHK(hipMemset(d_out, 0, n));
    const unsigned int np = (unsigned int)pieces.size();
    int mode = 0;
    auto launch = [&]() { hipLaunchKernelGGL(inflate_kernel, dim3((np + LANES - 1) / LANES), dim3(LANES), 0, 0, d_in, d_pieces, np, d_out, d_status, mode); };
    launch();
    CHK(hipDeviceSynchronize());
    unsigned int status = 0;
    CHK(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
    std::vector<char> back(n);
    CHK(hipMemcpy(back.data(), d_out, n, hipMemcpyDeviceToHost));
    const bool same = memcmp(back.data(), text.data(), n) == 0;
    printf("device inflate: status %u (error %u in block %u), text %s\n", status, status & 255, status >> 8, same ? "IDENTICAL" : "DIFFERS");
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    for (int it = 0; it < 3; ++it) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("device inflate: %.2f ms for %.1f MB of text = %.1f GB/s (%zu blocks, one lane each)\n", ms, n / 1e6, n / 1e9 / (ms * 1e-3), pieces.size());
    for (mode = 1; mode <= 2; ++mode) {                           // probes: where a lane's time goes
        launch();
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int it = 0; it < 3; ++it) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("  probe %s: %.2f ms\n", mode == 1 ? "matches decoded but not copied" : "nothing written at all", ms / 3);
    }
    return same && status == 0 ? 0 : 1;
}
