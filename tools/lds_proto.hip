// Prototype of the LDS-staged, bit-TRANSPOSED scoring tile (round 4).
//
// Layout: a chunk is 8192 positions = 128 entries of 64 bits per plane; entry i, bit b <-> offset i + 128 b.  An offset d
// from the modified base is then an ENTRY shift: lane l, which owns entries l and l + 64, reads entries l + d and l + 64 + d
// of the plane's row in LDS (two ds_read_b64, always 8-byte aligned, conflict-free) — no funnel shift, one v_and per 32
// positions and constraint instead of v_alignbit + v_and.  Entries that wrap past the row's ends are the chunk's own
// entries shifted by one bit (+ one bit of the neighbouring chunk): 32 extension entries either side, built with the tile.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_proto lds_proto.hip ; run: ./lds_proto [chunks] [cands] [check]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const uint32_t __attribute__((address_space(4))) *cu32p;

constexpr int ENT = 128;            // entries per chunk and plane
constexpr int HALO = 32;            // extension entries either side (offsets in [-32, 31])
constexpr int ROW = ENT + 2 * HALO; // entries of one plane row in LDS
constexpr int ROW_B = ROW * 8;
constexpr int TILE_B = 4 * ROW_B;   // four is-X planes
constexpr int SEG = 16;             // chunks per workgroup
constexpr int MAXC = 16;            // candidates per pass
constexpr int PROG_DW = 32;         // [0] n_f, [1] n_r, [2..15] forward byte offsets, [18..31] reverse

struct Args {
    const uint2 *H, *L, *M, *U;     // transposed planes, 128 uint2 per chunk (M / U per slot: slot s at + s * plane stride)
    size_t state_stride;            // entries between the slots' state planes
    uint32_t n_chunks;              // scored chunks are [1, n_chunks - 1)
    const uint32_t *prog;           // [n_cand][PROG_DW]
    uint32_t n_cand;
    unsigned long long *out;        // [slot][n_cand][2]
    uint32_t flags;                 // probes: 1 = no tile build, 2 = no loads after the first chunk
};

struct Raw {
    uint2 h0, h1, l0, l1, m0, m1, u0, u1, hn, ln;
};

__device__ __forceinline__ Raw load_raw(const Args &a, uint32_t chunk, uint32_t slot, int lane) {
    Raw r;
    const size_t b = (size_t)chunk * ENT + lane;
    r.h0 = a.H[b]; r.h1 = a.H[b + 64];
    r.l0 = a.L[b]; r.l1 = a.L[b + 64];
    const size_t sb = b + slot * a.state_stride;
    r.m0 = a.M[sb]; r.m1 = a.M[sb + 64];
    r.u0 = a.U[sb]; r.u1 = a.U[sb + 64];
    // lanes 0..31: entry `lane` of the NEXT chunk; lanes 32..63: entry 64 + lane of the PREVIOUS chunk
    const size_t nb = lane < 32 ? b + ENT : b - ENT + 64;
    r.hn = a.H[nb]; r.ln = a.L[nb];
    return r;
}

__device__ __forceinline__ uint2 plane_of(int p, uint2 h, uint2 l) {
    uint2 r;
    switch (p) {
    case 0: r.x = ~h.x & ~l.x; r.y = ~h.y & ~l.y; break;   // A = 00
    case 1: r.x = ~h.x & l.x;  r.y = ~h.y & l.y;  break;   // C = 01
    case 2: r.x = h.x & l.x;   r.y = h.y & l.y;   break;   // G = 11
    default: r.x = h.x & ~l.x; r.y = h.y & ~l.y; break;    // T = 10
    }
    return r;
}

__device__ __forceinline__ void build_tile(char *tile, const Raw &r, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint2 e0 = plane_of(p, r.h0, r.l0), e1 = plane_of(p, r.h1, r.l1), en = plane_of(p, r.hn, r.ln);
        *(uint2 *)(tile + p * ROW_B + (HALO + lane) * 8) = e0;
        *(uint2 *)(tile + p * ROW_B + (HALO + 64 + lane) * 8) = e1;
        // extension: lanes 0..31 -> entry ENT + lane = own entry `lane` one bit down, top bit from the next chunk;
        //            lanes 32..63 -> entry lane - 64 (row index lane - 32) = own entry 64 + lane one bit up, low bit from the previous chunk
        uint2 x;
        if (lane < 32) {
            x.x = (e0.x >> 1) | (e0.y << 31);
            x.y = (e0.y >> 1) | (en.x << 31);
        } else {
            x.y = (e1.y << 1) | (e1.x >> 31);
            x.x = (e1.x << 1) | (en.y >> 31);
        }
        const int row_idx = lane < 32 ? HALO + ENT + lane : lane - 32;
        *(uint2 *)(tile + p * ROW_B + row_idx * 8) = x;
    }
}

typedef const volatile unsigned long long __attribute__((address_space(3))) *lds64p;
#define RD(i) { const unsigned long long q0_ = *(lds64p)(row + off[i]); const unsigned long long q1_ = *(lds64p)(row + off[i] + 512); r0[i] = make_uint2((uint32_t)q0_, (uint32_t)(q0_ >> 32)); r1[i] = make_uint2((uint32_t)q1_, (uint32_t)(q1_ >> 32)); }
#define B3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0x80)
#define AN2(i, j) { a0.x = B3(a0.x, r0[i].x, r0[j].x); a0.y = B3(a0.y, r0[i].y, r0[j].y); a1.x = B3(a1.x, r1[i].x, r1[j].x); a1.y = B3(a1.y, r1[i].y, r1[j].y); }
#define AN1(i) { a0.x &= r0[i].x; a0.y &= r0[i].y; a1.x &= r1[i].x; a1.y &= r1[i].y; }

// acc &= every constraint of one strand; off[] = byte offsets into the tile (plane row + entry shift), n wave-uniform
__device__ __forceinline__ void eval_strand(const uint32_t row, const uint32_t (&off)[8], uint32_t n, uint2 &a0, uint2 &a1) {
    // off[0]: the single constraint when n is odd; then pairs.  Every step is a plain wave-uniform `if`: reads of all
    // constraints first (16 ds_read_b64 in flight at most), then one three-input AND per pair and half-row.
    uint2 r0[8], r1[8];
    const uint32_t odd = n & 1u, np = n >> 1;
    if (odd) RD(0)
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (np > (uint32_t)i) { RD(1 + 2 * i) RD(2 + 2 * i) }
    if (odd) AN1(0)
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (np > (uint32_t)i) AN2(1 + 2 * i, 2 + 2 * i)
}

template <int CAN>
__device__ __forceinline__ void score_chunk(const Args &a, const char *tile, const Raw &r, uint32_t *cnt, int lane) {
    // canonical plane / its complement at offset 0 start the accumulators (the modified base's own constraint)
    const uint2 f0 = plane_of(CAN ? 1 : 0, r.h0, r.l0), f1 = plane_of(CAN ? 1 : 0, r.h1, r.l1);
    const uint2 g0 = plane_of(CAN ? 2 : 3, r.h0, r.l0), g1 = plane_of(CAN ? 2 : 3, r.h1, r.l1);
    const uint32_t row = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)tile + (HALO + lane) * 8;
    for (uint32_t k = 0; k < a.n_cand; ++k) {
        cu32p prog = (cu32p)(a.prog + (size_t)k * PROG_DW);
        const uint32_t nf = prog[0], nr = prog[1];
        uint32_t of[8], orv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { of[i] = prog[2 + i]; orv[i] = prog[18 + i]; }
        uint2 af0 = f0, af1 = f1, ar0 = g0, ar1 = g1;
        eval_strand(row, of, nf, af0, af1);
        eval_strand(row, orv, nr, ar0, ar1);
        const uint32_t s0 = af0.x | ar0.x, s1 = af0.y | ar0.y, s2 = af1.x | ar1.x, s3 = af1.y | ar1.y;
        const uint32_t n_mod = __popc(s0 & r.m0.x) + __popc(s1 & r.m0.y) + __popc(s2 & r.m1.x) + __popc(s3 & r.m1.y);
        const uint32_t n_non = __popc(s0 & r.u0.x) + __popc(s1 & r.u0.y) + __popc(s2 & r.u1.x) + __popc(s3 & r.u1.y);
        atomicAdd(&cnt[k * 64 + lane], n_mod | (n_non << 16));
    }
}

__global__ __launch_bounds__(256) void score_t_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *cnt = (uint32_t *)smem;                              // [MAXC][64] packed n_mod | n_non << 16
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *tile = smem + MAXC * 64 * 4 + wave * TILE_B;
    const uint32_t slot = blockIdx.y;
    const uint32_t c0 = 1 + blockIdx.x * SEG;
    const uint32_t c1 = min(c0 + SEG, a.n_chunks - 1);
    for (uint32_t i = threadIdx.x; i < MAXC * 64; i += 256) cnt[i] = 0;
    __syncthreads();
    uint32_t ck = c0 + wave;
    if (ck < c1) {
        Raw cur = load_raw(a, ck, slot, lane);
        for (; ck < c1; ck += 4) {
            Raw nxt;
            const bool more = ck + 4 < c1;
            if (more && !(a.flags & 2)) nxt = load_raw(a, ck + 4, slot, lane);
            if (!(a.flags & 1)) build_tile(tile, cur, lane);
            if (slot) score_chunk<1>(a, tile, cur, cnt, lane);
            else score_chunk<0>(a, tile, cur, cnt, lane);
            if (more && !(a.flags & 2)) cur = nxt;
        }
    }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < a.n_cand * 2; idx += 256) {
        const uint32_t k = idx >> 1, which = idx & 1;
        uint32_t s = 0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t v = cnt[k * 64 + l];
            s += which ? (v >> 16) : (v & 0xFFFFu);
        }
        if (s) atomicAdd(a.out + ((size_t)slot * a.n_cand + k) * 2 + which, (unsigned long long)s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- host
struct Cand {
    std::vector<int> off;     // offsets from the modified base of the specified positions (excluding 0)
    std::vector<int> base;    // 0 A, 1 C, 2 G, 3 T
};

int main(int argc, char **argv) {
    const uint32_t n_chunks = argc > 1 ? (uint32_t)atoi(argv[1]) : 122072;   // ~1 Gbp
    const uint32_t n_cand = argc > 2 ? (uint32_t)atoi(argv[2]) : 10;
    const bool check = argc > 3 && atoi(argv[3]) != 0;
    const size_t n_pos = (size_t)n_chunks * 8192;
    std::mt19937_64 rng(12345);
    // candidates: 3..8 specified positions over a span of 4..15, canonical base at the modified position
    std::vector<Cand> cands[2];
    std::vector<uint32_t> prog(2 * (size_t)n_cand * PROG_DW, 0);
    double mean_reads = 0;
    for (int slot = 0; slot < 2; ++slot)
        for (uint32_t k = 0; k < n_cand; ++k) {
            Cand c;
            const int span = 4 + (int)(rng() % 12), nspec = 3 + (int)(rng() % 6);
            std::vector<int> pos;
            for (int j = 0; j < span; ++j) pos.push_back(j);
            for (int j = span - 1; j > 0; --j) std::swap(pos[j], pos[rng() % (j + 1)]);
            pos.resize(std::min(nspec, span));
            const int modpos = pos[0];
            for (size_t j = 1; j < pos.size() && c.off.size() < 7; ++j) { c.off.push_back(pos[j] - modpos); c.base.push_back((int)(rng() % 4)); }
            cands[slot].push_back(c);
            uint32_t *p = prog.data() + ((size_t)slot * n_cand + k) * PROG_DW;
            p[0] = p[1] = (uint32_t)c.off.size();
            mean_reads += 2.0 * c.off.size();
            const size_t shift = (c.off.size() & 1) ? 0 : 1;     // even: slot 0 (the single) stays unused
            for (size_t j = 0; j < c.off.size(); ++j) {
                p[2 + shift + j] = (uint32_t)(c.base[j] * ROW_B + c.off[j] * 8);                 // forward: base at +off
                p[18 + shift + j] = (uint32_t)((3 - c.base[j]) * ROW_B + (-c.off[j]) * 8);       // reverse: complement (A<->T, C<->G = 3 - b) at -off
            }
        }
    mean_reads /= 2.0 * n_cand;
    // data
    std::vector<uint64_t> H((size_t)n_chunks * ENT), L(H.size()), M(2 * H.size()), U(2 * H.size());
    std::vector<uint8_t> seq;
    if (check) seq.resize(n_pos);
    for (size_t c = 0; c < n_chunks; ++c)
        for (int i = 0; i < ENT; ++i) {
            const uint64_t h = rng(), l = rng(), m = rng() & rng(), u = rng() & ~m;
            H[c * ENT + i] = h; L[c * ENT + i] = l;
            // slot 0 (6mA): rows on A (fwd) and T (rev): h=0,l=0 / h=1,l=0 -> ~l ; slot 1 (5mC): C and G -> l
            M[c * ENT + i] = m & ~l; U[c * ENT + i] = u & ~l;
            M[H.size() + c * ENT + i] = m & l; U[H.size() + c * ENT + i] = u & l;
            if (check)
                for (int b = 0; b < 64; ++b) {
                    const int hb = (h >> b) & 1, lb = (l >> b) & 1;
                    seq[c * 8192 + i + 128 * b] = (uint8_t)(hb == 0 ? (lb ? 1 : 0) : (lb ? 2 : 3));
                }
        }
    uint2 *dH, *dL, *dM, *dU; uint32_t *dprog; unsigned long long *dout;
    CHK(hipMalloc(&dH, H.size() * 8)); CHK(hipMalloc(&dL, H.size() * 8)); CHK(hipMalloc(&dM, M.size() * 8)); CHK(hipMalloc(&dU, U.size() * 8));
    CHK(hipMalloc(&dprog, prog.size() * 4)); CHK(hipMalloc(&dout, 2 * (size_t)n_cand * 2 * 8));
    CHK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dL, L.data(), H.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dU, U.data(), U.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dprog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(dout, 0, 2 * (size_t)n_cand * 2 * 8));
    const uint32_t flags = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
    Args a{dH, dL, dM, dU, H.size(), n_chunks, dprog, n_cand, dout, flags};
    const uint32_t n_seg = (n_chunks - 2 + SEG - 1) / SEG;
    const size_t lds = MAXC * 64 * 4 + 4 * TILE_B;
    CHK(hipFuncSetAttribute((const void *)score_t_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // slot 1 uses the second half of the program table
    auto launch = [&]() {
        Args a0 = a; a0.prog = dprog;
        // (both slots in one grid: blockIdx.y; the kernel indexes programs by slot itself in the product — here the same table serves both)
        hipLaunchKernelGGL(score_t_kernel, dim3(n_seg, 2), dim3(256), lds, 0, a0);
    };
    launch();
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> out(2 * (size_t)n_cand * 2);
    CHK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
    if (check) {
        // CPU reference over the scored chunks; both slots use the slot-0 program table (see launch)
        int bad = 0;
        for (int slot = 0; slot < 2; ++slot)
            for (uint32_t k = 0; k < n_cand; ++k) {
                const Cand &c = cands[0][k];
                const int canon = slot ? 1 : 0;
                unsigned long long nm = 0, nn = 0;
                for (size_t p = 8192; p < n_pos - 8192; ++p) {
                    bool f = seq[p] == canon, r = seq[p] == 3 - canon;
                    if (!f && !r) continue;
                    for (size_t j = 0; j < c.off.size() && (f || r); ++j) {
                        if (f && seq[p + c.off[j]] != c.base[j]) f = false;
                        if (r && seq[p - c.off[j]] != 3 - c.base[j]) r = false;
                    }
                    if (!f && !r) continue;
                    const size_t ch = p / 8192, o = p % 8192, e = ch * ENT + (o & 127), b = o >> 7;
                    const size_t so = slot * H.size();
                    nm += (M[so + e] >> b) & 1;
                    nn += (U[so + e] >> b) & 1;
                }
                const unsigned long long gm = out[((size_t)slot * n_cand + k) * 2], gn = out[((size_t)slot * n_cand + k) * 2 + 1];
                if (gm != nm || gn != nn) { ++bad; printf("MISMATCH slot %d cand %u: gpu %llu %llu cpu %llu %llu (n=%zu)\n", slot, k, gm, gn, nm, nn, c.off.size()); }
            }
        printf("check: %s (%u candidates x 2 slots, %u chunks)\n", bad ? "FAILED" : "ok", n_cand, n_chunks);
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CHK(hipDeviceSynchronize());
    const int iters = 50;
    CHK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bp = (double)(n_chunks - 2) * 8192;
    printf("flags %u chunks %u cands/slot %u mean reads/strand %.2f: %.4f ms per launch, %.3e motif-sites/s, %.1f GB/s algorithmic (0.5 B/bp/slot)\n",
           flags, n_chunks, n_cand, mean_reads / 2, ms, 2.0 * bp * n_cand * 2 / (ms * 1e-3), bp * 0.5 * 2 / (ms * 1e-3) / 1e9);
    return 0;
}
