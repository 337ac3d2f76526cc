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

constexpr int ENT = 64;             // entries (128 bits each) per chunk and plane: entry i, bit b <-> offset i + 64 b
#ifndef HALO
#define HALO 32                     // extension entries either side (offsets in [-HALO, HALO - 1])
#endif
#ifndef BOTH
#define BOTH 0                      // 1: the reads of both strands are issued before the first is consumed
#endif
#ifndef WAVES
#define WAVES 5
#endif
constexpr int ROW = ENT + 2 * HALO; // entries of one plane row in LDS
constexpr int ROW_B = ROW * 16;
constexpr int TILE_B = 4 * ROW_B;   // four is-X planes
constexpr int SEG = 16;             // chunks per workgroup
constexpr int MAXC = 16;            // candidates per pass
constexpr int PROG_DW = 32;         // [0] n_f, [1] n_r, [2..9] forward byte offsets, [10..17] reverse

struct Args {
    const uint4 *H, *L, *M, *U;     // transposed planes, 64 uint4 per chunk (M / U per slot: slot s at + s * plane stride)
    size_t state_stride;            // entries between the slots' state planes
    uint32_t n_chunks;              // scored chunks are [1, n_chunks - 1)
    const uint32_t *prog;           // [n_cand][PROG_DW]
    uint32_t n_cand;
    unsigned long long *out;        // [slot][n_cand][2]
    uint32_t flags;                 // probes: 1 = no tile build, 2 = no loads after the first chunk
};

struct Raw {
    uint4 h, l, m, u, hn, ln;
};

__device__ __forceinline__ Raw load_raw(const Args &a, uint32_t chunk, uint32_t slot, int lane) {
    Raw r;
    const size_t b = (size_t)chunk * ENT + lane;
    r.h = a.H[b];
    r.l = a.L[b];
    const size_t sb = b + slot * a.state_stride;
    r.m = a.M[sb];
    r.u = a.U[sb];
    // lanes 0..HALO-1: entry `lane` of the NEXT chunk; lanes 64-HALO..63: entry `lane` of the PREVIOUS chunk
    r.hn = r.ln = make_uint4(0, 0, 0, 0);
    if (lane < HALO || lane >= 64 - HALO) {
        const size_t nb = lane < HALO ? b + ENT : b - ENT;
        r.hn = a.H[nb];
        r.ln = a.L[nb];
    }
    return r;
}

__device__ __forceinline__ uint32_t plane1(int p, uint32_t h, uint32_t l) {
    return p == 0 ? ~h & ~l : p == 1 ? ~h & l : p == 2 ? h & l : h & ~l;      // A = 00, C = 01, G = 11, T = 10
}
__device__ __forceinline__ uint4 plane_of(int p, uint4 h, uint4 l) {
    return make_uint4(plane1(p, h.x, l.x), plane1(p, h.y, l.y), plane1(p, h.z, l.z), plane1(p, h.w, l.w));
}

__device__ __forceinline__ void build_tile(char *tile, const Raw &r, int lane) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint4 e = plane_of(p, r.h, r.l), en = plane_of(p, r.hn, r.ln);
        *(uint4 *)(tile + p * ROW_B + (HALO + lane) * 16) = e;
        // extension: lanes 0..HALO-1 -> entry ENT + lane = own entry one bit down, top bit from the next chunk;
        //            lanes 64-HALO..63 -> entry lane - 64 = own entry one bit up, low bit from the previous chunk
        if (lane < HALO) {
            uint4 x;
            x.x = __builtin_amdgcn_alignbit(e.y, e.x, 1);
            x.y = __builtin_amdgcn_alignbit(e.z, e.y, 1);
            x.z = __builtin_amdgcn_alignbit(e.w, e.z, 1);
            x.w = __builtin_amdgcn_alignbit(en.x, e.w, 1);
            *(uint4 *)(tile + p * ROW_B + (HALO + ENT + lane) * 16) = x;
        } else if (lane >= 64 - HALO) {
            uint4 x;
            x.w = __builtin_amdgcn_alignbit(e.w, e.z, 31);
            x.z = __builtin_amdgcn_alignbit(e.z, e.y, 31);
            x.y = __builtin_amdgcn_alignbit(e.y, e.x, 31);
            x.x = __builtin_amdgcn_alignbit(e.x, en.w, 31);
            *(uint4 *)(tile + p * ROW_B + (lane - (64 - HALO)) * 16) = x;
        }
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const volatile u32x4 __attribute__((address_space(3))) *lds128p;
#define RD(i) { const u32x4 q_ = *(lds128p)(row + off[i]); r[i] = make_uint4(q_.x, q_.y, q_.z, q_.w); }
#define B3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0x80)
#define AN2(i, j) { acc.x = B3(acc.x, r[i].x, r[j].x); acc.y = B3(acc.y, r[i].y, r[j].y); acc.z = B3(acc.z, r[i].z, r[j].z); acc.w = B3(acc.w, r[i].w, r[j].w); }
#define AN1(i) { acc.x &= r[i].x; acc.y &= r[i].y; acc.z &= r[i].z; acc.w &= r[i].w; }

// One strand: acc = init & every constraint.  A straight-line body per constraint count behind ONE multi-way branch (a compare
// tree on the wave-uniform n): all reads of the strand in flight together, one three-input AND per pair of constraints and
// dword, no flags, no copies (the if-chains of the first version cost 48 scalar and 71 vector instructions per candidate
// and chunk, profiles of round 4).
#define RDN(i) const u32x4 q##i = *(lds128p)(row + off[i]);
#define A2(i, j, src) acc.x = B3(src.x, q##i.x, q##j.x); acc.y = B3(src.y, q##i.y, q##j.y); acc.z = B3(src.z, q##i.z, q##j.z); acc.w = B3(src.w, q##i.w, q##j.w);
#define A1(i, src) acc.x = src.x & q##i.x; acc.y = src.y & q##i.y; acc.z = src.z & q##i.z; acc.w = src.w & q##i.w;
__device__ __forceinline__ uint4 eval_strand(const uint32_t row, const uint32_t (&off)[8], uint32_t n, const uint4 init) {
    uint4 acc = init;
    switch (n) {
    case 0: break;
    case 1: { RDN(0) A1(0, init) } break;
    case 2: { RDN(0) RDN(1) A2(0, 1, init) } break;
    case 3: { RDN(0) RDN(1) RDN(2) A2(0, 1, init) A1(2, acc) } break;
    case 4: { RDN(0) RDN(1) RDN(2) RDN(3) A2(0, 1, init) A2(2, 3, acc) } break;
    case 5: { RDN(0) RDN(1) RDN(2) RDN(3) RDN(4) A2(0, 1, init) A2(2, 3, acc) A1(4, acc) } break;
    case 6: { RDN(0) RDN(1) RDN(2) RDN(3) RDN(4) RDN(5) A2(0, 1, init) A2(2, 3, acc) A2(4, 5, acc) } break;
    default: { RDN(0) RDN(1) RDN(2) RDN(3) RDN(4) RDN(5) RDN(6) A2(0, 1, init) A2(2, 3, acc) A2(4, 5, acc) A1(6, acc) } break;
    }
    return acc;
}

template <int CAN>
__device__ __forceinline__ void score_chunk(const Args &a, const char *tile, const Raw &raw, uint32_t *cnt, int lane) {
    // canonical plane / its complement at offset 0 start the accumulators (the modified base's own constraint)
    const uint4 f0 = plane_of(CAN ? 1 : 0, raw.h, raw.l), g0 = plane_of(CAN ? 2 : 3, raw.h, raw.l);
    const uint32_t row = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)tile + (HALO + lane) * 16;
    // the next candidate's program is requested (scalar loads) before the current one is evaluated
    uint32_t nf, nr, of[8], orv[8];
    {
        cu32p prog = (cu32p)a.prog;
        nf = prog[0]; nr = prog[1];
#pragma unroll
        for (int i = 0; i < 8; ++i) { of[i] = prog[2 + i]; orv[i] = prog[10 + i]; }
    }
    for (uint32_t k = 0; k < a.n_cand; ++k) {
        cu32p nxt = (cu32p)(a.prog + (size_t)min(k + 1, a.n_cand - 1) * PROG_DW);
        const uint32_t nf2 = nxt[0], nr2 = nxt[1];
        uint32_t of2[8], orv2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { of2[i] = nxt[2 + i]; orv2[i] = nxt[10 + i]; }
        const uint4 af = eval_strand(row, of, nf, f0);
        const uint4 ar = eval_strand(row, orv, nr, g0);
        const uint32_t s0 = af.x | ar.x, s1 = af.y | ar.y, s2 = af.z | ar.z, s3 = af.w | ar.w;
        const uint32_t n_mod = __popc(s0 & raw.m.x) + __popc(s1 & raw.m.y) + __popc(s2 & raw.m.z) + __popc(s3 & raw.m.w);
        const uint32_t n_non = __popc(s0 & raw.u.x) + __popc(s1 & raw.u.y) + __popc(s2 & raw.u.z) + __popc(s3 & raw.u.w);
        atomicAdd(&cnt[k * 64 + lane], n_mod | (n_non << 16));
        nf = nf2; nr = nr2;
#pragma unroll
        for (int i = 0; i < 8; ++i) { of[i] = of2[i]; orv[i] = orv2[i]; }
    }
}

__global__ __launch_bounds__(256, WAVES) void score_t_kernel(Args a) {
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
    Raw cur;
    bool have = false;
    for (uint32_t ck = c0 + wave; ck < c1; ck += 4) {
        if (!have || !(a.flags & 2)) cur = load_raw(a, ck, slot, lane);
        have = true;
        if (!(a.flags & 1)) build_tile(tile, cur, lane);
        if (slot) score_chunk<1>(a, tile, cur, cnt, lane);
        else score_chunk<0>(a, tile, cur, cnt, lane);
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
            const size_t shift = 0;
            for (size_t j = 0; j < c.off.size(); ++j) {
                p[2 + shift + j] = (uint32_t)(c.base[j] * ROW_B + c.off[j] * 16);                 // forward: base at +off
                p[10 + shift + j] = (uint32_t)((3 - c.base[j]) * ROW_B + (-c.off[j]) * 16);       // reverse: complement (A<->T, C<->G = 3 - b) at -off
            }
        }
    mean_reads /= 2.0 * n_cand;
    // data
    // an entry is 128 bits = two uint64 halves: entry i, bit b <-> offset i + 64 b
    const size_t NE = (size_t)n_chunks * ENT * 2;
    std::vector<uint64_t> H(NE), L(NE), M(2 * NE), U(2 * NE);
    std::vector<uint8_t> seq;
    if (check) seq.resize(n_pos);
    for (size_t c = 0; c < n_chunks; ++c)
        for (int i = 0; i < ENT; ++i)
            for (int half = 0; half < 2; ++half) {
                const uint64_t h = rng(), l = rng(), m = rng() & rng(), u = rng() & ~m;
                const size_t w = (c * ENT + i) * 2 + half;
                H[w] = h; L[w] = l;
                // slot 0 (6mA): rows on A (fwd) and T (rev): l = 0; slot 1 (5mC): C and G: l = 1
                M[w] = m & ~l; U[w] = u & ~l;
                M[NE + w] = m & l; U[NE + w] = u & l;
                if (check)
                    for (int b = 0; b < 64; ++b) {
                        const int hb = (h >> b) & 1, lb = (l >> b) & 1;
                        seq[c * 8192 + i + 64 * (64 * half + b)] = (uint8_t)(hb == 0 ? (lb ? 1 : 0) : (lb ? 2 : 3));
                    }
            }
    uint4 *dH, *dL, *dM, *dU; uint32_t *dprog; unsigned long long *dout;
    CHK(hipMalloc(&dH, H.size() * 8)); CHK(hipMalloc(&dL, H.size() * 8)); CHK(hipMalloc(&dM, M.size() * 8)); CHK(hipMalloc(&dU, U.size() * 8));
    CHK(hipMalloc(&dprog, prog.size() * 4)); CHK(hipMalloc(&dout, 2 * (size_t)n_cand * 2 * 8));
    CHK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dL, L.data(), H.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dU, U.data(), U.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dprog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(dout, 0, 2 * (size_t)n_cand * 2 * 8));
    const uint32_t flags = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
    Args a{dH, dL, dM, dU, NE / 2, n_chunks, dprog, n_cand, dout, flags};
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
                    const size_t ch = p / 8192, o = p % 8192, bit = o >> 6, w = (ch * ENT + (o & 63)) * 2 + (bit >> 6);
                    const size_t so = slot * NE;
                    nm += (M[so + w] >> (bit & 63)) & 1;
                    nn += (U[so + w] >> (bit & 63)) & 1;
                }
                const unsigned long long gm = out[((size_t)slot * n_cand + k) * 2], gn = out[((size_t)slot * n_cand + k) * 2 + 1];
                if (gm != nm || gn != nn) { ++bad; printf("MISMATCH slot %d cand %u: gpu %llu %llu cpu %llu %llu (n=%zu)\n", slot, k, gm, gn, nm, nn, c.off.size()); }
            }
        printf("check: %s (%u candidates x 2 slots, %u chunks)\n", bad ? "FAILED" : "ok", n_cand, n_chunks);
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = argc > 5 ? atoi(argv[5]) : 50;
    for (int i = 0; i < (iters < 20 ? iters : 20); ++i) launch();
    CHK(hipDeviceSynchronize());
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
