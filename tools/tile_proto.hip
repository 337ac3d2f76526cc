// Prototype of the "bit-row transposed tile" inner loop: offsets become register renaming.
// Proxy only — random plane data, no parity — to measure what the formulation can issue on gfx950:
// lane block = 512 positions as 16 registers per plane (+ 2*DMAX pre-shifted wrap registers), a constraint (plane p,
// offset d) = 16 ANDs behind a scalar switch, count = popcount against M / U words.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

constexpr int R = 16, DMAX = 12, EXT = R + 2 * DMAX, NCODE = 4 * (2 * DMAX + 1);
typedef const uint8_t __attribute__((address_space(4))) *cu8p;

// The accumulators live in FIXED physical registers (strand 0: v200..v215, strand 1: v216..v231) that only inline asm
// touches: left to the register allocator, the 100-way switch inside the constraint loop turns into 16 phi copies
// per case (measured: as many v_mov as v_and).
#define ACC_CLOBBER "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
                    "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231"
#define STR2(x) #x
#define STR(x) STR2(x)
template <int S, int J> struct AccReg;
#define DEF_ACC(S, J, N) template <> struct AccReg<S, J> { \
    static __device__ __forceinline__ void and_with(uint32_t x) { asm volatile("v_and_b32 v" STR(N) ", %0, v" STR(N) :: "v"(x) : ACC_CLOBBER); } \
    static __device__ __forceinline__ void set(uint32_t x) { asm volatile("v_mov_b32 v" STR(N) ", %0" :: "v"(x) : ACC_CLOBBER); } };
#define DEF_ROW(J, N0, N1) DEF_ACC(0, J, N0) DEF_ACC(1, J, N1)
DEF_ROW(0, 200, 216) DEF_ROW(1, 201, 217) DEF_ROW(2, 202, 218) DEF_ROW(3, 203, 219) DEF_ROW(4, 204, 220) DEF_ROW(5, 205, 221)
DEF_ROW(6, 206, 222) DEF_ROW(7, 207, 223) DEF_ROW(8, 208, 224) DEF_ROW(9, 209, 225) DEF_ROW(10, 210, 226) DEF_ROW(11, 211, 227)
DEF_ROW(12, 212, 228) DEF_ROW(13, 213, 229) DEF_ROW(14, 214, 230) DEF_ROW(15, 215, 231)
template <int J> struct HitReg;
#define DEF_HIT(J, N0, N1) template <> struct HitReg<J> { static __device__ __forceinline__ uint32_t get() { \
    uint32_t h; asm volatile("v_or_b32 %0, v" STR(N0) ", v" STR(N1) : "=v"(h) :: ACC_CLOBBER); return h; } };
DEF_HIT(0, 200, 216) DEF_HIT(1, 201, 217) DEF_HIT(2, 202, 218) DEF_HIT(3, 203, 219) DEF_HIT(4, 204, 220) DEF_HIT(5, 205, 221)
DEF_HIT(6, 206, 222) DEF_HIT(7, 207, 223) DEF_HIT(8, 208, 224) DEF_HIT(9, 209, 225) DEF_HIT(10, 210, 226) DEF_HIT(11, 211, 227)
DEF_HIT(12, 212, 228) DEF_HIT(13, 213, 229) DEF_HIT(14, 214, 230) DEF_HIT(15, 215, 231)

template <int S, int P, int D>
__device__ __forceinline__ void apply(const uint32_t (&E)[4][EXT]) {        // one asm block: no hazard nops in between
#define A16(B) "v_and_b32 v" STR(B) ", %0, v" STR(B) "\n\t"
    if constexpr (S == 0)
        asm volatile("v_and_b32 v200, %0, v200\n\tv_and_b32 v201, %1, v201\n\tv_and_b32 v202, %2, v202\n\tv_and_b32 v203, %3, v203\n\t"
                     "v_and_b32 v204, %4, v204\n\tv_and_b32 v205, %5, v205\n\tv_and_b32 v206, %6, v206\n\tv_and_b32 v207, %7, v207\n\t"
                     "v_and_b32 v208, %8, v208\n\tv_and_b32 v209, %9, v209\n\tv_and_b32 v210, %10, v210\n\tv_and_b32 v211, %11, v211\n\t"
                     "v_and_b32 v212, %12, v212\n\tv_and_b32 v213, %13, v213\n\tv_and_b32 v214, %14, v214\n\tv_and_b32 v215, %15, v215"
                     :: "v"(E[P][0 + D + DMAX]), "v"(E[P][1 + D + DMAX]), "v"(E[P][2 + D + DMAX]), "v"(E[P][3 + D + DMAX]),
                        "v"(E[P][4 + D + DMAX]), "v"(E[P][5 + D + DMAX]), "v"(E[P][6 + D + DMAX]), "v"(E[P][7 + D + DMAX]),
                        "v"(E[P][8 + D + DMAX]), "v"(E[P][9 + D + DMAX]), "v"(E[P][10 + D + DMAX]), "v"(E[P][11 + D + DMAX]),
                        "v"(E[P][12 + D + DMAX]), "v"(E[P][13 + D + DMAX]), "v"(E[P][14 + D + DMAX]), "v"(E[P][15 + D + DMAX])
                     : ACC_CLOBBER);
    else
        asm volatile("v_and_b32 v216, %0, v216\n\tv_and_b32 v217, %1, v217\n\tv_and_b32 v218, %2, v218\n\tv_and_b32 v219, %3, v219\n\t"
                     "v_and_b32 v220, %4, v220\n\tv_and_b32 v221, %5, v221\n\tv_and_b32 v222, %6, v222\n\tv_and_b32 v223, %7, v223\n\t"
                     "v_and_b32 v224, %8, v224\n\tv_and_b32 v225, %9, v225\n\tv_and_b32 v226, %10, v226\n\tv_and_b32 v227, %11, v227\n\t"
                     "v_and_b32 v228, %12, v228\n\tv_and_b32 v229, %13, v229\n\tv_and_b32 v230, %14, v230\n\tv_and_b32 v231, %15, v231"
                     :: "v"(E[P][0 + D + DMAX]), "v"(E[P][1 + D + DMAX]), "v"(E[P][2 + D + DMAX]), "v"(E[P][3 + D + DMAX]),
                        "v"(E[P][4 + D + DMAX]), "v"(E[P][5 + D + DMAX]), "v"(E[P][6 + D + DMAX]), "v"(E[P][7 + D + DMAX]),
                        "v"(E[P][8 + D + DMAX]), "v"(E[P][9 + D + DMAX]), "v"(E[P][10 + D + DMAX]), "v"(E[P][11 + D + DMAX]),
                        "v"(E[P][12 + D + DMAX]), "v"(E[P][13 + D + DMAX]), "v"(E[P][14 + D + DMAX]), "v"(E[P][15 + D + DMAX])
                     : ACC_CLOBBER);
}
template <int S, int J = 0>
__device__ __forceinline__ void init_acc(const uint32_t (&E)[4][EXT]) {
    if constexpr (J < R) {
        AccReg<S, J>::set(E[S][J + DMAX]);
        init_acc<S, J + 1>(E);
    }
}
template <int J = 0>
__device__ __forceinline__ void count(const uint32_t (&M)[R], const uint32_t (&U)[R], uint32_t &nm, uint32_t &nu) {
    if constexpr (J < R) {
        const uint32_t h = HitReg<J>::get();
        nm += __popc(h & M[J]);
        nu += __popc(h & U[J]);
        count<J + 1>(M, U, nm, nu);
    }
}

template <int S, int P>
__device__ __forceinline__ void dispatch_d(int d, const uint32_t (&E)[4][EXT]) {
    // binary decision tree over d in [-DMAX, DMAX], leaves are 16-AND blocks
    switch (d) {
#define CASE(x) case x: apply<S, P, x>(E); break;
        CASE(-12) CASE(-11) CASE(-10) CASE(-9) CASE(-8) CASE(-7) CASE(-6) CASE(-5) CASE(-4) CASE(-3) CASE(-2) CASE(-1) CASE(0)
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
#undef CASE
    }
}

template <int S>
__device__ __forceinline__ void strand(cu8p prog, int c, int NC, const uint32_t (&E)[4][EXT]) {
    init_acc<S>(E);                                                   // modified-base plane folded into the init
    for (int q = 0; q < NC; ++q) {
        const int code = __builtin_amdgcn_readfirstlane(prog[(c * 2 + S) * NC + q]);
        const int p = code / (2 * DMAX + 1), d = code % (2 * DMAX + 1) - DMAX;
        switch (p) {
            case 0: dispatch_d<S, 0>(d, E); break;
            case 1: dispatch_d<S, 1>(d, E); break;
            case 2: dispatch_d<S, 2>(d, E); break;
            default: dispatch_d<S, 3>(d, E); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Variant 2: VGPR index mode.  The extended planes are pinned to v40..v199 (plane p at v[40 + 40 p + k]) and a
// constraint (p, d) is ONE straight-line block of 16 ANDs whose src0 register number is offset by M0 = 40 p + d + DMAX
// (s_set_gpr_idx_on, SRC0 relative): no switch, no branch.
// ---------------------------------------------------------------------------------------------------
#define V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define E_CLOBBER V10(4), V10(5), V10(6), V10(7), V10(8), V10(9), V10(10), V10(11), V10(12), V10(13), V10(14), V10(15), V10(16), V10(17), V10(18), V10(19)
#define ALL_CLOBBER E_CLOBBER, ACC_CLOBBER

template <int N> struct PinE;     // write E value into physical register v[40 + N]
template <int N>
__device__ __forceinline__ void pin(uint32_t x);
#define PIN1(N) template <> __device__ __forceinline__ void pin<N>(uint32_t x) { asm volatile("v_mov_b32 v" STR(N) ", %0" :: "v"(x) : ALL_CLOBBER); }
#define PIN10(a) PIN1(a##0) PIN1(a##1) PIN1(a##2) PIN1(a##3) PIN1(a##4) PIN1(a##5) PIN1(a##6) PIN1(a##7) PIN1(a##8) PIN1(a##9)
PIN10(4) PIN10(5) PIN10(6) PIN10(7) PIN10(8) PIN10(9) PIN10(10) PIN10(11) PIN10(12) PIN10(13) PIN10(14) PIN10(15) PIN10(16) PIN10(17) PIN10(18) PIN10(19)

template <int P, int K = 0>
__device__ __forceinline__ void pin_plane(const uint32_t (&w)[R], const uint32_t (&nx)[R], const uint32_t (&pv)[R]) {
    if constexpr (K < EXT) {
        uint32_t x;
        if constexpr (K < DMAX) x = (w[R - DMAX + K] << 1) | (pv[R - DMAX + K] >> 31);          // one bit row down, carry from the previous block
        else if constexpr (K < DMAX + R) x = w[K - DMAX];
        else x = (w[K - DMAX - R] >> 1) | (nx[K - DMAX - R] << 31);                               // one bit row up
        pin<40 + 40 * P + K>(x);
        pin_plane<P, K + 1>(w, nx, pv);
    }
}

template <int S>
__device__ __forceinline__ void and_indexed(uint32_t idx) {
    if constexpr (S == 0)
        asm volatile("s_set_gpr_idx_on %0, 0x1\n\ts_nop 1\n\t"
                     "v_and_b32 v200, v40, v200\n\tv_and_b32 v201, v41, v201\n\tv_and_b32 v202, v42, v202\n\tv_and_b32 v203, v43, v203\n\t"
                     "v_and_b32 v204, v44, v204\n\tv_and_b32 v205, v45, v205\n\tv_and_b32 v206, v46, v206\n\tv_and_b32 v207, v47, v207\n\t"
                     "v_and_b32 v208, v48, v208\n\tv_and_b32 v209, v49, v209\n\tv_and_b32 v210, v50, v210\n\tv_and_b32 v211, v51, v211\n\t"
                     "v_and_b32 v212, v52, v212\n\tv_and_b32 v213, v53, v213\n\tv_and_b32 v214, v54, v214\n\tv_and_b32 v215, v55, v215\n\t"
                     "s_set_gpr_idx_off" :: "s"(idx) : ALL_CLOBBER, "m0");
    else
        asm volatile("s_set_gpr_idx_on %0, 0x1\n\ts_nop 1\n\t"
                     "v_and_b32 v216, v40, v216\n\tv_and_b32 v217, v41, v217\n\tv_and_b32 v218, v42, v218\n\tv_and_b32 v219, v43, v219\n\t"
                     "v_and_b32 v220, v44, v220\n\tv_and_b32 v221, v45, v221\n\tv_and_b32 v222, v46, v222\n\tv_and_b32 v223, v47, v223\n\t"
                     "v_and_b32 v224, v48, v224\n\tv_and_b32 v225, v49, v225\n\tv_and_b32 v226, v50, v226\n\tv_and_b32 v227, v51, v227\n\t"
                     "v_and_b32 v228, v52, v228\n\tv_and_b32 v229, v53, v229\n\tv_and_b32 v230, v54, v230\n\tv_and_b32 v231, v55, v231\n\t"
                     "s_set_gpr_idx_off" :: "s"(idx) : ALL_CLOBBER, "m0");
}
template <int S>
__device__ __forceinline__ void init_indexed(uint32_t idx) {          // acc = plane registers at idx (the modified-base plane at offset 0)
    if constexpr (S == 0)
        asm volatile("s_set_gpr_idx_on %0, 0x1\n\ts_nop 1\n\t"
                     "v_mov_b32 v200, v40\n\tv_mov_b32 v201, v41\n\tv_mov_b32 v202, v42\n\tv_mov_b32 v203, v43\n\tv_mov_b32 v204, v44\n\tv_mov_b32 v205, v45\n\t"
                     "v_mov_b32 v206, v46\n\tv_mov_b32 v207, v47\n\tv_mov_b32 v208, v48\n\tv_mov_b32 v209, v49\n\tv_mov_b32 v210, v50\n\tv_mov_b32 v211, v51\n\t"
                     "v_mov_b32 v212, v52\n\tv_mov_b32 v213, v53\n\tv_mov_b32 v214, v54\n\tv_mov_b32 v215, v55\n\ts_set_gpr_idx_off" :: "s"(idx) : ALL_CLOBBER, "m0");
    else
        asm volatile("s_set_gpr_idx_on %0, 0x1\n\ts_nop 1\n\t"
                     "v_mov_b32 v216, v40\n\tv_mov_b32 v217, v41\n\tv_mov_b32 v218, v42\n\tv_mov_b32 v219, v43\n\tv_mov_b32 v220, v44\n\tv_mov_b32 v221, v45\n\t"
                     "v_mov_b32 v222, v46\n\tv_mov_b32 v223, v47\n\tv_mov_b32 v224, v48\n\tv_mov_b32 v225, v49\n\tv_mov_b32 v226, v50\n\tv_mov_b32 v227, v51\n\t"
                     "v_mov_b32 v228, v52\n\tv_mov_b32 v229, v53\n\tv_mov_b32 v230, v54\n\tv_mov_b32 v231, v55\n\ts_set_gpr_idx_off" :: "s"(idx) : ALL_CLOBBER, "m0");
}

typedef const unsigned long long __attribute__((address_space(4))) *cu64p;
template <int S>
__device__ __forceinline__ void strand_indexed(cu64p prog8, int c, int NC) {
    // the (candidate, strand) program = up to 8 register-index bytes, fetched with ONE scalar load
    unsigned long long codes = prog8[c * 2 + S];
    init_indexed<S>(40 * S + DMAX);
    for (int q = 0; q < NC; ++q) {
        and_indexed<S>((uint32_t)(codes & 0xFF));
        codes >>= 8;
    }
}

__global__ __launch_bounds__(256, 2) void proto_indexed(const uint32_t *__restrict__ planes, size_t words,
                                                        const unsigned long long *__restrict__ prog_, int B, int NC, int tiles_per_block,
                                                        unsigned long long *out) {
    cu64p prog = (cu64p)prog_;
    const int lane_global = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long total = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t w0 = ((size_t)t * gridDim.x * blockDim.x + lane_global) * R;
        {
            uint32_t w[R], nx[R], pv[R];
#define LOAD_PLANE(P) \
            for (int j = 0; j < R; ++j) { w[j] = planes[P * words + w0 + j]; } \
            for (int j = 0; j < R; ++j) { nx[j] = __shfl_down(w[j], 1); pv[j] = __shfl_up(w[j], 1); } \
            pin_plane<P>(w, nx, pv);
            _Pragma("unroll") LOAD_PLANE(0)
            _Pragma("unroll") LOAD_PLANE(1)
            _Pragma("unroll") LOAD_PLANE(2)
            _Pragma("unroll") LOAD_PLANE(3)
        }
        uint32_t M[R], U[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { M[j] = planes[4 * words + w0 + j]; U[j] = planes[5 * words + w0 + j]; }
        for (int c = 0; c < B; ++c) {
            strand_indexed<0>(prog, c, NC);
            strand_indexed<1>(prog, c, NC);
            uint32_t nm = 0, nu = 0;
            count(M, U, nm, nu);
            total += ((unsigned long long)nm << 32) + nu + c;
        }
    }
    if (total == 0x123456789ull) out[0] = total;
    atomicAdd(out + 1, total & 0xFFFF);
}

__global__ __launch_bounds__(256, 2) void proto(const uint32_t *__restrict__ planes /*[6][words]*/, size_t words,
                                                const uint8_t *__restrict__ prog_ /*[B][2][NC] codes*/, int B, int NC,
                                                int tiles_per_block, unsigned long long *out) {
    cu8p prog = (cu8p)prog_;
    const int lane_global = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long total = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t w0 = ((size_t)t * gridDim.x * blockDim.x + lane_global) * R;
        uint32_t E[4][EXT], M[R], U[R];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int j = 0; j < R; ++j) E[p][j + DMAX] = planes[p * words + w0 + j];
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {                    // wrap registers: one bit row up / down
                E[p][R + DMAX + k] = E[p][DMAX + k] >> 1;
                E[p][k] = E[p][R + k] << 1;
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) { M[j] = planes[4 * words + w0 + j]; U[j] = planes[5 * words + w0 + j]; }
        for (int c = 0; c < B; ++c) {
            strand<0>(prog, c, NC, E);
            strand<1>(prog, c, NC, E);
            uint32_t nm = 0, nu = 0;
            count(M, U, nm, nu);
            total += ((unsigned long long)nm << 32) + nu + c;
        }
    }
    if (total == 0x123456789ull) out[0] = total;        // keep the work alive
    atomicAdd(out + 1, total & 0xFFFF);
}

// Variant 3: the whole (candidate, strand) program in ONE asm block — index mode stays on, s_set_gpr_idx_idx per
// constraint, scalar loop in asm; popcounts in four independent chains.
template <int S>
__device__ __forceinline__ void strand_asm(cu64p prog8, int c, int NC) {
    const unsigned long long codes = prog8[c * 2 + S];
    uint32_t lo = (uint32_t)codes, hi = (uint32_t)(codes >> 32);
    uint32_t n = NC, init_idx = 40 * S + DMAX, tmp;
#define ANDS(B0) \
    "v_and_b32 v" STR(B0) "0, v40, v" STR(B0) "0\n\tv_and_b32 v" STR(B0) "1, v41, v" STR(B0) "1\n\t"
    if constexpr (S == 0)
        asm volatile(
            "s_set_gpr_idx_on %4, 0x1\n\ts_nop 1\n\t"
            "v_mov_b32 v200, v40\n\tv_mov_b32 v201, v41\n\tv_mov_b32 v202, v42\n\tv_mov_b32 v203, v43\n\tv_mov_b32 v204, v44\n\tv_mov_b32 v205, v45\n\t"
            "v_mov_b32 v206, v46\n\tv_mov_b32 v207, v47\n\tv_mov_b32 v208, v48\n\tv_mov_b32 v209, v49\n\tv_mov_b32 v210, v50\n\tv_mov_b32 v211, v51\n\t"
            "v_mov_b32 v212, v52\n\tv_mov_b32 v213, v53\n\tv_mov_b32 v214, v54\n\tv_mov_b32 v215, v55\n\t"
            "s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 2f\n\t"
            "1:\n\t"
            "s_and_b32 %2, %0, 0xff\n\ts_set_gpr_idx_idx %2\n\ts_lshr_b32 %0, %0, 8\n\ts_lshl_b32 %2, %3, 24\n\ts_or_b32 %0, %0, %2\n\ts_lshr_b32 %3, %3, 8\n\ts_sub_u32 %1, %1, 1\n\t"
            "v_and_b32 v200, v40, v200\n\tv_and_b32 v201, v41, v201\n\tv_and_b32 v202, v42, v202\n\tv_and_b32 v203, v43, v203\n\t"
            "v_and_b32 v204, v44, v204\n\tv_and_b32 v205, v45, v205\n\tv_and_b32 v206, v46, v206\n\tv_and_b32 v207, v47, v207\n\t"
            "v_and_b32 v208, v48, v208\n\tv_and_b32 v209, v49, v209\n\tv_and_b32 v210, v50, v210\n\tv_and_b32 v211, v51, v211\n\t"
            "v_and_b32 v212, v52, v212\n\tv_and_b32 v213, v53, v213\n\tv_and_b32 v214, v54, v214\n\tv_and_b32 v215, v55, v215\n\t"
            "s_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1b\n\t"
            "2:\n\ts_set_gpr_idx_off"
            : "+s"(lo), "+s"(n), "=&s"(tmp), "+s"(hi) : "s"(init_idx) : ALL_CLOBBER, "m0", "scc");
    else
        asm volatile(
            "s_set_gpr_idx_on %4, 0x1\n\ts_nop 1\n\t"
            "v_mov_b32 v216, v40\n\tv_mov_b32 v217, v41\n\tv_mov_b32 v218, v42\n\tv_mov_b32 v219, v43\n\tv_mov_b32 v220, v44\n\tv_mov_b32 v221, v45\n\t"
            "v_mov_b32 v222, v46\n\tv_mov_b32 v223, v47\n\tv_mov_b32 v224, v48\n\tv_mov_b32 v225, v49\n\tv_mov_b32 v226, v50\n\tv_mov_b32 v227, v51\n\t"
            "v_mov_b32 v228, v52\n\tv_mov_b32 v229, v53\n\tv_mov_b32 v230, v54\n\tv_mov_b32 v231, v55\n\t"
            "s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 2f\n\t"
            "1:\n\t"
            "s_and_b32 %2, %0, 0xff\n\ts_set_gpr_idx_idx %2\n\ts_lshr_b32 %0, %0, 8\n\ts_lshl_b32 %2, %3, 24\n\ts_or_b32 %0, %0, %2\n\ts_lshr_b32 %3, %3, 8\n\ts_sub_u32 %1, %1, 1\n\t"
            "v_and_b32 v216, v40, v216\n\tv_and_b32 v217, v41, v217\n\tv_and_b32 v218, v42, v218\n\tv_and_b32 v219, v43, v219\n\t"
            "v_and_b32 v220, v44, v220\n\tv_and_b32 v221, v45, v221\n\tv_and_b32 v222, v46, v222\n\tv_and_b32 v223, v47, v223\n\t"
            "v_and_b32 v224, v48, v224\n\tv_and_b32 v225, v49, v225\n\tv_and_b32 v226, v50, v226\n\tv_and_b32 v227, v51, v227\n\t"
            "v_and_b32 v228, v52, v228\n\tv_and_b32 v229, v53, v229\n\tv_and_b32 v230, v54, v230\n\tv_and_b32 v231, v55, v231\n\t"
            "s_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1b\n\t"
            "2:\n\ts_set_gpr_idx_off"
            : "+s"(lo), "+s"(n), "=&s"(tmp), "+s"(hi) : "s"(init_idx) : ALL_CLOBBER, "m0", "scc");
}

template <int J = 0>
__device__ __forceinline__ void count4(const uint32_t (&M)[R], const uint32_t (&U)[R], uint32_t (&nm)[4], uint32_t (&nu)[4]) {
    if constexpr (J < R) {
        const uint32_t h = HitReg<J>::get();
        nm[J & 3] += __popc(h & M[J]);
        nu[J & 3] += __popc(h & U[J]);
        count4<J + 1>(M, U, nm, nu);
    }
}

__global__ __launch_bounds__(256, 2) void proto_asm(const uint32_t *__restrict__ planes, size_t words,
                                                    const unsigned long long *__restrict__ prog_, int B, int NC, int tiles_per_block,
                                                    unsigned long long *out) {
    cu64p prog = (cu64p)prog_;
    const int lane_global = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long total = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t w0 = ((size_t)t * gridDim.x * blockDim.x + lane_global) * R;
        {
            uint32_t w[R], nx[R], pv[R];
            _Pragma("unroll") LOAD_PLANE(0)
            _Pragma("unroll") LOAD_PLANE(1)
            _Pragma("unroll") LOAD_PLANE(2)
            _Pragma("unroll") LOAD_PLANE(3)
        }
        uint32_t M[R], U[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { M[j] = planes[4 * words + w0 + j]; U[j] = planes[5 * words + w0 + j]; }
        for (int c = 0; c < B; ++c) {
            strand_asm<0>(prog, c, NC);
            strand_asm<1>(prog, c, NC);
            uint32_t nm[4] = {0, 0, 0, 0}, nu[4] = {0, 0, 0, 0};
            count4(M, U, nm, nu);
            total += ((unsigned long long)(nm[0] + nm[1] + nm[2] + nm[3]) << 32) + nu[0] + nu[1] + nu[2] + nu[3] + c;
        }
    }
    if (total == 0x123456789ull) out[0] = total;
    atomicAdd(out + 1, total & 0xFFFF);
}

// Variant 4: statically addressed 16-AND blocks behind an INDIRECT JUMP.  One block per register offset (160 per
// strand, 68 bytes each, generated with assembler .rept), dispatch = s_mul / s_add / s_setpc_b64, every block ends with
// a direct s_branch back to the loop head.  The ANDs keep the plain 2.6-cycle rate.
template <int S>
__device__ __forceinline__ void strand_jump(cu64p prog8, int c, int NC) {
    const unsigned long long codes = prog8[c * 2 + S];
    uint32_t lo = (uint32_t)codes, hi = (uint32_t)(codes >> 32), n = NC, tmp;
    if constexpr (S == 0)
        asm volatile(
            "v_mov_b32 v200, v52\n\tv_mov_b32 v201, v53\n\tv_mov_b32 v202, v54\n\tv_mov_b32 v203, v55\n\tv_mov_b32 v204, v56\n\tv_mov_b32 v205, v57\n\t"
            "v_mov_b32 v206, v58\n\tv_mov_b32 v207, v59\n\tv_mov_b32 v208, v60\n\tv_mov_b32 v209, v61\n\tv_mov_b32 v210, v62\n\tv_mov_b32 v211, v63\n\t"
            "v_mov_b32 v212, v64\n\tv_mov_b32 v213, v65\n\tv_mov_b32 v214, v66\n\tv_mov_b32 v215, v67\n\t"
            "s_getpc_b64 s[96:97]\n"
            "10:\n\t"
            "s_add_u32 s96, s96, 30f-10b\n\ts_addc_u32 s97, s97, 0\n"
            "11:\n\t"
            "s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 40f\n\t"
            "s_and_b32 %2, %0, 0xff\n\ts_lshr_b32 %0, %0, 8\n\ts_lshl_b32 s98, %3, 24\n\ts_or_b32 %0, %0, s98\n\ts_lshr_b32 %3, %3, 8\n\ts_sub_u32 %1, %1, 1\n\t"
            "s_mul_i32 %2, %2, 68\n\ts_add_u32 s98, s96, %2\n\ts_addc_u32 s99, s97, 0\n\ts_setpc_b64 s[98:99]\n"
            "30:\n\t"
            ".set nm_k, 0\n\t.rept 160\n\t"
            "v_and_b32 v200, v[40+nm_k], v200\n\tv_and_b32 v201, v[41+nm_k], v201\n\tv_and_b32 v202, v[42+nm_k], v202\n\tv_and_b32 v203, v[43+nm_k], v203\n\t"
            "v_and_b32 v204, v[44+nm_k], v204\n\tv_and_b32 v205, v[45+nm_k], v205\n\tv_and_b32 v206, v[46+nm_k], v206\n\tv_and_b32 v207, v[47+nm_k], v207\n\t"
            "v_and_b32 v208, v[48+nm_k], v208\n\tv_and_b32 v209, v[49+nm_k], v209\n\tv_and_b32 v210, v[50+nm_k], v210\n\tv_and_b32 v211, v[51+nm_k], v211\n\t"
            "v_and_b32 v212, v[52+nm_k], v212\n\tv_and_b32 v213, v[53+nm_k], v213\n\tv_and_b32 v214, v[54+nm_k], v214\n\tv_and_b32 v215, v[55+nm_k], v215\n\t"
            "s_branch 11b\n\t.set nm_k, nm_k+1\n\t.endr\n"
            "40:\n\t"
            : "+s"(lo), "+s"(n), "=&s"(tmp), "+s"(hi) :: ALL_CLOBBER, "s96", "s97", "s98", "s99", "scc");
    else
        asm volatile(
            "v_mov_b32 v216, v92\n\tv_mov_b32 v217, v93\n\tv_mov_b32 v218, v94\n\tv_mov_b32 v219, v95\n\tv_mov_b32 v220, v96\n\tv_mov_b32 v221, v97\n\t"
            "v_mov_b32 v222, v98\n\tv_mov_b32 v223, v99\n\tv_mov_b32 v224, v100\n\tv_mov_b32 v225, v101\n\tv_mov_b32 v226, v102\n\tv_mov_b32 v227, v103\n\t"
            "v_mov_b32 v228, v104\n\tv_mov_b32 v229, v105\n\tv_mov_b32 v230, v106\n\tv_mov_b32 v231, v107\n\t"
            "s_getpc_b64 s[96:97]\n"
            "10:\n\t"
            "s_add_u32 s96, s96, 30f-10b\n\ts_addc_u32 s97, s97, 0\n"
            "11:\n\t"
            "s_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 40f\n\t"
            "s_and_b32 %2, %0, 0xff\n\ts_lshr_b32 %0, %0, 8\n\ts_lshl_b32 s98, %3, 24\n\ts_or_b32 %0, %0, s98\n\ts_lshr_b32 %3, %3, 8\n\ts_sub_u32 %1, %1, 1\n\t"
            "s_mul_i32 %2, %2, 68\n\ts_add_u32 s98, s96, %2\n\ts_addc_u32 s99, s97, 0\n\ts_setpc_b64 s[98:99]\n"
            "30:\n\t"
            ".set nm_k, 0\n\t.rept 160\n\t"
            "v_and_b32 v216, v[40+nm_k], v216\n\tv_and_b32 v217, v[41+nm_k], v217\n\tv_and_b32 v218, v[42+nm_k], v218\n\tv_and_b32 v219, v[43+nm_k], v219\n\t"
            "v_and_b32 v220, v[44+nm_k], v220\n\tv_and_b32 v221, v[45+nm_k], v221\n\tv_and_b32 v222, v[46+nm_k], v222\n\tv_and_b32 v223, v[47+nm_k], v223\n\t"
            "v_and_b32 v224, v[48+nm_k], v224\n\tv_and_b32 v225, v[49+nm_k], v225\n\tv_and_b32 v226, v[50+nm_k], v226\n\tv_and_b32 v227, v[51+nm_k], v227\n\t"
            "v_and_b32 v228, v[52+nm_k], v228\n\tv_and_b32 v229, v[53+nm_k], v229\n\tv_and_b32 v230, v[54+nm_k], v230\n\tv_and_b32 v231, v[55+nm_k], v231\n\t"
            "s_branch 11b\n\t.set nm_k, nm_k+1\n\t.endr\n"
            "40:\n\t"
            : "+s"(lo), "+s"(n), "=&s"(tmp), "+s"(hi) :: ALL_CLOBBER, "s96", "s97", "s98", "s99", "scc");
}

__global__ __launch_bounds__(256, 2) void proto_jump(const uint32_t *__restrict__ planes, size_t words,
                                                     const unsigned long long *__restrict__ prog_, int B, int NC, int tiles_per_block,
                                                     unsigned long long *out) {
    cu64p prog = (cu64p)prog_;
    const int lane_global = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long total = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t w0 = ((size_t)t * gridDim.x * blockDim.x + lane_global) * R;
        {
            uint32_t w[R], nx[R], pv[R];
            _Pragma("unroll") LOAD_PLANE(0)
            _Pragma("unroll") LOAD_PLANE(1)
            _Pragma("unroll") LOAD_PLANE(2)
            _Pragma("unroll") LOAD_PLANE(3)
        }
        uint32_t M[R], U[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { M[j] = planes[4 * words + w0 + j]; U[j] = planes[5 * words + w0 + j]; }
        for (int c = 0; c < B; ++c) {
            strand_jump<0>(prog, c, NC);
            strand_jump<1>(prog, c, NC);
            uint32_t nm[4] = {0, 0, 0, 0}, nu[4] = {0, 0, 0, 0};
            count4(M, U, nm, nu);
            total += ((unsigned long long)(nm[0] + nm[1] + nm[2] + nm[3]) << 32) + nu[0] + nu[1] + nu[2] + nu[3] + c;
        }
    }
    if (total == 0x123456789ull) out[0] = total;
    atomicAdd(out + 1, total & 0xFFFF);
}

int main(int argc, char **argv) {
    const int B = 20, NC = argc > 1 ? atoi(argv[1]) : 5, blocks = 2048, tiles = 8;
    const size_t words = (size_t)blocks * 256 * R * tiles;
    uint32_t *d_planes; unsigned long long *d_out; uint8_t *d_prog;
    (void)hipMalloc(&d_planes, words * 6 * 4); (void)hipMalloc(&d_out, 16); (void)hipMemset(d_out, 0, 16);
    std::vector<uint32_t> h(1 << 20);
    for (auto &x : h) x = (uint32_t)rand() * 2654435761u;
    for (size_t o = 0; o < words * 6; o += h.size()) (void)hipMemcpy(d_planes + o, h.data(), std::min(h.size(), words * 6 - o) * 4, hipMemcpyHostToDevice);
    std::vector<uint8_t> prog(B * 2 * NC);
    for (auto &c : prog) c = (uint8_t)((rand() % 4) * (2 * DMAX + 1) + (rand() % (2 * DMAX + 1)));
    (void)hipMalloc(&d_prog, prog.size()); (void)hipMemcpy(d_prog, prog.data(), prog.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    proto<<<blocks, 256>>>(d_planes, words, d_prog, B, NC, 1, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    proto<<<blocks, 256>>>(d_planes, words, d_prog, B, NC, tiles, d_out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double positions = (double)words * 32, owned = positions * 28.0 / 32.0;      // 2 bit rows of overlap each side
    printf("[switch dispatch] ");
    printf("transposed-tile proxy: %d candidates x %d constraints/strand, %.3f ms for %.3g positions (%.3g owned)\n", B, NC, ms, positions, owned);
    printf("  motif-sites/s (2 strands, owned positions): %.3e   [current kernel on cfg5: 6.9e13 kernel-only]\n", owned * 2 * B / (ms * 1e-3));
    printf("  bytes/s if planes were 0.5 B/bp per group: %.2f TB/s algorithmic\n", owned * 0.5 / (ms * 1e-3) / 1e12);
    std::vector<unsigned long long> prog8(B * 2, 0);
    for (int i = 0; i < B * 2; ++i)
        for (int q = 0; q < NC && q < 8; ++q) {
            const int code = prog[i * NC + q], pl = code / (2 * DMAX + 1), k = code % (2 * DMAX + 1);
            prog8[i] |= (unsigned long long)(40 * pl + k) << (8 * q);
        }
    unsigned long long *d_prog8; (void)hipMalloc(&d_prog8, prog8.size() * 8); (void)hipMemcpy(d_prog8, prog8.data(), prog8.size() * 8, hipMemcpyHostToDevice);
    proto_indexed<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, 1, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    proto_indexed<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, tiles, d_out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("[VGPR index mode] %d candidates x %d constraints/strand, %.3f ms: %.3e motif-sites/s\n", B, NC, ms, owned * 2 * B / (ms * 1e-3));
    proto_asm<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, 1, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    proto_asm<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, tiles, d_out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("[index mode, asm strand loop] %d candidates x %d constraints/strand, %.3f ms: %.3e motif-sites/s\n", B, NC, ms, owned * 2 * B / (ms * 1e-3));
    proto_jump<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, 1, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    proto_jump<<<blocks, 256>>>(d_planes, words, d_prog8, B, NC, tiles, d_out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("[indirect jump to static blocks] %d candidates x %d constraints/strand, %.3f ms: %.3e motif-sites/s\n", B, NC, ms, owned * 2 * B / (ms * 1e-3));
    printf("  hip error state: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
