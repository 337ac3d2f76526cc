// Device-side building blocks of the scan shared by the translation units that evaluate motifs against the resident
// planes (nmscan.hip: the scoring kernels and nm_hit_positions; nmmeth.hip: the per-contig read-methylation table), plus
// the host-side compiler of one motif into its constraint program.  Internal: not part of the C ABI.
#pragma once
#include "nmscan_internal.h"

namespace nmdetail {

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// Read-only tables are addressed through the constant address space so that wave-uniform reads become scalar
// loads (s_load_dwordx16 straight into SGPRs) instead of vector loads + v_readfirstlane.
typedef const uint32_t __attribute__((address_space(4))) *cu32p;
typedef const uint8_t __attribute__((address_space(4))) *cu8p;

// Kernel variant: GN / GP halo words left / right (narrow 1 / 1: offsets in [-32, 31]; wide 2 / 2), COMPACT state planes
// {M, U} or the four per-strand planes, NS mod-type slots fused per workgroup, LIT = every constraint of the batch is a
// literal (only the four is-X planes are built — every batch of the greedy search is: children are single letters,
// find_motifs_bin.py:1005-1018), CF = the constraints common to all candidates of a (slot, bin) group are evaluated
// once per tile (sibling children of one expansion differ in ONE position, :1116-1135).
// Tried and dropped for the light variants (same-device A/B, tools/gpu_ab.sh, 1 Gbp greedy round): a register double
// buffer of the next chunk's raw words (120 VGPRs, 4 waves: -18 %); touching the next chunk's lines with one dword per
// lane and plane so that they are on their way into L2 (-25 %; with nontemporal touches -45 %); 8 waves per SIMD by
// launch bounds (3 spills; -4 % with two children per group, +6 % with four); segments of 8 / 32 / 64 chunks (all
// slower than 16); handing segments out through a device-wide atomic queue instead of one workgroup per segment
// (-60 %: device-scope atomics are served at the memory side on this multi-die part and serialise).  A 10 Gbp run
// takes 10.0 x the 1 Gbp time: there is no partial-last-round tail worth chasing at that size.  An EVEN static split —
// exactly the resident number of workgroups, each walking an equal, candidate-weighted share of the active bins' chunks
// across bin boundaries (parity green) — was 28 % slower at 1 Gbp and 20 % slower on a 125 Mbp shard: equal shares do
// not finish together, and nothing rebalances them; the hardware dispatcher's one-workgroup-per-segment order does.
#ifndef NM_LIT_WAVES
#define NM_LIT_WAVES 4      // minimum waves per SIMD the literal-only variants are compiled for
#endif

template <int GN_, int GP_, bool COMPACT_, int NS_, bool LIT_, bool CF_, bool PC_ = false>
struct Variant {
    static constexpr int GN = GN_, GP = GP_, NS = NS_;
    static constexpr bool COMPACT = COMPACT_, LIT = LIT_, CF = CF_, PC = PC_;   // PC: counters keyed by (candidate, contig)
    static constexpr int NW = T_WORDS + GN + GP;
    static constexpr int NP = LIT ? 4 : 8;                 // planes per tile
    static constexpr int NST = COMPACT ? 2 : 4;            // state planes per slot
    static constexpr int PDW = (GN + GP) * NP;             // dwords of one strand's program
};

// The raw words one lane holds of one chunk: T main words of H / L (/ V) plus the halo, and the state words of the
// NS slots.  All loads of a chunk are issued back to back, nothing is waited for here.
template <class K>
struct RawChunk {
    uint32_t h[K::NW], l[K::NW], v[K::NW];
    uint32_t s[K::NS][K::NST][T_WORDS];
    bool need_v;                                           // wave-uniform

    __device__ __forceinline__ void load(const Planes &seq, const StatePlanes (&stp)[K::NS], uint32_t chunk, int lane) {
        constexpr int GN = K::GN, GP = K::GP;
        need_v = ((cu8p)seq.needs_v)[chunk] != 0;          // scalar load, issued first
        const size_t base = (size_t)chunk * CHUNK_WORDS + (size_t)lane * T_WORDS;
#pragma unroll
        for (int j = 0; j < K::NS; ++j) {
            const uint32_t *src[4] = {K::COMPACT ? stp[j].M : stp[j].MP, K::COMPACT ? stp[j].U : stp[j].UP, stp[j].MM, stp[j].UM};
#pragma unroll
            for (int i = 0; i < K::NST; ++i) {
                const uint4 q = *reinterpret_cast<const uint4 *>(src[i] + base);
                s[j][i][0] = q.x; s[j][i][1] = q.y; s[j][i][2] = q.z; s[j][i][3] = q.w;
            }
        }
        const uint4 h4 = *reinterpret_cast<const uint4 *>(seq.H + base);
        const uint4 l4 = *reinterpret_cast<const uint4 *>(seq.L + base);
        h[GN + 0] = h4.x; h[GN + 1] = h4.y; h[GN + 2] = h4.z; h[GN + 3] = h4.w;
        l[GN + 0] = l4.x; l[GN + 1] = l4.y; l[GN + 2] = l4.z; l[GN + 3] = l4.w;
#pragma unroll
        for (int j = 0; j < GN; ++j) { h[j] = seq.H[base - GN + j]; l[j] = seq.L[base - GN + j]; }
#pragma unroll
        for (int j = 0; j < GP; ++j) { h[GN + T_WORDS + j] = seq.H[base + T_WORDS + j]; l[GN + T_WORDS + j] = seq.L[base + T_WORDS + j]; }
        if (need_v) {                                      // wave-uniform
            const uint4 v4 = *reinterpret_cast<const uint4 *>(seq.V + base);
            v[GN + 0] = v4.x; v[GN + 1] = v4.y; v[GN + 2] = v4.z; v[GN + 3] = v4.w;
#pragma unroll
            for (int j = 0; j < GN; ++j) v[j] = seq.V[base - GN + j];
#pragma unroll
            for (int j = 0; j < GP; ++j) v[GN + T_WORDS + j] = seq.V[base + T_WORDS + j];
        } else {
#pragma unroll
            for (int j = 0; j < K::NW; ++j) v[j] = 0xFFFFFFFFu;
        }
    }
};

// Derived planes of the tile: is-A/C/G/T and (unless LIT) valid-not-A/C/G/T, NW words each.
template <class K>
struct Tile {
    uint32_t w[K::NP][K::NW];

    __device__ __forceinline__ void expand(const RawChunk<K> &r) {
#pragma unroll
        for (int j = 0; j < K::NW; ++j) {
            const uint32_t hh = r.h[j], ll = r.l[j], vv = r.v[j];
            w[0][j] = vv & ~hh & ~ll;          // A = 00
            w[1][j] = vv & ~hh & ll;           // C = 01
            w[2][j] = vv & hh & ll;            // G = 11
            w[3][j] = vv & hh & ~ll;           // T = 10
            if (!K::LIT) {
                w[4][j] = vv & (hh | ll);          // valid, not A
                w[5][j] = vv & (hh | ~ll);         // valid, not C
                w[6][j] = vv & ~(hh & ll);         // valid, not G
                w[7][j] = vv & (~hh | ll);         // valid, not T
            }
        }
    }
};

// acc[t] &= plane p at offset d (d = 32 g + r) for every constraint bit of one strand's program
// (prog[g * NP + p], bit r; g counts from the leftmost word-group the variant reads).  Control flow is scalar and
// wave-uniform (s_ff1 over the SGPR masks), register indices are static.
template <class K>
__device__ __forceinline__ void eval_masks(const uint32_t (&m)[K::PDW], const Tile<K> &tile, uint32_t (&acc)[T_WORDS]) {
#pragma unroll
    for (int g = 0; g < K::GN + K::GP; ++g) {            // word pair (t + g, t + g + 1)
#pragma unroll
        for (int p = 0; p < K::NP; ++p) {
            uint32_t mm = m[g * K::NP + p];
            while (mm) {
                const uint32_t r = __builtin_ctz(mm);
                mm &= mm - 1;
#pragma unroll
                for (int t = 0; t < T_WORDS; ++t) acc[t] &= alignbit(tile.w[p][t + g + 1], tile.w[p][t + g], r);
            }
        }
    }
}

template <class K>
__device__ __forceinline__ void eval_strand(cu32p prog, const Tile<K> &tile, uint32_t (&acc)[T_WORDS]) {
    uint32_t m[K::PDW];
#pragma unroll
    for (int i = 0; i < K::PDW; ++i) m[i] = prog[i];
    eval_masks<K>(m, tile, acc);
}

// ---- host: one stripped motif -> per-strand constraint masks ---------------------------------------------------------
inline uint32_t comp_mask(uint32_t m) { return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3); }

// Compile one stripped motif into the per-strand constraint masks of the general path (accumulator starting from all
// ones: the modified position's own constraint is part of the program).  Layout: prog[strand * 48 + gi * 8 + plane],
// gi = floor(d / 32) + 3 (six word-groups: offsets in [-96, 95]), bit r = d mod 32.  *reach_class: 0 when every offset lies
// in [-32, 31], 1 in [-64, 63], 2 beyond (the Variant<G, G, ...> with G = class + 1 reads exactly those groups).
constexpr int PROG6_DW = 96;
inline int compile_program(const uint8_t *masks, uint32_t len, uint32_t modpos, uint32_t *prog, int *reach_class) {
    if (len == 0 || len > NM_MAX_MOTIF_LEN) return fail(NM_ERANGE, "motif length %u outside 1..%d", len, NM_MAX_MOTIF_LEN);
    if (modpos >= len) return fail(NM_EINVAL, "mod_position %u outside motif of length %u", modpos, len);
    memset(prog, 0, PROG6_DW * sizeof(uint32_t));
    bool any = false;
    *reach_class = 0;
    for (uint32_t j = 0; j < len; ++j) {
        const uint32_t m = masks[j] & 15u;
        if (m == 0) return fail(NM_EINVAL, "empty base set at motif position %u", j);
        if (m == 15u) continue;
        any = true;
        for (int strand = 0; strand < 2; ++strand) {
            const int d = strand == 0 ? (int)j - (int)modpos : (int)modpos - (int)j;
            const uint32_t set = strand == 0 ? m : comp_mask(m);
            const int gi = (d >> 5) + 3;                            // arithmetic shift = floor
            const uint32_t r = (uint32_t)d & 31u;
            if (gi < 0 || gi > 5) return fail(NM_ERANGE, "offset %d from the modified base is outside [-96, 95]", d);
            *reach_class = std::max(*reach_class, gi == 0 || gi == 5 ? 2 : gi == 1 || gi == 4 ? 1 : 0);
            uint32_t *row = prog + strand * 48 + gi * 8;
            if (__builtin_popcount(set) == 1) {
                row[__builtin_ctz(set)] |= 1u << r;                 // literal: plane of that base
            } else {
                uint32_t missing = (~set) & 15u;                    // 3-set: one "valid and not x"; 2-set: two of them
                while (missing) {
                    row[4 + __builtin_ctz(missing)] |= 1u << r;
                    missing &= missing - 1;
                }
            }
        }
    }
    if (!any) return fail(NM_EINVAL, "motif has no specified position");
    return NM_OK;
}

// The word-groups Variant<G, G, ...> reads (3 - G .. 2 + G) of a six-group program, as [strand][2 G][8 planes].
inline void slice_program(const uint32_t *full, int G, uint32_t *out) {
    for (int strand = 0; strand < 2; ++strand)
        memcpy(out + strand * 2 * G * 8, full + strand * 48 + (3 - G) * 8, (size_t)2 * G * 8 * sizeof(uint32_t));
}

// ---- candidate records -> constraint programs (nmscan.hip: compile_kernel / common_kernel / compile_common_kernel; nmwindows.hip: the
// speculative children of the search compile their own)
// Compile the staged candidates into constraint programs ON THE DEVICE: one thread per candidate.  A literal is one
// constraint on an is-X plane, a 3-set one on a valid-not-X plane, a 2-set two of those; the reverse strand takes
// the complemented set at the negated offset (motif.py:260-266).  Program layout: [strand][word-group][plane] with
// the word-groups the launched variant reads (narrow: groups 1..2, wide: 0..3) and its planes (np = 4: literal-only
// batch, is-X planes; np = 8); bit r of a word = offset 32 g + r.
// fold_modpos: the modified position's own constraint is left out (compact batches start the accumulator from the
// canonical plane instead).
// prog_slot: where in `programs` the program goes (default: slot k)
__device__ __forceinline__ void compile_one(uint32_t k, const CandRec *__restrict__ rec, const uint8_t *__restrict__ masks,
                                            uint32_t *__restrict__ programs, int wide, int np, int fold_modpos, uint32_t prog_slot = 0xFFFFFFFFu) {
    const int groups = 2 + 2 * wide, g0 = 1 - wide;        // wide: 0 narrow (word-groups 1..2), 1 wide (0..3), 2 extra wide (-1..4)
    const int pdw = 2 * groups * np;
    uint32_t *prog = programs + (size_t)(prog_slot == 0xFFFFFFFFu ? k : prog_slot) * pdw;
    for (int i = 0; i < pdw; ++i) prog[i] = 0;
    const CandRec c = rec[k];
    const uint8_t *m = masks + c.mask_off;
    for (int j = 0; j < c.len; ++j) {
        const uint32_t set_f = m[j] & 15u;
        if (set_f == 15u || (fold_modpos && j == c.modpos)) continue;
        for (int strand = 0; strand < 2; ++strand) {
            const int d = strand == 0 ? j - (int)c.modpos : (int)c.modpos - j;
            const uint32_t set = strand == 0 ? set_f
                                             : (((set_f & 1) << 3) | ((set_f & 2) << 1) | ((set_f & 4) >> 1) | ((set_f & 8) >> 3));
            const int g = (d >> 5) + 2 - g0;
            const uint32_t bit = 1u << ((uint32_t)d & 31u);
            uint32_t *row = prog + (strand * groups + g) * np;
            if (__popc(set) == 1) {
                row[__ffs(set) - 1] |= bit;
            } else {                                    // never reached with np = 4: the host checked the batch
                uint32_t missing = (~set) & 15u;
                while (missing) {
                    row[4 + __ffs(missing) - 1] |= bit;
                    missing &= missing - 1;
                }
            }
        }
    }
}

// Light batches (a round of the greedy search): the constraints shared by ALL candidates of a (slot, bin) group — the
// parent of sibling children (find_motifs_bin.py:1116-1135), the motif under its parents in a pruning round
// (:1408-1432) — become the group's COMMON program (index n_prog + group), evaluated once per tile; the candidates keep
// the rest.  One thread per group; groups of 1 or of more than max_group candidates are left alone (range.z = ~0).
// programs: the candidates' programs (global memory, or the LDS copy compile_common_kernel works on); commons: where the
// common program of entry g goes (global memory, program index n_prog + g)
__device__ __forceinline__ void common_one(uint32_t g, uint4 *__restrict__ range, uint32_t *programs, uint32_t *commons, uint32_t pdw,
                                           uint32_t n_prog, uint32_t max_group) {
    uint4 r = range[g];
    r.z = 0xFFFFFFFFu;
    r.w = 0;
    if (r.y >= 2 && r.y <= max_group) {
        uint32_t *common = commons + (size_t)g * pdw;
        uint32_t any = 0;
        for (uint32_t i = 0; i < pdw; ++i) {
            uint32_t c = programs[(size_t)r.x * pdw + i];
            for (uint32_t k = 1; k < r.y; ++k) c &= programs[(size_t)(r.x + k) * pdw + i];
            common[i] = c;
            any |= c;
        }
        if (any) {
            // siblings: every candidate keeps at most ONE constraint per strand -> its program shrinks to two
            // descriptors (mask index << 5 | r; index = dwords per strand when nothing is left) and range.w = 1
            bool single = true;
            const uint32_t sdw = pdw / 2;
            for (uint32_t k = 0; k < r.y; ++k) {
                uint32_t *prog = programs + (size_t)(r.x + k) * pdw;
                for (uint32_t i = 0; i < pdw; ++i) prog[i] &= ~common[i];
                for (uint32_t st = 0; st < 2; ++st) {
                    uint32_t bits = 0;
                    for (uint32_t i = 0; i < sdw; ++i) bits += __popc(prog[st * sdw + i]);
                    if (bits > 1) single = false;
                }
            }
            if (single) {
                for (uint32_t k = 0; k < r.y; ++k) {
                    uint32_t *prog = programs + (size_t)(r.x + k) * pdw;
                    uint32_t desc[2];
                    for (uint32_t st = 0; st < 2; ++st) {
                        desc[st] = sdw << 5;
                        for (uint32_t i = 0; i < sdw; ++i)
                            if (prog[st * sdw + i]) desc[st] = (i << 5) | (uint32_t)(__ffs(prog[st * sdw + i]) - 1);
                    }
                    prog[0] = desc[0];
                    prog[1] = desc[1];
                }
                r.w = 1;
            }
            r.z = n_prog + g;
        }
    }
    range[g] = r;
}


}  // namespace nmdetail
