// libnmscan — MI355X (gfx950 / CDNA4) motif scan + methylation count engine.  C ABI: include/nmscan.h.
//
// Design (DESIGN.md has the long form):
//   * Assembly resident in HBM as three bit planes over one padded coordinate space: H, L (the two bits of
//     the base code, "bit-sliced 2-bit") and V (1 = A/C/G/T inside a contig).  Contigs are grouped by bin and
//     start on 8192-bp chunk boundaries with >= 96 invalid positions (GAP_BP) after each one, so a match can never
//     straddle two contigs and a chunk never straddles two bins.
//   * Per modification type two state planes M / U (methylated, unmethylated; the strand of a row is implied by
//     the base under it: '+' rows sit on the canonical base, '-' rows on its complement) — 0.5 byte per bp per
//     scoring step in total with H and L.  Four per-strand planes (MP, UP, MM, UM) back the general case
//     (motifs whose modified position is not the canonical literal).
//   * Scoring kernel: one wavefront owns one chunk (64 lanes x 4 consecutive 32-bit words = 8192 bp) held in
//     VGPRs as eight derived planes (is-A/C/G/T and valid-not-A/C/G/T) with one halo word each side.  A
//     candidate motif is a set of wave-uniform constraints "plane p must be set at offset d from the modified
//     base"; each costs one v_alignbit + one v_and per word, driven from SGPR bit masks (s_ff1 loop), no LDS
//     traffic and no divergent control flow.  Forward and reverse-complement sites are disjoint (canonical vs
//     complement base), so they are OR-ed and counted with two popcounts against M and U.
//   * Per-lane counts go to LDS with ds_add, are reduced once per workgroup segment and leave the CU as one
//     64-bit atomic per counter.
#include "nmscan_device.h"

using namespace nmdetail;

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------------
struct ScoreArgs {
    Planes seq;
    StatePlanes st[NM_MAX_MOD_SLOTS];
    const uint4 *segments;      // {first chunk, n chunks, bin, 0}
    uint32_t n_segments;
    uint32_t split_log2;        // every segment is cut into 1 << split_log2 pieces
    uint32_t pieces_per_run, j_big, fine_log2;   // per run of pieces: the first j_big go whole, the rest in 1 << fine_log2 parts
    uint32_t n_bins;
    const uint4 *cand_range;    // [active_slot_index][bin] -> {begin, count, common program or ~0, -} into programs
    const uint32_t *programs;   // [n_prog][2 * (GN + GP) * 8], sorted by (slot, bin)
    const uint32_t *orig_index; // [n_cand] sorted -> caller order
    unsigned long long *out;    // [n_cand][2]; per-contig mode: [rows][2], row = row_base[candidate] + rank of the contig in its bin
    const uint32_t *chunk_rank; // per chunk: rank of its contig within its bin (per-contig mode)
    const uint64_t *row_base;   // [n_prog] sorted order (per-contig mode)
    uint32_t active_slot[NM_MAX_MOD_SLOTS];
    uint32_t slot_is_c[NM_MAX_MOD_SLOTS];   // canonical base of the slot is C (else A)
};

// Pack: one workgroup per chunk, 64 positions per wave per step, wave ballot builds the plane words.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii,
                                                   const uint64_t *__restrict__ contig_src,
                                                   const uint64_t *__restrict__ contig_len,
                                                   const uint32_t *__restrict__ chunk_contig,
                                                   const uint32_t *__restrict__ chunk_first,
                                                   uint32_t *__restrict__ H, uint32_t *__restrict__ L,
                                                   uint32_t *__restrict__ V, unsigned long long *__restrict__ other) {
    const uint32_t chunk = blockIdx.x;
    const uint32_t contig = chunk_contig[chunk];
    if (contig == 0xFFFFFFFFu) return;                               // pad chunk, planes already zero
    const uint64_t beg = contig_src[contig], len = contig_len[contig];     // (src: where the contig's bytes start in `ascii`)
    const uint64_t local0 = (uint64_t)(chunk - chunk_first[contig]) * CHUNK_BP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = wave; it < CHUNK_BP / 64; it += 4) {
        const uint64_t p = local0 + (uint64_t)it * 64 + lane;
        uint8_t c = 0;
        if (p < len) c = ascii[beg + p] & 0xDF;                      // upper-case (seq.py:55)
        const bool valid = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
        const unsigned long long bv = __ballot(valid);
        const unsigned long long bh = __ballot(valid && (c & 4));    // A=00 C=01 G=11 T=10 as (H,L) = bits 2,1
        const unsigned long long bl = __ballot(valid && (c & 2));
        const unsigned long long bo = __ballot(p < len && !valid && c != 'N');   // IUPAC ambiguity codes etc.
        if (bo && lane == 0) atomicAdd(other, (unsigned long long)__popcll(bo));
        if (lane < 2) {
            const size_t w = (size_t)chunk * CHUNK_WORDS + (size_t)it * 2 + lane;
            H[w] = (uint32_t)(bh >> (32 * lane));
            L[w] = (uint32_t)(bl >> (32 * lane));
            V[w] = (uint32_t)(bv >> (32 * lane));
        }
    }
}

// needs_v[c] = 0 iff chunk c is entirely valid and so are the three words either side of it (the widest halo a variant reads).
__global__ void needs_v_kernel(const uint32_t *__restrict__ V, uint8_t *__restrict__ needs_v, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x;
    __shared__ uint32_t all_and;
    if (threadIdx.x == 0) all_and = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t v = V[(size_t)c * CHUNK_WORDS + threadIdx.x];
    if (threadIdx.x < 3) {
        if (c > 0) v &= V[(size_t)c * CHUNK_WORDS - 1 - threadIdx.x];
        else v = 0;
        if (c + 1 < n_chunks) v &= V[(size_t)(c + 1) * CHUNK_WORDS + threadIdx.x];
        else v = 0;
    }
    if (v != 0xFFFFFFFFu) atomicAnd(&all_and, v);
    __syncthreads();
    if (threadIdx.x == 0) needs_v[c] = all_and != 0xFFFFFFFFu;
}

// State build: one thread per pileup row; classification is the reference's float64 compare.
__global__ void state_kernel(uint64_t n_rows, const uint32_t *__restrict__ contig_id,
                             const uint32_t *__restrict__ position, const uint8_t *__restrict__ strand,
                             const double *__restrict__ frac, double low, double high,
                             const uint32_t *__restrict__ contig_chunk, const uint64_t *__restrict__ contig_len,
                             uint32_t n_contigs, uint32_t can_h, uint32_t can_l,
                             const uint32_t *__restrict__ H, const uint32_t *__restrict__ L,
                             const uint32_t *__restrict__ V, uint32_t *M, uint32_t *U, uint32_t *MP, uint32_t *UP,
                             uint32_t *MM, uint32_t *UM, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t cid = contig_id[i];
    const uint32_t pos = position[i];
    if (cid >= n_contigs || pos >= contig_len[cid]) { atomicOr(err, 1u); return; }
    const double f = frac[i];
    const bool meth = f >= high, non = f <= low;
    if (!meth && !non) return;
    const uint64_t g = (uint64_t)contig_chunk[cid] * CHUNK_BP + pos;
    const size_t w = g >> 5;
    const uint32_t bit = 1u << (g & 31);
    const bool plus = strand[i] == '+';
    if (!plus && strand[i] != '-') { atomicOr(err, 2u); return; }
    uint32_t *gen = plus ? (meth ? MP : UP) : (meth ? MM : UM);
    const uint32_t old = atomicOr(gen + w, bit);
    if (old & bit) atomicOr(err, 4u);                                // duplicate (contig, position, strand)
    // compact planes: keep the row only when the base under it is the one its strand implies
    const uint32_t h = (H[w] & bit) != 0, l = (L[w] & bit) != 0, v = (V[w] & bit) != 0;
    // complement in the (H,L) code: A(00)<->T(10), C(01)<->G(11) — H flips, L stays
    const uint32_t want_h = plus ? can_h : (can_h ^ 1u);
    if (v && h == want_h && l == can_l) atomicOr((meth ? M : U) + w, bit);
}

// acc = base & (the ONE constraint `desc` = (mask index << 5) | r of a strand): the residual of a sibling child once the
// parent's constraints are in `base`.  The mask index selects registers, so it is dispatched through a switch.
template <class K, int I = 0>
__device__ __forceinline__ void apply_single(uint32_t idx, uint32_t r, const Tile<K> &tile, const uint32_t (&base)[T_WORDS],
                                             uint32_t (&acc)[T_WORDS]) {
    if constexpr (I < K::PDW) {
        if (idx == I) {
            constexpr int g = I / K::NP, p = I % K::NP;
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) acc[t] = base[t] & alignbit(tile.w[p][t + g + 1], tile.w[p][t + g], r);
        } else {
            apply_single<K, I + 1>(idx, r, tile, base, acc);
        }
    } else {
#pragma unroll
        for (int t = 0; t < T_WORDS; ++t) acc[t] = base[t];      // idx == PDW: no constraint left on this strand
    }
}

// One slot's candidates [k0, k0 + nb) against the tile this wave holds: match masks, site counts, per-lane counts into
// LDS rows k * NS + lds_row0 (lds_row0 = the slot's index in the workgroup).  CAN: canonical base of the slot, 0 = A (reverse-strand sites sit on T), 1 = C (reverse on G).
// common != ~0u: program index of the constraints shared by all nb candidates (their own programs hold the rest).
template <class K, int CAN>
__device__ __forceinline__ void score_candidates(const ScoreArgs &a, const Tile<K> &tile, const uint32_t (&sw)[K::NST][T_WORDS],
                                                 uint32_t k0, uint32_t nb, uint32_t common, bool siblings, uint32_t *lds_acc,
                                                 uint32_t lds_row0, int lane, uint32_t contig_rank = 0) {
    constexpr int PF = CAN == 0 ? 0 : 1;   // plane of the canonical base: A or C
    constexpr int PR = CAN == 0 ? 3 : 2;   // plane of its complement:     T or G
    uint32_t basef[T_WORDS], baser[T_WORDS];
#pragma unroll
    for (int t = 0; t < T_WORDS; ++t) {
        // compact batches: every candidate has the canonical literal at its modified position, the compiler leaves
        // that constraint out of the program and it becomes the accumulator's initial value
        basef[t] = K::COMPACT ? tile.w[PF][t + K::GN] : 0xFFFFFFFFu;
        baser[t] = K::COMPACT ? tile.w[PR][t + K::GN] : 0xFFFFFFFFu;
    }
    if (K::CF && common != 0xFFFFFFFFu) {                                    // wave-uniform
        cu32p prog = (cu32p)(a.programs + (size_t)common * (2 * K::PDW));
        eval_strand<K>(prog, tile, basef);
        eval_strand<K>(prog + K::PDW, tile, baser);
    }
    // Scalar loads of the masks are software-pipelined at strand granularity: the reverse masks of candidate k are
    // requested before its forward strand is evaluated, the forward masks of candidate k + 1 before its reverse strand —
    // two sets of SGPRs like before, but a load's latency hides behind ~40 vector instructions instead of standing in
    // front of every strand.
    uint32_t mf[K::PDW], mr[K::PDW];
    if (!(K::CF && siblings)) {
        cu32p prog = (cu32p)(a.programs + (size_t)k0 * (2 * K::PDW));
#pragma unroll
        for (int i = 0; i < K::PDW; ++i) mf[i] = prog[i];
    }
    for (uint32_t k = 0; k < nb; ++k) {
        cu32p prog = (cu32p)(a.programs + (size_t)(k0 + k) * (2 * K::PDW));
        uint32_t accf[T_WORDS], accr[T_WORDS];
        if (K::CF && siblings) {                                             // wave-uniform: one constraint per strand left
            const uint32_t df = prog[0], dr = prog[1];
            apply_single<K>(df >> 5, df & 31u, tile, basef, accf);
            apply_single<K>(dr >> 5, dr & 31u, tile, baser, accr);
        } else {
#pragma unroll
            for (int i = 0; i < K::PDW; ++i) mr[i] = prog[K::PDW + i];
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) accf[t] = basef[t];
            eval_masks<K>(mf, tile, accf);
            cu32p next = (cu32p)(a.programs + (size_t)(k0 + min(k + 1, nb - 1)) * (2 * K::PDW));
#pragma unroll
            for (int i = 0; i < K::PDW; ++i) mf[i] = next[i];
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) accr[t] = baser[t];
            eval_masks<K>(mr, tile, accr);
        }
        uint32_t n_mod = 0, n_non = 0;
#pragma unroll
        for (int t = 0; t < T_WORDS; ++t) {
            if (K::COMPACT) {
                const uint32_t sites = accf[t] | accr[t];       // forward sites sit on the canonical base, reverse on its complement
                n_mod += __popc(sites & sw[0][t]);
                n_non += __popc(sites & sw[1][t]);
            } else {
                n_mod += __popc(accf[t] & sw[0][t]) + __popc(accr[t] & sw[K::COMPACT ? 0 : 2][t]);
                n_non += __popc(accf[t] & sw[1][t]) + __popc(accr[t] & sw[K::COMPACT ? 1 : 3][t]);
            }
        }
        if (K::PC) {
            // per-contig counters (motif_model_contig per contig, find_motifs_bin.py:1285-1331): a chunk lies inside ONE
            // contig, so the wave's sum goes straight to that (candidate, contig) row
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                n_mod += __shfl_xor(n_mod, o);
                n_non += __shfl_xor(n_non, o);
            }
            if (lane == 0 && (n_mod | n_non)) {
                unsigned long long *row = a.out + (a.row_base[k0 + k] + contig_rank) * 2;
                if (n_mod) atomicAdd(row, (unsigned long long)n_mod);
                if (n_non) atomicAdd(row + 1, (unsigned long long)n_non);
            }
            continue;
        }
        atomicAdd(&lds_acc[((k * K::NS + lds_row0) * 2 + 0) * 64 + lane], n_mod);
        atomicAdd(&lds_acc[((k * K::NS + lds_row0) * 2 + 1) * 64 + lane], n_non);
    }
}

// With NS > 1 a tile's sequence planes are loaded and expanded once and serve the candidates of all NS slots (each slot
// brings its own state planes); with NS = 1 the slot comes from blockIdx.y.  A pass handles up to BMAX / NS candidates
// per slot; LDS rows are [candidate k][slot j].
// One PIECE of work: the chunks [sg.x, sg.x + sg.y) of bin sg.z, all candidates of the workgroup's slot column(s), counters
// accumulated in LDS and flushed to the count table at the end.  Called once per workgroup by score_kernel (piece = a
// segment or a part of one).
template <class K>
__device__ __forceinline__ void score_piece(const ScoreArgs &a, const uint4 sg, const StatePlanes (&stp)[K::NS], const bool (&is_c)[K::NS],
                                            uint32_t *lds_acc, const int lane, const uint32_t wave) {
    constexpr int NS = K::NS;
    constexpr uint32_t H = BMAX / NS;
    // the wave's first chunk is requested before anything else of the segment is looked at (candidate ranges, LDS
    // clearing, the barrier): a workgroup lives for four chunks per wave, its start-up chain would otherwise sit
    // in front of every fourth memory round trip
    constexpr bool EARLY = K::CF && K::LIT;          // (the 8-plane light tiles would drop from 5 to 4 waves per SIMD)
    RawChunk<K> first;
    if (EARLY && wave < sg.y) first.load(a.seq, stp, sg.x + wave, lane);
    uint4 range[NS];                                // {first program, candidates, common program or ~0, siblings}
    uint32_t most = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const uint32_t slot_i = NS == 1 ? blockIdx.y : (uint32_t)j;
        range[j] = a.cand_range[(size_t)slot_i * a.n_bins + sg.z];
        range[j].x = __builtin_amdgcn_readfirstlane(range[j].x);
        range[j].y = __builtin_amdgcn_readfirstlane(range[j].y);
        range[j].z = __builtin_amdgcn_readfirstlane(range[j].z);
        range[j].w = __builtin_amdgcn_readfirstlane(range[j].w);
        most = max(most, range[j].y);
    }
    if (most == 0) return;

    // one chunk of one pass: tile, then every slot's candidates of this pass
    auto score_chunk = [&](const RawChunk<K> &cur, uint32_t pass0, uint32_t chunk) {
        Tile<K> tile;
        tile.expand(cur);
        const uint32_t rank = K::PC ? ((cu32p)a.chunk_rank)[chunk] : 0u;
        // one slot at a time with the slot index a compile-time constant: left to `#pragma unroll`, the optimizer gave up on
        // the NS = 2 non-literal bodies ("loop not unrolled") and indexed cur.s[j] / range[j] / is_c[j] through private
        // memory — 192-288 bytes of scratch per lane in five variants (round-3 review)
        auto one_slot = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (range[j].y <= pass0) return;                         // wave-uniform
            const uint32_t nbj = min(H, range[j].y - pass0);
            if (K::COMPACT && is_c[j])
                score_candidates<K, 1>(a, tile, cur.s[j], range[j].x + pass0, nbj, range[j].z, range[j].w != 0, lds_acc, j, lane, rank);
            else
                score_candidates<K, 0>(a, tile, cur.s[j], range[j].x + pass0, nbj, range[j].z, range[j].w != 0, lds_acc, j, lane, rank);
        };
        one_slot(std::integral_constant<int, 0>{});
        if constexpr (NS > 1) one_slot(std::integral_constant<int, 1>{});
        static_assert(NS <= 2, "slot fusion is written for one or two slots");
    };
    auto clear_rows = [&](uint32_t rows_hi) {
        if (K::PC) return;
        for (uint32_t i = threadIdx.x; i < rows_hi * 128; i += 256) lds_acc[i] = 0;
        __syncthreads();
    };
    // 4 threads per counter, 16 lane-slots each, then a 4-lane butterfly; one 64-bit atomic per counter
    auto flush_rows = [&](uint32_t rows_hi, uint32_t pass0) {
        if (K::PC) return;
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < rows_hi * 8; idx += 256) {
            const uint32_t i = idx >> 2, q = idx & 3;               // i = counter row: (k * NS + j) * 2 + which
            const uint32_t j = (i >> 1) % NS, k = (i >> 1) / NS;
            uint32_t s = 0;
#pragma unroll 4
            for (int jj = 0; jj < 16; ++jj) s += lds_acc[i * 64 + q * 16 + jj];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (q == 0 && s) {
                uint4 rj = range[0];
#pragma unroll
                for (int t = 1; t < NS; ++t)
                    if (j == (uint32_t)t) rj = range[t];
                if (pass0 + k < rj.y) {
                    const uint32_t orig = a.orig_index[rj.x + pass0 + k];
                    atomicAdd(a.out + (size_t)orig * 2 + (i & 1), (unsigned long long)s);
                }
            }
        }
        __syncthreads();
    };
    // LDS rows in use in a pass: row = k * NS + j for candidate k of slot j
    uint32_t pass0 = 0;
    if (EARLY) {                                                     // pass 0 with the chunk already under way
        const uint32_t rows_hi = NS * min(H, most);
        clear_rows(rows_hi);
        if (wave < sg.y) score_chunk(first, 0, sg.x + wave);
        for (uint32_t ck = wave + 4; ck < sg.y; ck += 4) {
            RawChunk<K> cur;
            cur.load(a.seq, stp, sg.x + ck, lane);
            score_chunk(cur, 0, sg.x + ck);
        }
        flush_rows(rows_hi, 0);
        pass0 = H;
    }
    for (; pass0 < most; pass0 += H) {
        const uint32_t rows_hi = NS * min(H, most - pass0);
        clear_rows(rows_hi);
        for (uint32_t ck = wave; ck < sg.y; ck += 4) {
            RawChunk<K> cur;
            cur.load(a.seq, stp, sg.x + ck, lane);
            score_chunk(cur, pass0, sg.x + ck);
        }
        flush_rows(rows_hi, pass0);
    }
}

#define NM_SCORE_BOUNDS __launch_bounds__(256, (K::GN + K::GP > 2 ? 2 : (K::LIT ? NM_LIT_WAVES : 4)))

// per-slot facts are read from the kernel arguments once per workgroup
#define NM_SLOT_SETUP                                                                                               \
    const int lane = threadIdx.x & 63;                                                                              \
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); /* provably uniform: chunk indices stay scalar */ \
    StatePlanes stp[K::NS];                                                                                         \
    bool is_c[K::NS];                                                                                               \
    _Pragma("unroll") for (int j = 0; j < K::NS; ++j) {                                                             \
        const uint32_t slot = a.active_slot[K::NS == 1 ? blockIdx.y : (uint32_t)j];                                 \
        stp[j] = a.st[slot];                                                                                        \
        is_c[j] = a.slot_is_c[slot] != 0;                                                                           \
    }

template <class K>
__global__ NM_SCORE_BOUNDS void score_kernel(ScoreArgs a) {
    __shared__ uint32_t lds_acc[BMAX * 2 * 64];
    // XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), give every XCD a contiguous run
    // of segments so candidate programs and counters of one bin stay in one L2.
    // (a light batch streams: there the remap costs 3 % of the read rate, tools/stream_pattern.hip)
    const uint32_t lanes_x = K::CF ? 1u : 8u;                  // runs of pieces: one per XCD, or a single one
    const uint32_t x = K::CF ? 0u : blockIdx.x % 8, j = K::CF ? blockIdx.x : blockIdx.x / 8;
    // Pieces: a segment (16 chunks), or a half / quarter of one for assemblies that fill the device for less than two
    // rounds of workgroups (split_log2).  The workgroups dispatched LAST (j >= j_big in every run) take pieces cut
    // finer still (fine_log2): the last, partly filled round of workgroups then lasts a quarter as long.
    uint32_t piece, sub = 0, n_sub = 1;
    if (j < a.j_big) piece = x * a.pieces_per_run + j;
    else {
        const uint32_t k = j - a.j_big;
        piece = x * a.pieces_per_run + a.j_big + (k >> a.fine_log2);
        sub = k & ((1u << a.fine_log2) - 1);
        n_sub = 1u << a.fine_log2;
        if (a.j_big + (k >> a.fine_log2) >= a.pieces_per_run) return;
    }
    (void)lanes_x;
    const uint32_t seg = piece >> a.split_log2;
    if (seg >= a.n_segments) return;
    uint4 sg = a.segments[seg];
    sg.x = __builtin_amdgcn_readfirstlane(sg.x);   // everything below is wave-uniform: keep it in SGPRs
    sg.y = __builtin_amdgcn_readfirstlane(sg.y);
    sg.z = __builtin_amdgcn_readfirstlane(sg.z);
    if (a.split_log2) {
        const uint32_t len = (sg.y + (1u << a.split_log2) - 1) >> a.split_log2;
        const uint32_t at = (piece & ((1u << a.split_log2) - 1)) * len;
        if (at >= sg.y) return;
        sg.x += at;
        sg.y = min(len, sg.y - at);
    }
    if (n_sub > 1) {
        const uint32_t len = (sg.y + n_sub - 1) / n_sub;
        const uint32_t at = sub * len;
        if (at >= sg.y) return;
        sg.x += at;
        sg.y = min(len, sg.y - at);
    }
    NM_SLOT_SETUP
    score_piece<K>(a, sg, stp, is_c, lds_acc, lane, wave);
}

__global__ void compile_kernel(uint32_t n_prog, const CandRec *__restrict__ rec, const uint8_t *__restrict__ masks,
                               uint32_t *__restrict__ programs, int wide, int np, int fold_modpos) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_prog) compile_one(k, rec, masks, programs, wide, np, fold_modpos);
}

__global__ void common_kernel(uint32_t n_entries, uint4 *__restrict__ range, uint32_t *__restrict__ programs, uint32_t pdw,
                              uint32_t n_prog, uint32_t max_group) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_entries) common_one(g, range, programs, programs + (size_t)n_prog * pdw, pdw, n_prog, max_group);
}

// Both steps in ONE launch for the small light batches of the search's long tail (a few dozen candidates, where a round
// is a chain of launch latencies): a single workgroup compiles, synchronises, and factors the groups.
__global__ __launch_bounds__(1024) void compile_common_kernel(uint32_t n_prog, const CandRec *__restrict__ rec,
                                                               const uint8_t *__restrict__ masks, uint32_t *__restrict__ programs, int wide,
                                                               int np, int fold_modpos, uint32_t n_entries, uint4 *__restrict__ range,
                                                               uint32_t pdw, uint32_t max_group, uint32_t lds_words) {
    // the candidates' programs are built and factored in LDS (a program is a chain of read-modify-writes: one dependent
    // round trip each in global memory) and leave in one coalesced copy; a batch too large for that works in place
    extern __shared__ uint32_t lds_programs[];
    uint32_t *work = lds_words ? lds_programs : programs;
    for (uint32_t k = threadIdx.x; k < n_prog; k += blockDim.x) compile_one(k, rec, masks, work, wide, np, fold_modpos);
    __threadfence_block();
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < n_entries; g += blockDim.x) common_one(g, range, work, programs + (size_t)n_prog * pdw, pdw, n_prog, max_group);
    if (lds_words) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_prog * pdw; i += blockDim.x) programs[i] = lds_programs[i];
    }
}

// ---- candidates that reach further than 95 positions from the modified base (a search frame above 191: the reference takes any
// --search_frame_size, find_motifs_bin.py:110-130): a plain, slow kernel — speed is irrelevant there.  One workgroup per (chunk of the
// candidate's bin, candidate), one thread per 32-position word; per specified motif position the plane words are FETCHED at the
// shifted address (no register tile), a site survives only while every constrained position lies inside its own contig
// (utils.py:44-67: a regex match cannot leave the string; beyond 96 positions the padding between contigs no longer guarantees it).
struct WideArgs {
    Planes seq;
    StatePlanes st[NM_MAX_MOD_SLOTS];
    const uint32_t *cand_bin, *cand_mask_off;
    const uint16_t *cand_len, *cand_modpos;
    const uint8_t *cand_slot, *masks;
    const uint32_t *bin_chunk0, *bin_nchunks, *chunk_contig, *contig_chunk;
    const uint64_t *contig_len;
    unsigned long long *out;
    long long n_words;
};

__device__ __forceinline__ uint32_t plane_bits_at(const uint32_t *__restrict__ pl, long long n_words, long long g) {      // the 32 bits of a plane from position g on
    const long long i = g >> 5;                                           // (arithmetic shift: floor)
    const uint32_t sh = (uint32_t)(g & 31);
    const uint32_t lo = i >= 0 && i < n_words ? pl[i] : 0u;
    const uint32_t hi = sh && i + 1 >= 0 && i + 1 < n_words ? pl[i + 1] : 0u;
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
}

__global__ __launch_bounds__(256) void score_wide_kernel(WideArgs a) {
    __shared__ unsigned int part[2][4];
    const uint32_t k = blockIdx.y, bin = a.cand_bin[k];
    if (blockIdx.x >= a.bin_nchunks[bin]) return;
    const uint32_t chunk = a.bin_chunk0[bin] + blockIdx.x, contig = a.chunk_contig[chunk];
    if (contig == 0xFFFFFFFFu) return;
    const uint32_t len = a.cand_len[k], mp = a.cand_modpos[k], slot = a.cand_slot[k];
    const uint8_t *m = a.masks + a.cand_mask_off[k];
    const size_t w = (size_t)chunk * CHUNK_WORDS + threadIdx.x;
    const long long g0 = (long long)w * 32;                               // global position of the word's bit 0
    const long long p0 = (long long)(chunk - a.contig_chunk[contig]) * CHUNK_BP + 32ll * threadIdx.x;      // ... inside its contig
    const long long clen = (long long)a.contig_len[contig];
    auto inside = [&](long long d) -> uint32_t {                          // bits b of the word with 0 <= p0 + b + d < clen
        const long long lo = -(p0 + d), hi = clen - (p0 + d);
        const long long l = lo < 0 ? 0 : lo, h = hi > 32 ? 32 : hi;
        if (h <= l) return 0u;
        const uint32_t upto_h = h >= 32 ? 0xFFFFFFFFu : ((1u << h) - 1u);
        return upto_h & ~(l ? ((1u << l) - 1u) : 0u);
    };
    uint32_t acc[2];
    for (int strand = 0; strand < 2; ++strand) {
        uint32_t s = inside(0);
        for (uint32_t j = 0; j < len && s; ++j) {
            const uint32_t set_f = m[j] & 15u;
            if (set_f == 15u) continue;
            const long long d = strand == 0 ? (long long)j - mp : (long long)mp - j;
            const uint32_t set = strand == 0 ? set_f : (((set_f & 1) << 3) | ((set_f & 2) << 1) | ((set_f & 4) >> 1) | ((set_f & 8) >> 3));
            const uint32_t ok = inside(d);
            if (!ok) { s = 0; break; }
            const uint32_t h = plane_bits_at(a.seq.H, a.n_words, g0 + d), l = plane_bits_at(a.seq.L, a.n_words, g0 + d), v = plane_bits_at(a.seq.V, a.n_words, g0 + d);
            uint32_t match = 0;                                           // A = 00, C = 01, G = 11, T = 10 as (H, L)
            if (set & NM_BASE_A) match |= ~h & ~l;
            if (set & NM_BASE_C) match |= ~h & l;
            if (set & NM_BASE_G) match |= h & l;
            if (set & NM_BASE_T) match |= h & ~l;
            s &= match & v & ok;
        }
        acc[strand] = s;
    }
    const StatePlanes &st = a.st[slot];
    unsigned int n_mod = __popc(acc[0] & st.MP[w]) + __popc(acc[1] & st.MM[w]);
    unsigned int n_non = __popc(acc[0] & st.UP[w]) + __popc(acc[1] & st.UM[w]);
    for (int o = 32; o; o >>= 1) { n_mod += __shfl_xor(n_mod, o); n_non += __shfl_xor(n_non, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = n_mod; part[1][threadIdx.x >> 6] = n_non; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int x = part[0][0] + part[0][1] + part[0][2] + part[0][3], y = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (x) atomicAdd(a.out + 2 * (size_t)k, (unsigned long long)x);
        if (y) atomicAdd(a.out + 2 * (size_t)k + 1, (unsigned long long)y);
    }
}

// Site masks of one candidate over the chunks of one contig (general planes) for nm_hit_positions.
template <int GN, int GP>
__global__ __launch_bounds__(256) void hits_kernel(Planes seq, StatePlanes st, uint32_t chunk0, uint32_t n_chunks,
                                                   const uint32_t *__restrict__ prog, int which,
                                                   uint32_t *__restrict__ out) {
    using K = Variant<GN, GP, false, 1, false, false>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ck = blockIdx.x * 4 + wave;
    if (ck >= n_chunks) return;
    const uint32_t chunk = chunk0 + ck;
    const StatePlanes stp[1] = {st};
    RawChunk<K> raw;
    raw.load(seq, stp, chunk, lane);
    Tile<K> tile;
    tile.expand(raw);
    uint32_t acc[T_WORDS];
#pragma unroll
    for (int t = 0; t < T_WORDS; ++t) acc[t] = 0xFFFFFFFFu;
    eval_strand<K>((cu32p)(prog + (which >= 2 ? K::PDW : 0)), tile, acc);
#pragma unroll
    for (int t = 0; t < T_WORDS; ++t) {
        // (a select chain, not raw.s[0][which][t]: a register array indexed at run time lives in private memory)
        const uint32_t state = which == 0 ? raw.s[0][0][t] : which == 1 ? raw.s[0][1][t] : which == 2 ? raw.s[0][2][t] : raw.s[0][3][t];
        out[(size_t)ck * CHUNK_WORDS + lane * T_WORDS + t] = acc[t] & state;
    }
}

// nm_hit_positions, device-side compaction of the site masks hits_kernel wrote: only the hit INDICES cross PCIe.
// (1) set bits per chunk (256 words = one workgroup)
__global__ __launch_bounds__(256) void hits_count_kernel(const uint32_t *__restrict__ masks, uint32_t *__restrict__ chunk_cnt) {
    __shared__ uint32_t part[4];
    uint32_t n = __popc(masks[(size_t)blockIdx.x * CHUNK_WORDS + threadIdx.x]);
    for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// (2) exclusive prefix over the chunks (one workgroup walks them 256 at a time), total in off[n_chunks]
__global__ __launch_bounds__(256) void hits_scan_kernel(const uint32_t *__restrict__ chunk_cnt, uint32_t n_chunks, unsigned long long *__restrict__ off) {
    __shared__ unsigned long long buf[256];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_chunks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_chunks ? chunk_cnt[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                   // Hillis-Steele inclusive scan
            const unsigned long long add = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_chunks) off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n_chunks] = carry;
}

// (3) every word writes the contig-local positions of its bits at its rank (ascending: words, then bits)
__global__ __launch_bounds__(256) void hits_scatter_kernel(const uint32_t *__restrict__ masks, const unsigned long long *__restrict__ off,
                                                           long long *__restrict__ out, unsigned long long capacity) {
    __shared__ uint32_t buf[256];
    uint32_t m = masks[(size_t)blockIdx.x * CHUNK_WORDS + threadIdx.x];
    const uint32_t n = __popc(m);
    buf[threadIdx.x] = n;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long at = off[blockIdx.x] + buf[threadIdx.x] - n;
    const long long w0 = ((long long)blockIdx.x * CHUNK_WORDS + threadIdx.x) * 32;
    while (m && at < capacity) {
        out[at++] = w0 + (__ffs(m) - 1);
        m &= m - 1;
    }
}

}  // namespace


// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
namespace nmdetail {

static nm_alloc_fn g_alloc = nullptr;
static nm_free_fn g_free = nullptr;
static void *g_alloc_user = nullptr;
static int g_live_ctx = 0;          // contexts alive: the allocator pair may only change while there is none

hipError_t device_alloc(void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) return hipSuccess;
    if (g_alloc) return g_alloc(g_alloc_user, p, bytes) == 0 && *p ? hipSuccess : hipErrorOutOfMemory;
    return hipMalloc(p, bytes);
}

hipError_t device_free(void *p) {
    if (!p) return hipSuccess;
    if (g_free) return g_free(g_alloc_user, p) == 0 ? hipSuccess : hipErrorInvalidValue;
    return hipFree(p);
}

__global__ __launch_bounds__(256) void stage_in_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n_copy, uint4 *__restrict__ zero,
                                                       uint32_t n_zero) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t k = i; k < n_copy; k += stride) dst[k] = src[k];
    for (uint32_t k = i; k < n_zero; k += stride) zero[k] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void stage_out_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t k = i; k < n; k += stride) dst[k] = src[k];
}

int stage_in(hipStream_t st, const void *h_src, void *d_dst, size_t bytes, void *d_zero, size_t zero_bytes) {
    const uint32_t n_copy = (uint32_t)((bytes + 15) / 16), n_zero = (uint32_t)((zero_bytes + 15) / 16);
    const uint32_t blocks = std::max(1u, std::min(256u, (std::max(n_copy, n_zero) + 255) / 256));
    hipLaunchKernelGGL(stage_in_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint4 *>(h_src), static_cast<uint4 *>(d_dst), n_copy,
                       static_cast<uint4 *>(d_zero), n_zero);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

int stage_out(hipStream_t st, const void *d_src, void *h_dst, size_t bytes) {
    const uint32_t n = (uint32_t)((bytes + 15) / 16);
    hipLaunchKernelGGL(stage_out_kernel, dim3(std::max(1u, std::min(256u, (n + 255) / 256))), dim3(256), 0, st, static_cast<const uint4 *>(d_src),
                       static_cast<uint4 *>(h_dst), n);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

int ensure_stage(nm_ctx *c, size_t bytes, int mode) {
    // A pair whose pinned half still holds the results of an open begin / end batch (nm_score_batch_begin,
    // nm_win_batch_w_begin) is never handed out: the next user would copy over — or, growing the pair, free — what
    // nm_*_end has yet to read.  At most two pairs are held at a time, so the walk always finds a free one.
    auto held = [&](int i) {
        for (int f = 0; f < NM_FLIGHTS; ++f)
            if ((c->score_wait[f].open && c->score_wait[f].stage == &c->stage[i]) || (c->win_wait[f].open && c->win_wait[f].stage == &c->stage[i]) ||
                (c->spec_wait[f].open && c->spec_wait[f].stage == &c->stage[i]))
                return true;
        return false;
    };
    int idx;
    std::unique_lock<std::mutex> lk(c->wait_mu);
    if (mode >= 2) idx = NM_STAGE_RING + (mode - 2);
    else if (mode == 1) {
        idx = c->stage_next_deep;
        for (int tries = 0; tries < NM_STAGE_RING && held(idx); ++tries) idx = (idx + 1) % NM_STAGE_RING;
        c->stage_next_deep = (idx + 1) % NM_STAGE_RING;
    } else {
        idx = c->stage_next;
        if (held(idx)) idx ^= 1;
        c->stage_next = idx ^ 1;
        for (int tries = 0; tries < NM_STAGE_RING && held(idx); ++tries) idx = (idx + 1) % NM_STAGE_RING;     // (two flights may hold both)
    }
    if (held(idx)) return fail(NM_ESTATE, "staging ring: every pair is held by an uncollected batch");
    lk.unlock();
    nm_ctx::Stage &st = c->stage[idx];
    if (!st.busy) HIP_TRY(hipEventCreateWithFlags(&st.busy, hipEventDisableTiming));
    if (st.pending) {
        HIP_TRY(hipEventSynchronize(st.busy));
        st.pending = false;
    }
    bytes += 32;                                           // stage_in / stage_out move whole 16-byte words
    if (bytes > st.bytes) {
        const size_t nb = std::max(bytes, st.bytes * 2);
        if (st.d) (void)nmdetail::dev_free(st.d);
        if (st.h) (void)hipHostFree(st.h);
        st.d = st.h = nullptr;
        st.bytes = 0;
        HIP_TRY(nmdetail::dev_malloc(&st.d, nb));
        HIP_TRY(hipHostMalloc(&st.h, nb, hipHostMallocDefault));
        st.bytes = nb;
    }
    c->d_stage = st.d;
    c->h_stage = st.h;
    c->cur_stage = &st;
    return NM_OK;
}

int release_stage(nm_ctx *c, hipStream_t s) {   // call after the last device work that reads the acquired pair was enqueued
    HIP_TRY(hipEventRecord(c->cur_stage->busy, s ? s : c->stream));
    c->cur_stage->pending = true;
    return NM_OK;
}

void busy_begin(nm_ctx *c) {
    if (!c->ev_collect_all) return;
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= 65536) return;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        c->ev_pool.emplace_back(a, b);
    }
    (void)hipEventRecord(c->ev_pool[c->ev_used].first, c->stream);
    c->busy_open = true;
}

void busy_end(nm_ctx *c) {
    if (!c->ev_collect_all || !c->busy_open) return;
    (void)hipEventRecord(c->ev_pool[c->ev_used].second, c->stream);
    c->ev_used += 1;
    c->busy_open = false;
}

int join_lanes(nm_ctx *c) {
    if (c->lane_stream && c->lane_pending) {
        HIP_TRY(hipStreamSynchronize(c->lane_stream));
        c->lane_pending = false;
    }
    return NM_OK;
}

}  // namespace nmdetail

namespace {


// Launch shape of a batch.  heavy: one workgroup column per classification (blockIdx.y), candidates VALU-bound;
// light (<= 6 candidates per (slot, bin) group on average — a round of the greedy search): HBM-bound, groups share their
// common constraints (CF), two classifications (the usual 6mA + 5mC batch) share one pass over the sequence planes.
struct LaunchShape {
    int wide;                 // 0: offsets in [-32, 31], 1: [-64, 63], 2: [-96, 95]
    bool compact, lit, light;
    uint32_t n_active;
    bool per_contig = false;
};

template <class K>
void launch_variant(const ScoreArgs &a, uint32_t gx, uint32_t gy, hipStream_t s) {
    hipLaunchKernelGGL((score_kernel<K>), dim3(gx, gy), dim3(256), 0, s, a);
}

template <int G, bool COMPACT, bool LIT>
void launch_by_load(const ScoreArgs &a, uint32_t gx, const LaunchShape &sh, hipStream_t s) {
    if (!sh.light) launch_variant<Variant<G, G, COMPACT, 1, LIT, false>>(a, gx, std::max(sh.n_active, 1u), s);
    else if (sh.n_active == 2 && LIT) launch_variant<Variant<G, G, COMPACT, 2, LIT, true>>(a, gx, 1, s);   // (fusing two slots on the 8-plane tile needs 128 VGPRs)
    else launch_variant<Variant<G, G, COMPACT, 1, LIT, true>>(a, gx, std::max(sh.n_active, 1u), s);
}

void launch_score(const ScoreArgs &a, uint32_t gx, const LaunchShape &sh, hipStream_t s) {
    if (sh.per_contig) {      // (candidate, contig) counters: the plain one-slot-per-column kernels with the other reduction key
        const uint32_t gy = std::max(sh.n_active, 1u);
        if (sh.wide == 2 && sh.compact) launch_variant<Variant<3, 3, true, 1, false, false, true>>(a, gx, gy, s);
        else if (sh.wide == 2) launch_variant<Variant<3, 3, false, 1, false, false, true>>(a, gx, gy, s);
        else if (!sh.wide && sh.compact) launch_variant<Variant<1, 1, true, 1, false, false, true>>(a, gx, gy, s);
        else if (!sh.wide) launch_variant<Variant<1, 1, false, 1, false, false, true>>(a, gx, gy, s);
        else if (sh.compact) launch_variant<Variant<2, 2, true, 1, false, false, true>>(a, gx, gy, s);
        else launch_variant<Variant<2, 2, false, 1, false, false, true>>(a, gx, gy, s);
        return;
    }
    if (sh.wide == 2) {       // motifs that reach more than 63 positions from the modified base (search frames above 127)
        if (sh.compact) launch_by_load<3, true, false>(a, gx, sh, s);
        else launch_by_load<3, false, false>(a, gx, sh, s);
    } else if (!sh.wide && sh.compact) {
        if (sh.lit) launch_by_load<1, true, true>(a, gx, sh, s);
        else launch_by_load<1, true, false>(a, gx, sh, s);
    } else if (!sh.wide) launch_by_load<1, false, false>(a, gx, sh, s);
    else if (sh.compact) launch_by_load<2, true, false>(a, gx, sh, s);
    else launch_by_load<2, false, false>(a, gx, sh, s);
}

int score_impl(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
               const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
               const uint8_t *cand_masks, unsigned long long *d_out, int64_t *h_out, const uint64_t *row_offset = nullptr,
               bool defer = false, const SpecSource *spec = nullptr, hipStream_t spec_stream = nullptr, int flight = 0) {
    // defer: host counts, collected later by nm_score_batch_end (the call returns with everything enqueued)
    // spec: the motifs are written on the device (SpecSource, nmscan_internal.h); always deferred, noted in c->spec_wait, on spec_stream
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    nm_ctx::Waiting &wait_slot = spec ? c->spec_wait[flight] : c->score_wait[flight];
    if (defer) {
        if (wait_slot.open) return fail(NM_ESTATE, "nm_score_batch_begin: the previous batch has not been collected (nm_score_batch_end)");
        nmdetail::wait_set(c, wait_slot, nm_ctx::Waiting{nullptr, 0, nullptr, true});
    }
    if (n_cand == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    // ---- one pass over the candidates: validate, classify the batch, count per (slot, bin) bucket.
    // Candidates whose bin has no contig on this device (multi-GPU shard) contribute zero and get no program.
    const uint32_t n_bins = c->n_bins;
    std::vector<uint32_t> &bucket = c->bucket;          // counting sort by (slot, bin)
    bucket.assign((size_t)NM_MAX_MOD_SLOTS * n_bins + 1, 0);
    int any_wide = 0;                                    // widest offset class of the batch: 0 / 1 / 2 (see LaunchShape)
    bool all_compact = true, all_literal = true;
    uint32_t n_prog = 0;
    uint64_t mask_bytes = 0, mask_lo = ~0ull;   // byte range of cand_masks the resident candidates reference
    bool slot_used[NM_MAX_MOD_SLOTS] = {};
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t slot = cand_mod_slot[k], bin = cand_bin[k];
        if (slot >= NM_MAX_MOD_SLOTS || !c->slots[slot].present)
            return fail(NM_ESTATE, "candidate %u uses mod slot %u with no pileup uploaded", k, slot);
        if (bin >= n_bins) return fail(NM_EINVAL, "candidate %u: bin %u >= n_bins %u", k, bin, n_bins);
        if (c->bin_nchunks[bin] == 0) continue;         // not resident here: the rank that holds the bin validates it
        if (spec) {                                      // motif unknown here: literal, compact, narrow by contract; mask_stride bytes each
            mask_lo = 0;
            mask_bytes = std::max<uint64_t>(mask_bytes, (uint64_t)(k + 1) * spec->mask_stride);
            bucket[(size_t)slot * n_bins + bin + 1] += 1;
            slot_used[slot] = true;
            n_prog += 1;
            continue;
        }
        const uint32_t len = cand_len[k], mp = cand_modpos[k];
        if (len == 0 || len > NM_MAX_MOTIF_LEN) return fail(NM_ERANGE, "candidate %u: motif length %u outside 1..%d", k, len, NM_MAX_MOTIF_LEN);
        if (mp >= len) return fail(NM_EINVAL, "candidate %u: mod_position %u outside motif of length %u", k, mp, len);
        const uint8_t *m = cand_masks + cand_mask_offset[k];
        uint32_t all_and = 15u, any_zero = 0, any_set = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const uint32_t v = m[j] & 15u;
            all_and &= v;
            any_zero |= (v == 0);
            any_set |= (v != 15u) & ((v & (v - 1)) != 0);      // a 2- or 3-base set
        }
        if (any_set) all_literal = false;
        if (any_zero) return fail(NM_EINVAL, "candidate %u: empty base set in the motif", k);
        if (all_and == 15u) return fail(NM_EINVAL, "candidate %u: motif has no specified position", k);
        mask_lo = std::min<uint64_t>(mask_lo, cand_mask_offset[k]);
        mask_bytes = std::max<uint64_t>(mask_bytes, (uint64_t)cand_mask_offset[k] + len);
        // offsets relative to the modified base span [-mp, len-1-mp] forward and the mirror image in reverse
        {
            const uint32_t reach = std::max(mp, len - 1 - mp);
            if (reach > 95) return fail(NM_ERANGE, "candidate %u: a position %u away from the modified base (the engine reaches 95)", k, reach);
            any_wide = std::max(any_wide, reach > 63 ? 2 : reach > 31 ? 1 : 0);
        }
        const uint32_t can_mask = c->slots[slot].canonical == 'A' ? NM_BASE_A : NM_BASE_C;
        if ((m[mp] & 15u) != can_mask) all_compact = false;
        bucket[(size_t)slot * n_bins + bin + 1] += 1;
        slot_used[slot] = true;
        n_prog += 1;
    }
    if (n_prog == 0) { mask_lo = 0; mask_bytes = 0; }
    uint32_t active[NM_MAX_MOD_SLOTS], n_active = 0;
    int slot_to_active[NM_MAX_MOD_SLOTS];
    for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) {
        slot_to_active[sl] = -1;
        if (slot_used[sl]) { slot_to_active[sl] = (int)n_active; active[n_active++] = (uint32_t)sl; }
    }
    // ---- staging layout: candidate records (sorted) | orig_index | mask bytes | cand_range
    const bool per_contig = row_offset != nullptr;
    const uint64_t out_rows = per_contig ? row_offset[n_cand] : n_cand;
    if (per_contig)
        for (uint32_t k = 0; k < n_cand; ++k)
            if (row_offset[k + 1] - row_offset[k] != c->bin_ncontigs[cand_bin[k]])
                return fail(NM_EINVAL, "candidate %u: %llu output rows for a bin of %u resident contigs", k,
                            (unsigned long long)(row_offset[k + 1] - row_offset[k]), c->bin_ncontigs[cand_bin[k]]);
    const bool lit = all_literal && all_compact && !any_wide && !c->opt_no_lit && !per_contig;   // literal-only tiles exist for the narrow compact kernels
    const uint32_t np = lit ? 4u : 8u;
    const uint32_t pdw = 2u * (2u + 2u * (uint32_t)any_wide) * np;         // dwords per program: [strand][word-group][plane]
    const size_t rec_bytes = (size_t)n_prog * sizeof(CandRec);
    const size_t off_orig = (rec_bytes + 15) & ~(size_t)15;
    const size_t off_masks = (off_orig + (size_t)n_prog * 4 + 15) & ~(size_t)15;
    const size_t off_range = (off_masks + (mask_bytes - mask_lo) + 15) & ~(size_t)15;
    const uint32_t n_entries = std::max(n_active, 1u) * n_bins;
    const size_t range_bytes = (size_t)n_entries * sizeof(uint4);
    const size_t off_rows = (off_range + range_bytes + 15) & ~(size_t)15;          // per-contig mode: row base per sorted candidate
    // The static segment table covers every bin; a batch whose candidates sit in a few bins only — the long tail of the
    // search: a handful of tasks still open among hundreds of bins — would launch thousands of workgroups that find no
    // candidate range and leave.  Such a batch brings its own table: the segments of the bins it has candidates for.
    std::vector<uint8_t> bin_active(n_bins, 0);
    uint32_t n_active_bins = 0;
    size_t n_active_segs = 0;
    for (uint32_t b = 0; b < n_bins; ++b) {
        bool act = false;
        for (int sl = 0; sl < NM_MAX_MOD_SLOTS && !act; ++sl) act = slot_used[sl] && bucket[(size_t)sl * n_bins + b + 1] > 0;
        if (!act) continue;
        bin_active[b] = 1;
        n_active_bins += 1;
        n_active_segs += (c->bin_nchunks[b] + c->seg_chunks - 1) / c->seg_chunks;
    }
    const bool own_segments = n_prog && n_active_segs * 4 <= (size_t)c->n_segments * 3 && getenv("NM_ALL_SEGMENTS") == nullptr;
    const size_t off_segs = (off_rows + (per_contig ? (size_t)n_prog * 8 : 0) + 15) & ~(size_t)15;
    const size_t off_posof = ((own_segments ? off_segs + n_active_segs * sizeof(uint4) : off_segs) + 15) & ~(size_t)15;     // spec: sorted position of candidate k
    const size_t off_xin = (off_posof + (size_t)n_cand * 4 + 15) & ~(size_t)15;                                            // spec: the writer's extra input
    const size_t total = spec ? off_xin + spec->extra_in
                              : own_segments ? off_segs + n_active_segs * sizeof(uint4) : off_rows + (per_contig ? (size_t)n_prog * 8 : 0);
    // host counts come back through the pinned half of the staging pair (a copy into pageable memory is staged by the
    // runtime and costs tens of microseconds more per round of the search); tables too large for that go directly
    const size_t out_bytes = (size_t)out_rows * 2 * sizeof(int64_t);
    const bool via_stage = defer || (h_out && out_bytes <= ((size_t)4 << 20));
    const size_t off_counts = (total + 15) & ~(size_t)15;
    const size_t off_xout = (off_counts + out_bytes + 15) & ~(size_t)15;                // spec: the writer's extra output, behind the counts
    const size_t stage_end = spec ? off_xout + spec->extra_out : off_counts + out_bytes;
    int rc = ensure_stage(c, via_stage ? stage_end : total, 1);
    if (rc) return rc;
    // scoring lane of this call (nm_set_score_lanes): asynchronous device-output batches take the lane of their staging
    // pair — a pair, its half of the program table and its stream are reused together; every other call runs on the
    // ctx stream after the second lane has drained
    hipStream_t sst = spec ? spec_stream : c->stream;
    const bool laned = c->score_lanes == 2 && d_out && !h_out && !spec && !defer;
    if (!laned && !spec) {
        rc = join_lanes(c);
        if (rc) return rc;
    } else if (laned && ((c->cur_stage - c->stage) & 1)) {
        sst = c->lane_stream;
        c->lane_pending = true;
    }
    if (laned) {
        // two launches that write the same count table are ordered (memset, kernel) only within one stream: when an
        // earlier launch in flight wrote d_out from the other lane, this lane waits for it (its staging pair's event)
        for (auto &st : c->stage)
            if (&st != c->cur_stage && st.pending && st.last_out == d_out && st.last_stream && st.last_stream != sst)
                HIP_TRY(hipStreamWaitEvent(sst, st.busy, 0));
    }
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage);
    CandRec *h_rec = reinterpret_cast<CandRec *>(hs);
    uint32_t *h_orig = reinterpret_cast<uint32_t *>(hs + off_orig);
    uint4 *h_range = reinterpret_cast<uint4 *>(hs + off_range);
    if (!spec) memcpy(hs + off_masks, cand_masks + mask_lo, mask_bytes - mask_lo);
    memset(h_range, 0, range_bytes);
    if (own_segments) {
        uint4 *h_segs = reinterpret_cast<uint4 *>(hs + off_segs);
        size_t at = 0;
        for (uint32_t b = 0; b < n_bins; ++b)
            if (bin_active[b])
                for (uint32_t k = 0; k < c->bin_nchunks[b]; k += c->seg_chunks)
                    h_segs[at++] = make_uint4(c->bin_chunk0[b] + k, std::min<uint32_t>(c->seg_chunks, c->bin_nchunks[b] - k), b, 0);
    }
    uint32_t n_groups = 0;
    // exclusive prefix over the buckets -> first sorted index of each (slot, bin); stable within a bucket
    for (size_t i = 1; i < bucket.size(); ++i) bucket[i] += bucket[i - 1];
    for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) {
        if (slot_to_active[sl] < 0) continue;
        for (uint32_t b = 0; b < n_bins; ++b) {
            const uint32_t lo = bucket[(size_t)sl * n_bins + b], hi = bucket[(size_t)sl * n_bins + b + 1];
            if (hi > lo) {
                h_range[(size_t)slot_to_active[sl] * n_bins + b] = make_uint4(lo, hi - lo, 0xFFFFFFFFu, 0);
                n_groups += 1;
            }
        }
    }
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t slot = cand_mod_slot[k], bin = cand_bin[k];
        if (c->bin_nchunks[bin] == 0) continue;
        const uint32_t at = bucket[(size_t)slot * n_bins + bin]++;
        if (spec) {
            h_rec[at] = CandRec{k * spec->mask_stride, k, 0, 0, (uint8_t)slot, 0};
            h_orig[at] = k;
            reinterpret_cast<uint32_t *>(hs + off_posof)[k] = at;
            continue;
        }
        h_rec[at] = CandRec{(uint32_t)(cand_mask_offset[k] - mask_lo), k, cand_len[k], cand_modpos[k], (uint8_t)slot, 0};
        h_orig[at] = k;
        if (per_contig) reinterpret_cast<uint64_t *>(hs + off_rows)[at] = row_offset[k];
    }
    // measured crossover (profiles/): up to ~6 candidates per (slot, bin) group the launch is HBM-bound
    const bool light = (uint64_t)n_prog <= 6ull * n_groups && !per_contig;
    const bool cf = light && !c->opt_no_cf;
    // device-side program buffer: one part per staging pair, so that compiling the next batches (on the copy stream)
    // overlaps the scoring kernel of batch k; reuse of a part is gated like its staging pair (ensure_stage).  A light
    // batch appends one common program per (slot, bin) entry.
    const size_t need_dw = ((size_t)std::max(n_prog, 1u) + (cf ? n_entries : 0)) * pdw;
    if (c->prog_cap_dw < need_dw) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->copy_stream));
        HIP_TRY(nmdetail::sync_flight_streams(c));
        rc = join_lanes(c);
        if (rc) return rc;
        if (c->d_programs) (void)nmdetail::dev_free(c->d_programs);
        c->d_programs = nullptr;
        c->prog_cap_dw = 0;
        HIP_TRY(nmdetail::dev_malloc(&c->d_programs, need_dw * 2 * 4 * NM_STAGE_RING));
        c->prog_cap_dw = need_dw * 2;
    }
    uint32_t *const d_prog = c->d_programs + (size_t)(c->cur_stage - c->stage) * c->prog_cap_dw;
    // staged tables travel and are compiled on the copy stream; the scoring stream waits for the compiled programs
    // a call that returns host counts has nothing to overlap the compile with (the caller waits for this very batch):
    // it prepares on the scoring stream itself — no event, no cross-stream wait in its chain of launch latencies
    const bool inline_prep = (h_out || defer) && !c->opt_no_inline;
    hipStream_t pst = inline_prep ? sst : c->copy_stream;
    if (spec && spec->extra_in) spec->fill_host(hs + off_xin);
    uint8_t *ds = static_cast<uint8_t *>(c->d_stage);
    // ---- output counters
    unsigned long long *out = spec ? reinterpret_cast<unsigned long long *>(ds + off_counts) : d_out;
    if (!out) {
        if (c->counts_cap < out_rows) {
            if (c->d_counts) (void)nmdetail::dev_free(c->d_counts);
            c->d_counts = nullptr;
            c->counts_cap = 0;
            HIP_TRY(nmdetail::dev_malloc(&c->d_counts, (size_t)out_rows * 2 * sizeof(unsigned long long) * 2));
            c->counts_cap = (size_t)out_rows * 2;
        }
        out = c->d_counts;
    }
    // a batch somebody waits for (a round of the search): tables in, outputs cleared, results out by KERNELS on its own stream — no
    // copy engine in the chain (nmscan_internal.h: stage_in); everything else keeps the copy engines
    const bool by_kernels = inline_prep && pst == sst && total <= STAGE_KERNEL_MAX && (via_stage || !h_out) && getenv("NM_STAGE_COPIES") == nullptr;
    const size_t clear_bytes = spec ? stage_end - off_counts : (size_t)out_rows * 2 * sizeof(unsigned long long);
    if (by_kernels) {
        rc = stage_in(pst, c->h_stage, c->d_stage, total, out, clear_bytes);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, total, hipMemcpyHostToDevice, pst));
    }
    hipEvent_t e0 = c->ev0, e1 = c->ev1;
    const bool timed_launch = c->ev_collect || !defer;          // (a deferred batch of the search is not asked for its kernel time)
    if (c->ev_collect) {
        if (c->ev_used == c->ev_pool.size()) {
            if (c->ev_pool.size() >= 65536) return fail(NM_ESTATE, "timing pool exhausted: call nm_timing_reset");
            hipEvent_t a_ = nullptr, b_ = nullptr;
            HIP_TRY(hipEventCreate(&a_));
            HIP_TRY(hipEventCreate(&b_));
            c->ev_pool.emplace_back(a_, b_);
        }
        e0 = c->ev_pool[c->ev_used].first;
        e1 = c->ev_pool[c->ev_used].second;
        c->ev_used += 1;
    }
    if (spec) {
        // counts and the writer's extra output: one clear; the writer (window kernel + children + their programs); timed with the scoring
        if (!by_kernels) HIP_TRY(hipMemsetAsync(ds + off_counts, 0, clear_bytes, pst));
        if (timed_launch) HIP_TRY(hipEventRecord(e0, pst));
        const SpecCompile cc{reinterpret_cast<CandRec *>(ds), reinterpret_cast<const uint32_t *>(ds + off_posof), ds + off_masks,
                             reinterpret_cast<uint4 *>(ds + off_range), d_prog, pdw, n_prog, (int)np, any_wide, all_compact ? 1 : 0, cf ? 1 : 0};
        rc = spec->launch(pst, ds + off_xin, ds + off_xout, cc);
        if (rc) return rc;
        *spec->h_extra_out = hs + off_xout;
    }
    if (n_prog && !spec) {
        if (cf && n_prog <= 2048 && n_entries <= 8192 && !c->opt_no_inline) {
            const size_t lds_bytes = (size_t)n_prog * pdw * 4 <= 48 * 1024 ? (size_t)n_prog * pdw * 4 : 0;
            hipLaunchKernelGGL(compile_common_kernel, dim3(1), dim3(1024), lds_bytes, pst, n_prog, reinterpret_cast<const CandRec *>(ds),
                               ds + off_masks, d_prog, any_wide, (int)np, all_compact ? 1 : 0, n_entries,
                               reinterpret_cast<uint4 *>(ds + off_range), pdw, 8u, (uint32_t)(lds_bytes / 4));
            HIP_TRY(hipGetLastError());
        } else {
            hipLaunchKernelGGL(compile_kernel, dim3((n_prog + 255) / 256), dim3(256), 0, pst, n_prog,
                               reinterpret_cast<const CandRec *>(ds), ds + off_masks, d_prog, any_wide, (int)np,
                               all_compact ? 1 : 0);
            HIP_TRY(hipGetLastError());
            if (cf) {
                hipLaunchKernelGGL(common_kernel, dim3((n_entries + 63) / 64), dim3(64), 0, pst, n_entries,
                                   reinterpret_cast<uint4 *>(ds + off_range), d_prog, pdw, n_prog, 8u);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    if (!inline_prep) {
        HIP_TRY(hipEventRecord(c->copy_done, c->copy_stream));
        // strict order: the host waits for the compiled programs (tens of microseconds, the previous batch is still being
        // scored) and the scoring queue carries no cross-stream barrier packet — 5-6 us less between two launches; with two
        // lanes the host must run ahead instead, so there the stream waits
        if (laned || c->opt_stream_wait) HIP_TRY(hipStreamWaitEvent(sst, c->copy_done, 0));
        else HIP_TRY(hipEventSynchronize(c->copy_done));
    }
#ifdef NM_SCORE_PROBES
    // probe builds only (NM_CXXFLAGS=-DNM_SCORE_PROBES, tools/gpu_step_gap.sh): NM_SCORE_PROBE=1 leaves the clear out (the counts are garbage)
    static const int score_probe = getenv("NM_SCORE_PROBE") ? atoi(getenv("NM_SCORE_PROBE")) : 0;
#else
    constexpr int score_probe = 0;                       // (the shipped library has no switch that changes what a call computes)
#endif
    if (!spec && !by_kernels && !(score_probe & 1)) HIP_TRY(hipMemsetAsync(out, 0, clear_bytes, sst));
    // ---- launch
    ScoreArgs a{};
    a.seq = Planes{c->dH, c->dL, c->dV, c->d_needs_v};
    for (int s = 0; s < NM_MAX_MOD_SLOTS; ++s) {
        const ModSlot &ms = c->slots[s];
        a.st[s] = StatePlanes{ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4], ms.planes[5]};
    }
    a.segments = own_segments ? reinterpret_cast<const uint4 *>(ds + off_segs) : c->d_segments;
    a.n_segments = own_segments ? (uint32_t)n_active_segs : c->n_segments;
    a.n_bins = c->n_bins;
    a.programs = d_prog;
    a.orig_index = reinterpret_cast<uint32_t *>(ds + off_orig);
    a.cand_range = reinterpret_cast<uint4 *>(ds + off_range);
    a.out = out;
    a.chunk_rank = c->d_chunk_rank;
    a.row_base = reinterpret_cast<const uint64_t *>(ds + off_rows);
    for (uint32_t i = 0; i < n_active; ++i) a.active_slot[i] = active[i];
    for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) a.slot_is_c[sl] = c->slots[sl].canonical == 'C';
    // workgroups that will find candidates, against what the device runs at once (~6 per CU): below ~2 rounds of
    // workgroups the last, partly filled round is a large share of the launch -> smaller pieces
    uint64_t est_wgs = (uint64_t)a.n_segments * n_groups / std::max<uint32_t>(own_segments ? n_active_bins : n_bins, 1);
    if (light && n_active == 2 && lit) est_wgs = (est_wgs + 1) / 2;   // fused slots: one workgroup serves both
    const uint64_t resident = (uint64_t)c->n_cus * 6;
    uint32_t split_log2 = 0;
    while (split_log2 < 2 && (est_wgs << split_log2) < 2 * resident) ++split_log2;      // measured (tools/gpu_r2f.sh): halves win below ~2 rounds, quarters never
    if (c->opt_split >= 0) split_log2 = (uint32_t)c->opt_split;
    a.split_log2 = split_log2;
    // grid: runs of pieces (8 for the XCD-remapped heavy variants, 1 for the streaming ones); in every run the pieces a
    // full round of resident workgroups would take LAST are cut finer (down to 4 chunks = one per wave)
    const bool streaming = light && !c->opt_no_cf;
    const uint32_t runs = streaming ? 1u : 8u;
    const uint32_t n_pieces = a.n_segments << split_log2;
    a.pieces_per_run = (n_pieces + runs - 1) / runs;
    a.fine_log2 = c->opt_fine >= 0 ? (uint32_t)c->opt_fine : 2u - split_log2;
    const uint32_t cols = std::max(1u, (streaming && n_active == 2 && lit) ? 1u : n_active);
    uint32_t tail_pieces = (uint32_t)std::min<uint64_t>(a.pieces_per_run, (resident / cols + runs - 1) / runs);
    if (a.fine_log2 == 0) tail_pieces = 0;
    a.j_big = a.pieces_per_run - tail_pieces;
    const uint32_t gx = (a.j_big + (tail_pieces << a.fine_log2)) * runs;
    const LaunchShape shape{any_wide, all_compact, lit, light && !c->opt_no_cf, n_active, per_contig};
    const bool fuse = n_active == 2 && shape.light && lit;

    if (timed_launch && !spec) HIP_TRY(hipEventRecord(e0, sst));
    if (n_prog) launch_score(a, gx, shape, sst);   // else nothing resident for this batch: the zeroed table is the answer
    HIP_TRY(hipGetLastError());
    if (timed_launch) HIP_TRY(hipEventRecord(e1, sst));
    if (by_kernels && (spec || via_stage)) {
        rc = stage_out(sst, out, hs + off_counts, spec ? stage_end - off_counts : out_bytes);
        if (rc) return rc;
    } else if (spec) HIP_TRY(hipMemcpyAsync(hs + off_counts, ds + off_counts, stage_end - off_counts, hipMemcpyDeviceToHost, sst));
    else if (via_stage) HIP_TRY(hipMemcpyAsync(hs + off_counts, out, out_bytes, hipMemcpyDeviceToHost, sst));
    rc = release_stage(c, sst);
    if (rc) return rc;
    c->cur_stage->last_out = d_out;
    c->cur_stage->last_stream = sst;
    c->last_score_stream = sst;
    c->timed = !c->ev_collect && timed_launch;
    c->launches += 1;
    c->last_wgs = (uint64_t)gx * (fuse ? 1 : std::max(n_active, 1u));
    c->last_compact = all_compact ? n_prog : 0;
    c->last_general = all_compact ? 0 : n_prog;
    if (defer) {
        nmdetail::wait_set(c, wait_slot, nm_ctx::Waiting{hs + off_counts, out_bytes, c->cur_stage, true});
    } else if (via_stage) {
        HIP_TRY(hipEventSynchronize(c->cur_stage->busy));
        memcpy(h_out, hs + off_counts, out_bytes);
    } else if (h_out) {
        HIP_TRY(hipMemcpyAsync(h_out, out, out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return NM_OK;
}

}  // namespace

// error sink for the other translation units of the library (nmbed.cpp)
int nm_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" {

int nm_abi_version(void) { return 1; }

int nm_set_device_allocator(nm_alloc_fn alloc, nm_free_fn free_fn, void *user) {
    if ((alloc == nullptr) != (free_fn == nullptr)) return fail(NM_EINVAL, "give both functions or neither");
    // a block must go back to the allocator it came from: with a ctx alive its planes, staging buffers and tables would
    // be released through the other pair (hipFree on a pool sub-block, the pool's free on a hipMalloc pointer)
    if (nmdetail::g_live_ctx > 0 && (alloc != nmdetail::g_alloc || free_fn != nmdetail::g_free || user != nmdetail::g_alloc_user))
        return fail(NM_ESTATE, "nm_set_device_allocator with %d nm_ctx alive: destroy them first", nmdetail::g_live_ctx);
    nmdetail::g_alloc = alloc;
    nmdetail::g_free = free_fn;
    nmdetail::g_alloc_user = user;
    return NM_OK;
}

const char *nm_last_error(void) { return g_err.c_str(); }

static int ctx_init(nm_ctx *c) {
    c->opt_no_lit = getenv("NM_NO_LIT") != nullptr;
    c->opt_no_cf = getenv("NM_NO_CF") != nullptr;
    c->opt_no_inline = getenv("NM_NO_INLINE_PREP") != nullptr;
    c->opt_stream_wait = getenv("NM_STREAM_WAIT") != nullptr;      // A/B switch: never wait for the compile on the host
    if (const char *e = getenv("NM_FINE")) c->opt_fine = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("NM_SPLIT")) c->opt_split = std::max(0, std::min(2, atoi(e)));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->n_cus = (uint32_t)cus;
    }
    if (const char *e = getenv("NM_SEG_CHUNKS")) c->seg_chunks = (uint32_t)std::max(4, atoi(e));
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    // the second scoring lane is made here, next to the first: the runtime deals its hardware queues to streams in
    // creation order, and two lanes that share a queue do not overlap (profiles/r2/lanes_queue_ab.txt)
    HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(nmdetail::dev_malloc(&c->d_err, sizeof(unsigned int)));
    HIP_TRY(nmdetail::dev_malloc(&c->d_other, sizeof(unsigned long long)));
    return NM_OK;
}

int nm_ctx_create(int device, nm_ctx **out) {
    if (!out) return fail(NM_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(NM_EINVAL, "device %d not in 0..%d", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    nm_ctx *c = new (std::nothrow) nm_ctx();
    if (!c) return fail(NM_ENOMEM, "out of host memory");
    c->device = device;
    c->stream = nullptr;
    nmdetail::g_live_ctx += 1;
    const int rc = ctx_init(c);
    if (rc != NM_OK) {                 // nm_ctx_destroy releases whatever was created (the error text is already set)
        const std::string keep = g_err;
        (void)nm_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return NM_OK;
}

static void free_assembly(nm_ctx *c) {
    drop_ingest_rows(c);
    nmdetail::free_readstats(c);
    void *ptrs[] = {c->dH /* owns L and V too */, c->d_needs_v, c->d_contig_chunk, c->d_contig_len, c->d_segments, c->d_chunk_rank};
    c->d_chunk_rank = nullptr;
    for (void *p : ptrs)
        if (p) (void)nmdetail::dev_free(p);
    c->dH = c->dL = c->dV = nullptr;
    c->d_needs_v = nullptr;
    c->d_contig_chunk = nullptr;
    c->d_contig_len = nullptr;
    c->d_segments = nullptr;
    for (int b = 0; b < 4; ++b) {
        if (c->d_rank[b]) (void)nmdetail::dev_free(c->d_rank[b]);
        if (c->d_base_total[b]) (void)nmdetail::dev_free(c->d_base_total[b]);
        c->d_rank[b] = nullptr;
        c->d_base_total[b] = nullptr;
    }
    for (auto &s : c->slots) {
        free_slot_planes(s);
        s.present = false;
        s.n_rows = 0;
        drop_slot_ranks(s);
    }
}

int nm_ctx_destroy(nm_ctx *c) {
    if (!c) return NM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);      // a registered pool's free does not wait like hipFree
    if (c->lane_stream) (void)hipStreamSynchronize(c->lane_stream);
    c->lane_pending = false;
    (void)nm_comm_destroy(c);
    free_assembly(c);
    for (auto &st : c->stage) {
        if (st.d) (void)nmdetail::dev_free(st.d);
        if (st.h) (void)hipHostFree(st.h);
        if (st.busy) (void)hipEventDestroy(st.busy);
    }
    if (c->d_counts) (void)nmdetail::dev_free(c->d_counts);
    if (c->d_programs) (void)nmdetail::dev_free(c->d_programs);
    if (c->d_spec_bg) (void)nmdetail::dev_free(c->d_spec_bg);
    for (int f = 0; f < NM_FLIGHTS; ++f) {
        if (c->d_spec_counts[f]) (void)nmdetail::dev_free(c->d_spec_counts[f]);
        if (c->d_flight_counts[f]) (void)nmdetail::dev_free(c->d_flight_counts[f]);
    }
    for (hipStream_t s : c->flight_stream) if (s) (void)hipStreamDestroy(s);
    if (c->d_win_planes) (void)nmdetail::dev_free(c->d_win_planes);
    if (c->d_win_alive) (void)nmdetail::dev_free(c->d_win_alive);
    if (c->d_win_tasks) (void)nmdetail::dev_free(c->d_win_tasks);
    if (c->d_err) (void)nmdetail::dev_free(c->d_err);
    if (c->d_other) (void)nmdetail::dev_free(c->d_other);
    for (auto &pr : c->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->lane_stream) (void)hipStreamDestroy(c->lane_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    nmdetail::g_live_ctx -= 1;
    return NM_OK;
}

int nm_set_stream(nm_ctx *c, void *hip_stream) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = join_lanes(c);
    if (rc) return rc;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->last_score_stream = nullptr;
    c->timed = false;
    return NM_OK;
}

int nm_set_score_lanes(nm_ctx *c, int lanes) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (lanes != 1 && lanes != 2) return fail(NM_EINVAL, "lanes must be 1 or 2, got %d", lanes);
    HIP_TRY(hipSetDevice(c->device));
    int rc = join_lanes(c);
    if (rc) return rc;
    if (lanes == 2 && !c->lane_stream) HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream, hipStreamNonBlocking));
    c->score_lanes = lanes;
    return NM_OK;
}

int nm_sync(nm_ctx *c) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return join_lanes(c);
}

static int upload_contigs_impl(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id,
                               uint32_t n_bins, const uint8_t *seq_ascii, bool on_device, const uint64_t *src_off = nullptr) {
    // src_off (device sources only): where each contig's bytes start in seq_ascii; NULL = back to back, at offsets[i]
    // n_contigs == 0 is a valid shard: a rank of a multi-GPU run that received no contig (more GPUs than pieces) holds
    // the bins' numbering and two pad chunks, scores every candidate to zero and still joins every collective
    if (!c || !offsets || (n_contigs && (!bin_id || !seq_ascii))) return fail(NM_EINVAL, "NULL argument");
    if (n_bins == 0) return fail(NM_EINVAL, "need at least one bin");
    if (offsets[0] != 0) return fail(NM_EINVAL, "offsets[0] must be 0");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    {
        const int rcj = join_lanes(c);
        if (rcj) return rcj;
    }
    free_assembly(c);
    // any failure below leaves the ctx WITHOUT an assembly (dH == nullptr, so scoring refuses) and frees the temporaries
    struct Rollback {
        nm_ctx *c;
        void *tmp[3] = {nullptr, nullptr, nullptr};
        bool keep = false;
        ~Rollback() {
            (void)hipStreamSynchronize(c->stream);          // nothing may still read the temporaries
            for (void *p : tmp)
                if (p) (void)nmdetail::dev_free(p);
            if (!keep) free_assembly(c);
        }
    } guard{c};
    c->n_contigs = n_contigs;
    c->n_bins = n_bins;
    c->contig_len.assign(n_contigs, 0);
    c->contig_bin.assign(bin_id, bin_id + (bin_id ? n_contigs : 0));
    c->contig_chunk.assign(n_contigs, 0);
    c->contig_nchunks.assign(n_contigs, 0);
    c->total_bp = offsets[n_contigs];
    for (uint32_t i = 0; i < n_contigs; ++i) {
        if (offsets[i + 1] <= offsets[i]) return fail(NM_EINVAL, "contig %u is empty (seq.py:70 asserts non-empty)", i);
        if (bin_id[i] >= n_bins) return fail(NM_EINVAL, "contig %u: bin %u >= n_bins %u", i, bin_id[i], n_bins);
        c->contig_len[i] = offsets[i + 1] - offsets[i];
        if (c->contig_len[i] >= 0xFFFFFFFFull - CHUNK_BP) return fail(NM_ERANGE, "contig %u longer than 4 Gbp", i);
    }
    // group contigs by bin; chunk 0 and the last chunk are zero pads for the halo reads
    std::vector<uint32_t> order(n_contigs);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return bin_id[x] < bin_id[y]; });
    c->bin_chunk0.assign(n_bins, 0);
    c->bin_nchunks.assign(n_bins, 0);
    c->bin_ncontigs.assign(n_bins, 0);
    c->contig_rank.assign(n_contigs, 0);
    uint64_t next = 1;
    std::vector<uint32_t> chunk_contig(1, 0xFFFFFFFFu);
    for (uint32_t oi = 0; oi < n_contigs; ++oi) {
        const uint32_t i = order[oi];
        const uint64_t nch = (c->contig_len[i] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP;
        if (next + nch + 1 >= 0xFFFFFFFFull / CHUNK_WORDS * 8) return fail(NM_ERANGE, "assembly too large for one device context");
        c->contig_chunk[i] = (uint32_t)next;
        c->contig_nchunks[i] = (uint32_t)nch;
        const uint32_t b = bin_id[i];
        c->contig_rank[i] = c->bin_ncontigs[b]++;              // position of the contig inside its bin (upload order)
        if (c->bin_nchunks[b] == 0) c->bin_chunk0[b] = (uint32_t)next;
        c->bin_nchunks[b] += (uint32_t)nch;
        chunk_contig.insert(chunk_contig.end(), nch, i);
        next += nch;
    }
    chunk_contig.push_back(0xFFFFFFFFu);
    c->n_chunks = (uint32_t)(next + 1);
    const size_t words = plane_words(c);
    HIP_TRY(nmdetail::dev_malloc(&c->dH, words * 4 * 3));          // one block for H, L, V (see alloc_slot_planes)
    c->dL = c->dH + words;
    c->dV = c->dL + words;
    HIP_TRY(nmdetail::dev_malloc(&c->d_needs_v, c->n_chunks));
    HIP_TRY(hipMemsetAsync(c->dH, 0, words * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->dL, 0, words * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->dV, 0, words * 4, c->stream));
    HIP_TRY(nmdetail::dev_malloc(&c->d_contig_chunk, (size_t)std::max(n_contigs, 1u) * 4));
    HIP_TRY(nmdetail::dev_malloc(&c->d_contig_len, (size_t)std::max(n_contigs, 1u) * 8));
    HIP_TRY(hipMemcpyAsync(c->d_contig_chunk, c->contig_chunk.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_contig_len, c->contig_len.data(), (size_t)n_contigs * 8, hipMemcpyHostToDevice, c->stream));
    // temporaries for the pack pass
    uint8_t *d_ascii = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_chunk_contig = nullptr;
    if (on_device) d_ascii = const_cast<uint8_t *>(seq_ascii);
    else {
        HIP_TRY(nmdetail::dev_malloc(&d_ascii, std::max<uint64_t>(c->total_bp, 1)));
        guard.tmp[0] = d_ascii;
    }
    HIP_TRY(nmdetail::dev_malloc(&d_off, (size_t)(n_contigs + 1) * 8));
    guard.tmp[1] = d_off;
    HIP_TRY(nmdetail::dev_malloc(&d_chunk_contig, (size_t)c->n_chunks * 4));
    guard.tmp[2] = d_chunk_contig;
    if (!on_device && c->total_bp) HIP_TRY(hipMemcpyAsync(d_ascii, seq_ascii, c->total_bp, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_off, src_off ? src_off : offsets, (size_t)(n_contigs + (src_off ? 0 : 1)) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_chunk_contig, chunk_contig.data(), (size_t)c->n_chunks * 4, hipMemcpyHostToDevice, c->stream));
    {   // per chunk: rank of its contig within its bin (the row of the per-contig counters)
        std::vector<uint32_t> chunk_rank(c->n_chunks, 0);
        for (uint32_t ch = 0; ch < c->n_chunks; ++ch)
            if (chunk_contig[ch] != 0xFFFFFFFFu) chunk_rank[ch] = c->contig_rank[chunk_contig[ch]];
        HIP_TRY(nmdetail::dev_malloc(&c->d_chunk_rank, (size_t)c->n_chunks * 4));
        HIP_TRY(hipMemcpy(c->d_chunk_rank, chunk_rank.data(), (size_t)c->n_chunks * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemsetAsync(c->d_other, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(pack_kernel, dim3(c->n_chunks), dim3(256), 0, c->stream, d_ascii, d_off, c->d_contig_len, d_chunk_contig,
                       c->d_contig_chunk, c->dH, c->dL, c->dV, c->d_other);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(needs_v_kernel, dim3(c->n_chunks), dim3(CHUNK_WORDS), 0, c->stream, c->dV, c->d_needs_v, c->n_chunks);
    HIP_TRY(hipGetLastError());
    // static segment table: every bin's chunk range cut into pieces of SEG_CHUNKS
    std::vector<uint4> segs;
    for (uint32_t b = 0; b < n_bins; ++b)
        for (uint32_t k = 0; k < c->bin_nchunks[b]; k += c->seg_chunks)
            segs.push_back(make_uint4(c->bin_chunk0[b] + k, std::min<uint32_t>(c->seg_chunks, c->bin_nchunks[b] - k), b, 0));
    c->n_segments = (uint32_t)segs.size();
    HIP_TRY(nmdetail::dev_malloc(&c->d_segments, std::max<size_t>(segs.size(), 1) * sizeof(uint4)));
    if (!segs.empty()) HIP_TRY(hipMemcpyAsync(c->d_segments, segs.data(), segs.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
    unsigned long long other = 0;
    HIP_TRY(hipMemcpyAsync(&other, c->d_other, sizeof other, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->other_letters = other;
    guard.keep = true;
    return NM_OK;
}

static int upload_pileup_impl(nm_ctx *c, uint32_t mod_slot, uint8_t canonical_base, double low, double high,
                              uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                              const uint8_t *strand, const double *fraction_mod, int append, bool on_device) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (mod_slot >= NM_MAX_MOD_SLOTS) return fail(NM_EINVAL, "mod_slot %u >= %d", mod_slot, NM_MAX_MOD_SLOTS);
    if (canonical_base != 'A' && canonical_base != 'C') return fail(NM_EINVAL, "canonical base must be 'A' or 'C'");
    if (!(high > low)) return fail(NM_EINVAL, "high threshold must exceed low (find_motifs_bin.py:117-118)");
    if (n_rows && (!contig_id || !position || !strand || !fraction_mod)) return fail(NM_EINVAL, "NULL column");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));      // an asynchronous scoring launch may still read the planes replaced below
    {
        const int rcj = join_lanes(c);
        if (rcj) return rcj;
    }
    ModSlot &ms = c->slots[mod_slot];
    drop_slot_ranks(ms);
    const size_t words = plane_words(c);
    if (!ms.present || !append) {
        HIP_TRY(alloc_slot_planes(ms, words));
        HIP_TRY(hipMemsetAsync(ms.planes[0], 0, words * 4 * 6, c->stream));
        ms.n_rows = 0;
    } else if (ms.canonical != canonical_base || ms.low != low || ms.high != high) {
        return fail(NM_EINVAL, "append with different canonical base / thresholds than the slot holds");
    }
    ms.present = true;
    ms.canonical = canonical_base;
    ms.low = low;
    ms.high = high;
    if (n_rows == 0) return NM_OK;
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
    const uint32_t can_h = 0, can_l = canonical_base == 'C' ? 1u : 0u;   // A = 00, C = 01
    // host columns stream through the pinned staging ring in slabs; device columns are consumed in place
    const uint64_t slab = on_device ? n_rows : (8ull << 20);  // rows per slab (17 B each)
    for (uint64_t r0 = 0; r0 < n_rows; r0 += slab) {
        const uint64_t n = std::min(slab, n_rows - r0);
        const uint32_t *d_cid = contig_id + r0, *d_pos = position + r0;
        const uint8_t *d_str = strand + r0;
        const double *d_frac = fraction_mod + r0;
        int rc = NM_OK;
        if (!on_device) {
            const size_t o_pos = n * 4, o_frac = ((o_pos + n * 4) + 7) & ~(size_t)7, o_str = o_frac + n * 8;
            rc = ensure_stage(c, o_str + n);
            if (rc) return rc;
            uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
            memcpy(hs, contig_id + r0, n * 4);
            memcpy(hs + o_pos, position + r0, n * 4);
            memcpy(hs + o_frac, fraction_mod + r0, n * 8);
            memcpy(hs + o_str, strand + r0, n);
            HIP_TRY(hipMemcpyAsync(ds, hs, o_str + n, hipMemcpyHostToDevice, c->stream));
            d_cid = reinterpret_cast<uint32_t *>(ds);
            d_pos = reinterpret_cast<uint32_t *>(ds + o_pos);
            d_str = ds + o_str;
            d_frac = reinterpret_cast<double *>(ds + o_frac);
        }
        hipLaunchKernelGGL(state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, d_cid, d_pos,
                           d_str, d_frac, low, high, c->d_contig_chunk, c->d_contig_len, c->n_contigs, can_h, can_l,
                           c->dH, c->dL, c->dV, ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4],
                           ms.planes[5], c->d_err);
        HIP_TRY(hipGetLastError());
        if (!on_device) {
            rc = release_stage(c);
            if (rc) return rc;
        }
    }
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err) {
        // the planes already carry part of the bad rows: the slot goes back to "no pileup" (scoring then refuses it with
        // NM_ESTATE) instead of staying half-written; an append loses what the slot held before, which is corrupt too
        ms.present = false;
        ms.n_rows = 0;
    } else {
        ms.n_rows += n_rows;
    }
    if (err & 1u) return fail(NM_EINVAL, "pileup row with contig_id / position outside the uploaded assembly");
    if (err & 2u) return fail(NM_EINVAL, "pileup strand must be '+' or '-'");
    if (err & 4u) return fail(NM_EINVAL, "duplicate (contig, position, strand) rows: the reference's np.isin(assume_unique=True) requires unique positions (find_motifs_bin.py:1258)");
    return NM_OK;
}

int nm_upload_contigs(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id, uint32_t n_bins,
                      const uint8_t *seq_ascii) {
    return upload_contigs_impl(c, n_contigs, offsets, bin_id, n_bins, seq_ascii, false);
}

int nm_upload_contigs_device(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id,
                             uint32_t n_bins, const uint8_t *d_seq_ascii) {
    return upload_contigs_impl(c, n_contigs, offsets, bin_id, n_bins, d_seq_ascii, true);
}

}  // extern "C"

// the count table of a deferred batch of a flight: grown on demand behind the stream that may still write the old one
static int flight_table(nm_ctx *c, unsigned long long **table, size_t *cap, size_t rows, hipStream_t st) {
    if (*cap >= rows) return NM_OK;
    HIP_TRY(hipStreamSynchronize(st));
    if (*table) (void)nmdetail::dev_free(*table);
    *table = nullptr;
    *cap = 0;
    HIP_TRY(nmdetail::dev_malloc(table, rows * 2 * sizeof(unsigned long long) * 2));
    *cap = rows * 2;
    return NM_OK;
}

int nmdetail::score_batch_spec_begin(nm_ctx *c, int flight, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const SpecSource &spec,
                                     hipStream_t st) {
    if (n_cand && (!cand_bin || !cand_mod_slot)) return fail(NM_EINVAL, "NULL argument");
    const int rc = score_impl(c, n_cand, cand_bin, cand_mod_slot, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, &spec, st, flight);
    if (rc) nmdetail::wait_close(c, c->spec_wait[flight]);
    return rc;
}

int nmdetail::score_batch_flight_begin(nm_ctx *c, int flight, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const uint8_t *cand_len,
                                       const uint8_t *cand_modpos, const uint32_t *cand_mask_offset, const uint8_t *cand_masks) {
    if (!c || flight < 0 || flight >= NM_FLIGHTS) return fail(NM_EINVAL, "bad flight");
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks)) return fail(NM_EINVAL, "NULL argument");
    if (c->score_wait[flight].open) return fail(NM_ESTATE, "score batch: the previous batch of the flight has not been collected");
    int rc = flight_table(c, &c->d_flight_counts[flight], &c->flight_counts_cap[flight], n_cand, c->stream);
    if (rc) return rc;
    rc = score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks, c->d_flight_counts[flight], nullptr, nullptr, true,
                    nullptr, nullptr, flight);
    if (rc) nmdetail::wait_close(c, c->score_wait[flight]);
    return rc;
}

int nmdetail::score_batch_flight_end(nm_ctx *c, int flight, int64_t *out_counts) {
    if (!c || flight < 0 || flight >= NM_FLIGHTS) return fail(NM_EINVAL, "bad flight");
    if (!c->score_wait[flight].open) return fail(NM_ESTATE, "score batch end without begin");
    const nm_ctx::Waiting w = c->score_wait[flight];
    nmdetail::WaitCloser closer{c, &c->score_wait[flight], nullptr};          // (the pair stays held until the counts are copied out)
    if (w.bytes == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(w.stage->busy));
    if (out_counts) memcpy(out_counts, w.h, w.bytes);           // NULL: the batch is dropped
    return NM_OK;
}

// nmfasta.hip: the contigs picked out of a device-resident packed sequence (any subset, any order, a record more than once)
int nmdetail::upload_contigs_gather(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint64_t *src_off, const uint32_t *bin_id,
                                    uint32_t n_bins, const uint8_t *d_seq_ascii) {
    return upload_contigs_impl(c, n_contigs, offsets, bin_id, n_bins, d_seq_ascii, true, src_off);
}

extern "C" {

int nm_upload_pileup(nm_ctx *c, uint32_t mod_slot, uint8_t canonical_base, double low, double high, uint64_t n_rows,
                     const uint32_t *contig_id, const uint32_t *position, const uint8_t *strand,
                     const double *fraction_mod, int append) {
    return upload_pileup_impl(c, mod_slot, canonical_base, low, high, n_rows, contig_id, position, strand, fraction_mod,
                              append, false);
}

int nm_upload_pileup_device(nm_ctx *c, uint32_t mod_slot, uint8_t canonical_base, double low, double high,
                            uint64_t n_rows, const uint32_t *d_contig_id, const uint32_t *d_position,
                            const uint8_t *d_strand, const double *d_fraction_mod, int append) {
    return upload_pileup_impl(c, mod_slot, canonical_base, low, high, n_rows, d_contig_id, d_position, d_strand,
                              d_fraction_mod, append, true);
}

int nm_score_batch(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                   const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                   const uint8_t *cand_masks, int64_t *out_counts) {
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !out_counts))
        return fail(NM_EINVAL, "NULL argument");
    return score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks, nullptr, out_counts);
}

int nm_score_batch_wide(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const uint16_t *cand_len,
                        const uint16_t *cand_modpos, const uint32_t *cand_mask_offset, const uint8_t *cand_masks, int64_t *out_counts) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !out_counts)) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    if (n_cand == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = join_lanes(c);
    if (rc) return rc;
    uint64_t mask_bytes = 0;
    uint32_t max_chunks = 0;
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t slot = cand_mod_slot[k], bin = cand_bin[k], len = cand_len[k], mp = cand_modpos[k];
        if (slot >= NM_MAX_MOD_SLOTS || !c->slots[slot].present) return fail(NM_ESTATE, "candidate %u uses mod slot %u with no pileup uploaded", k, slot);
        if (bin >= c->n_bins) return fail(NM_EINVAL, "candidate %u: bin %u >= n_bins %u", k, bin, c->n_bins);
        if (len == 0 || len > NM_MAX_WIDE_MOTIF_LEN) return fail(NM_ERANGE, "candidate %u: motif length %u outside 1..%d", k, len, NM_MAX_WIDE_MOTIF_LEN);
        if (mp >= len) return fail(NM_EINVAL, "candidate %u: mod_position %u outside motif of length %u", k, mp, len);
        const uint8_t *m = cand_masks + cand_mask_offset[k];
        uint32_t all_and = 15u;
        for (uint32_t j = 0; j < len; ++j) {
            if ((m[j] & 15u) == 0) return fail(NM_EINVAL, "candidate %u: empty base set in the motif", k);
            all_and &= m[j] & 15u;
        }
        if (all_and == 15u) return fail(NM_EINVAL, "candidate %u: motif has no specified position", k);
        mask_bytes = std::max<uint64_t>(mask_bytes, (uint64_t)cand_mask_offset[k] + len);
        max_chunks = std::max(max_chunks, c->bin_nchunks[bin]);
    }
    memset(out_counts, 0, (size_t)n_cand * 2 * sizeof(int64_t));
    if (max_chunks == 0) return NM_OK;                                    // none of the bins is resident on this rank
    // (plain synchronous staging: tables in, counts out; this path is not timed by anything)
    std::vector<uint32_t> chunk_contig(c->n_chunks, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < c->n_contigs; ++i)
        for (uint32_t q = 0; q < c->contig_nchunks[i]; ++q) chunk_contig[c->contig_chunk[i] + q] = i;
    const size_t o_bin = 0, o_off = o_bin + (size_t)n_cand * 4, o_len = o_off + (size_t)n_cand * 4, o_mp = o_len + (size_t)n_cand * 2,
                 o_slot = o_mp + (size_t)n_cand * 2, o_masks = (o_slot + n_cand + 15) & ~(size_t)15, o_b0 = (o_masks + mask_bytes + 15) & ~(size_t)15,
                 o_bn = o_b0 + (size_t)c->n_bins * 4, o_cc = o_bn + (size_t)c->n_bins * 4, o_out = (o_cc + (size_t)c->n_chunks * 4 + 15) & ~(size_t)15,
                 total = o_out + (size_t)n_cand * 16;
    std::vector<uint8_t> h(total, 0);
    memcpy(h.data() + o_bin, cand_bin, (size_t)n_cand * 4);
    memcpy(h.data() + o_off, cand_mask_offset, (size_t)n_cand * 4);
    memcpy(h.data() + o_len, cand_len, (size_t)n_cand * 2);
    memcpy(h.data() + o_mp, cand_modpos, (size_t)n_cand * 2);
    memcpy(h.data() + o_slot, cand_mod_slot, n_cand);
    memcpy(h.data() + o_masks, cand_masks, mask_bytes);
    memcpy(h.data() + o_b0, c->bin_chunk0.data(), (size_t)c->n_bins * 4);
    memcpy(h.data() + o_bn, c->bin_nchunks.data(), (size_t)c->n_bins * 4);
    memcpy(h.data() + o_cc, chunk_contig.data(), (size_t)c->n_chunks * 4);
    uint8_t *d = nullptr;
    HIP_TRY(nmdetail::dev_malloc(&d, total));
    struct Free { uint8_t *p; nm_ctx *c; ~Free() { (void)hipStreamSynchronize(c->stream); (void)nmdetail::dev_free(p); } } guard{d, c};
    HIP_TRY(hipMemcpyAsync(d, h.data(), total, hipMemcpyHostToDevice, c->stream));
    WideArgs a{};
    a.seq = Planes{c->dH, c->dL, c->dV, c->d_needs_v};
    for (int s = 0; s < NM_MAX_MOD_SLOTS; ++s) {
        const ModSlot &ms = c->slots[s];
        a.st[s] = StatePlanes{ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4], ms.planes[5]};
    }
    a.cand_bin = reinterpret_cast<const uint32_t *>(d + o_bin);
    a.cand_mask_off = reinterpret_cast<const uint32_t *>(d + o_off);
    a.cand_len = reinterpret_cast<const uint16_t *>(d + o_len);
    a.cand_modpos = reinterpret_cast<const uint16_t *>(d + o_mp);
    a.cand_slot = d + o_slot;
    a.masks = d + o_masks;
    a.bin_chunk0 = reinterpret_cast<const uint32_t *>(d + o_b0);
    a.bin_nchunks = reinterpret_cast<const uint32_t *>(d + o_bn);
    a.chunk_contig = reinterpret_cast<const uint32_t *>(d + o_cc);
    a.contig_chunk = c->d_contig_chunk;
    a.contig_len = c->d_contig_len;
    a.out = reinterpret_cast<unsigned long long *>(d + o_out);
    a.n_words = (long long)plane_words(c);
    hipLaunchKernelGGL(score_wide_kernel, dim3(max_chunks, n_cand), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_counts, d + o_out, (size_t)n_cand * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NM_OK;
}

int nm_score_batch_begin(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                         const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                         const uint8_t *cand_masks) {
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks))
        return fail(NM_EINVAL, "NULL argument");
    if (c && c->score_wait[0].open) return fail(NM_ESTATE, "nm_score_batch_begin: the previous batch has not been collected (nm_score_batch_end)");
    const int rc = score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks, nullptr, nullptr, nullptr, true);
    if (rc && c) c->score_wait[0].open = false;      // (a batch that failed half way is not open)
    return rc;
}

int nm_score_batch_end(nm_ctx *c, int64_t *out_counts) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->score_wait[0].open) return fail(NM_ESTATE, "nm_score_batch_end without nm_score_batch_begin");
    const nm_ctx::Waiting w = c->score_wait[0];
    c->score_wait[0].open = false;
    if (w.bytes == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(w.stage->busy));
    if (out_counts) memcpy(out_counts, w.h, w.bytes);           // NULL: the batch is dropped
    return NM_OK;
}

int nm_score_batch_device(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                          const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                          const uint8_t *cand_masks, int64_t *d_out_counts) {
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !d_out_counts))
        return fail(NM_EINVAL, "NULL argument");
    return score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks,
                      reinterpret_cast<unsigned long long *>(d_out_counts), nullptr);
}

int nm_score_batch_per_contig(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const uint8_t *cand_len,
                              const uint8_t *cand_modpos, const uint32_t *cand_mask_offset, const uint8_t *cand_masks,
                              const uint64_t *row_offset, int64_t *out_counts) {
    if (!row_offset || (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !out_counts)))
        return fail(NM_EINVAL, "NULL argument");
    if (row_offset[0] != 0) return fail(NM_EINVAL, "row_offset[0] must be 0");
    return score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks, nullptr, out_counts, row_offset);
}

int nm_bin_contigs(nm_ctx *c, uint32_t bin, uint32_t *contig_ids, uint32_t capacity, uint32_t *n_contigs) {
    if (!c || !n_contigs) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    if (bin >= c->n_bins) return fail(NM_EINVAL, "bin %u >= n_bins %u", bin, c->n_bins);
    *n_contigs = c->bin_ncontigs[bin];
    if (contig_ids)
        for (uint32_t i = 0; i < c->n_contigs; ++i)
            if (c->contig_bin[i] == bin && c->contig_rank[i] < capacity) contig_ids[c->contig_rank[i]] = i;
    return NM_OK;
}

int nm_hit_positions(nm_ctx *c, uint32_t contig_id, uint32_t mod_slot, uint8_t len, uint8_t modpos, const uint8_t *masks,
                     int which, int64_t *out, uint64_t capacity, uint64_t *n_out) {
    if (!c || !masks || !n_out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    if (contig_id >= c->n_contigs) return fail(NM_EINVAL, "contig_id %u >= %u", contig_id, c->n_contigs);
    if (mod_slot >= NM_MAX_MOD_SLOTS || !c->slots[mod_slot].present) return fail(NM_ESTATE, "mod slot %u has no pileup", mod_slot);
    if (which < 0 || which > 3) return fail(NM_EINVAL, "which must be 0..3");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t full[PROG6_DW], prog[PROG6_DW];
    int reach = 0;
    int rc = compile_program(masks, len, modpos, full, &reach);
    if (rc) return rc;
    slice_program(full, reach + 1, prog);
    const uint32_t nch = c->contig_nchunks[contig_id];
    const size_t out_words = (size_t)nch * CHUNK_WORDS;
    // scratch: site masks of the contig | set bits per chunk | their exclusive prefix (+ total)
    const size_t o_cnt = PROG6_DW * 4 + out_words * 4, o_off = (o_cnt + (size_t)nch * 4 + 7) & ~(size_t)7;
    rc = ensure_stage(c, o_off + ((size_t)nch + 1) * 8);
    if (rc) return rc;
    memcpy(c->h_stage, prog, sizeof prog);
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, sizeof prog, hipMemcpyHostToDevice, c->stream));
    uint8_t *ds = static_cast<uint8_t *>(c->d_stage);
    uint32_t *d_prog = reinterpret_cast<uint32_t *>(ds);
    uint32_t *d_masks = d_prog + PROG6_DW;
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(ds + o_cnt);
    unsigned long long *d_off = reinterpret_cast<unsigned long long *>(ds + o_off);
    const ModSlot &ms = c->slots[mod_slot];
    Planes seq{c->dH, c->dL, c->dV, c->d_needs_v};
    StatePlanes st{ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4], ms.planes[5]};
    dim3 grid((nch + 3) / 4);
    if (reach == 2) hipLaunchKernelGGL((hits_kernel<3, 3>), grid, dim3(256), 0, c->stream, seq, st, c->contig_chunk[contig_id], nch, d_prog, which, d_masks);
    else if (reach == 1) hipLaunchKernelGGL((hits_kernel<2, 2>), grid, dim3(256), 0, c->stream, seq, st, c->contig_chunk[contig_id], nch, d_prog, which, d_masks);
    else hipLaunchKernelGGL((hits_kernel<1, 1>), grid, dim3(256), 0, c->stream, seq, st, c->contig_chunk[contig_id], nch, d_prog, which, d_masks);
    HIP_TRY(hipGetLastError());
    // compaction on the device (popcount prefix + scatter): only the hit indices cross PCIe, not the contig's masks
    hipLaunchKernelGGL(hits_count_kernel, dim3(nch), dim3(256), 0, c->stream, d_masks, d_cnt);
    hipLaunchKernelGGL(hits_scan_kernel, dim3(1), dim3(256), 0, c->stream, d_cnt, nch, d_off);
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_off + nch, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t n = total, n_write = out ? std::min<uint64_t>(n, capacity) : 0;
    if (n_write) {
        long long *d_pos = nullptr;
        HIP_TRY(nmdetail::dev_malloc(&d_pos, n_write * 8));
        hipLaunchKernelGGL(hits_scatter_kernel, dim3(nch), dim3(256), 0, c->stream, d_masks, d_off, d_pos, (unsigned long long)n_write);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_pos, n_write * 8, hipMemcpyDeviceToHost, c->stream);
        const hipError_t e2 = hipStreamSynchronize(c->stream);
        (void)nmdetail::dev_free(d_pos);
        if (e != hipSuccess || e2 != hipSuccess) return fail(NM_EHIP, "hit compaction failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    rc = release_stage(c);
    if (rc) return rc;
    *n_out = n;
    return NM_OK;
}

int nm_parse_motifs(uint32_t n, const char *text, const uint32_t *text_offset, const int32_t *mod_position,
                    uint8_t *out_len, uint8_t *out_modpos, uint32_t *out_mask_offset, uint8_t *out_masks,
                    uint64_t masks_capacity, uint64_t *masks_used) {
    if (n && (!text || !text_offset || !mod_position || !out_len || !out_modpos || !out_mask_offset || !out_masks || !masks_used))
        return fail(NM_EINVAL, "NULL argument");
    uint64_t used = 0;
    uint8_t tmp[4096];
    for (uint32_t k = 0; k < n; ++k) {
        const char *s = text + text_offset[k];
        const uint32_t slen = text_offset[k + 1] - text_offset[k];
        uint32_t nt = 0;
        for (uint32_t i = 0; i < slen;) {
            uint8_t m = 0;
            const char ch = s[i];
            if (ch == '[') {
                uint32_t j = i + 1;
                while (j < slen && s[j] != ']') {
                    const char b = s[j++];
                    m |= b == 'A' ? NM_BASE_A : b == 'C' ? NM_BASE_C : b == 'G' ? NM_BASE_G : b == 'T' ? NM_BASE_T : 0x80;
                }
                if (j >= slen) return fail(NM_EINVAL, "motif %u: unmatched '[' (motif.py:239)", k);
                if (m == 0 || (m & 0x80)) return fail(NM_EINVAL, "motif %u: a bracket may only list A, C, G, T", k);
                i = j + 1;
            } else {
                m = ch == 'A' ? NM_BASE_A : ch == 'C' ? NM_BASE_C : ch == 'G' ? NM_BASE_G : ch == 'T' ? NM_BASE_T
                    : ch == '.' ? 15 : 0;
                if (m == 0)
                    return fail(NM_EINVAL, "motif %u: character '%c' is not A/C/G/T/./[..] — the reference scans motifs as "
                                           "regular expressions (utils.py:61), other letters would be literals", k, ch);
                i += 1;
            }
            if (nt >= sizeof tmp) return fail(NM_ERANGE, "motif %u longer than %zu positions", k, sizeof tmp);
            tmp[nt++] = m;
        }
        uint32_t lo = 0, hi = nt;
        while (lo < nt && tmp[lo] == 15) ++lo;
        if (lo == nt) { lo = 0; }                      // all dots: left as is (motif.py:218-219), rejected at scoring
        else while (hi > lo && tmp[hi - 1] == 15) --hi;
        const uint32_t len = hi - lo;
        const int64_t mp = (int64_t)mod_position[k] - (int64_t)lo;
        if (len > NM_MAX_MOTIF_LEN) return fail(NM_ERANGE, "motif %u: stripped length %u > %d", k, len, NM_MAX_MOTIF_LEN);
        if (mp < 0 || mp >= (int64_t)len) return fail(NM_EINVAL, "motif %u: mod_position %d outside the stripped motif", k, mod_position[k]);
        if (used + len > masks_capacity) return fail(NM_ERANGE, "masks buffer too small");
        memcpy(out_masks + used, tmp + lo, len);
        out_len[k] = (uint8_t)len;
        out_modpos[k] = (uint8_t)mp;
        out_mask_offset[k] = (uint32_t)used;
        used += len;
    }
    *masks_used = used;
    return NM_OK;
}

int nm_stats(nm_ctx *c, uint64_t what[8]) {
    if (!c || !what) return fail(NM_EINVAL, "NULL argument");
    what[0] = c->total_bp;
    what[1] = (uint64_t)c->n_chunks * CHUNK_BP;
    what[2] = (uint64_t)plane_words(c) * 4 * 3;
    what[3] = (uint64_t)plane_words(c) * 4 * 2;
    what[4] = c->launches;
    what[5] = c->last_wgs;
    what[6] = c->last_compact;
    what[7] = c->last_general;
    return NM_OK;
}

int nm_timing_reset(nm_ctx *c, int enable) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    {
        const int rcj = join_lanes(c);          // the event pairs are reused: nothing may still be about to record them
        if (rcj) return rcj;
    }
    c->ev_used = 0;
    c->ev_collect = enable != 0;
    c->ev_collect_all = enable == 2;
    c->busy_open = false;
    c->timed = false;
    return NM_OK;
}

int nm_timing_total_ms(nm_ctx *c, double *total_ms, uint64_t *n_launches) {
    if (!c || !total_ms || !n_launches) return fail(NM_EINVAL, "NULL argument");
    double tot = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(c->ev_pool[i].second));
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *n_launches = c->ev_used;
    return NM_OK;
}

int nm_last_kernel_ms(nm_ctx *c, float *ms) {
    if (!c || !ms) return fail(NM_EINVAL, "NULL argument");
    if (!c->timed) return fail(NM_ESTATE, "no scoring launch recorded on this stream");
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return NM_OK;
}

}  // extern "C"
