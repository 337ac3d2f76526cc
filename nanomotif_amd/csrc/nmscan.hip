// libnmscan — MI355X (gfx950 / CDNA4) motif scan + methylation count engine.  C ABI: include/nmscan.h.
//
// Design (DESIGN.md has the long form):
//   * Assembly resident in HBM as three bit planes over one padded coordinate space: H, L (the two bits of
//     the base code, "bit-sliced 2-bit") and V (1 = A/C/G/T inside a contig).  Contigs are grouped by bin and
//     start on 8192-bp chunk boundaries with >= 64 invalid positions after each one, so a match can never
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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/nmscan.h"

namespace {

constexpr int T_WORDS = 4;                              // 32-bit words per lane
constexpr int CHUNK_WORDS = 64 * T_WORDS;               // 256 words
constexpr int CHUNK_BP = CHUNK_WORDS * 32;              // 8192 positions per wave-chunk
constexpr int GAP_BP = 64;                              // invalid positions guaranteed after every contig
constexpr int SEG_CHUNKS = 16;                          // chunks per workgroup segment (128 Kbp)
constexpr int BMAX = 32;                                // candidates per LDS accumulation pass
constexpr int PROG_DW = 64;                             // host-side program: [strand 2][word-group 4][plane 8]
// device-side programs are packed to the word-groups the launched kernel variant reads:
// narrow (offsets in [-32, 31]) = groups 1..2 -> 32 dwords (128 B), wide = all four -> 64 dwords

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(NM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------------
struct Planes {
    const uint32_t *H, *L, *V;
    const uint8_t *needs_v;   // per chunk: 1 => V (and halo) must be consulted
};

struct StatePlanes {
    const uint32_t *M, *U;               // compact (strand implied by base)
    const uint32_t *MP, *UP, *MM, *UM;   // general
};

struct ScoreArgs {
    Planes seq;
    StatePlanes st[NM_MAX_MOD_SLOTS];
    const uint4 *segments;      // {first chunk, n chunks, bin, 0}
    uint32_t n_segments;
    uint32_t n_bins;
    const uint2 *cand_range;    // [active_slot_index][bin] -> {begin, count} into programs
    const uint32_t *programs;   // [n_prog][2 * (GN + GP) * 8], sorted by (slot, bin)
    const uint32_t *orig_index; // [n_cand] sorted -> caller order
    unsigned long long *out;    // [n_cand][2]
    uint32_t active_slot[NM_MAX_MOD_SLOTS];
    uint32_t slot_is_c[NM_MAX_MOD_SLOTS];   // canonical base of the slot is C (else A)
};

// Pack: one workgroup per chunk, 64 positions per wave per step, wave ballot builds the plane words.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii,
                                                   const uint64_t *__restrict__ contig_off,
                                                   const uint32_t *__restrict__ chunk_contig,
                                                   const uint32_t *__restrict__ chunk_first,
                                                   uint32_t *__restrict__ H, uint32_t *__restrict__ L,
                                                   uint32_t *__restrict__ V, unsigned long long *__restrict__ other) {
    const uint32_t chunk = blockIdx.x;
    const uint32_t contig = chunk_contig[chunk];
    if (contig == 0xFFFFFFFFu) return;                               // pad chunk, planes already zero
    const uint64_t beg = contig_off[contig], len = contig_off[contig + 1] - beg;
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

// needs_v[c] = 0 iff chunk c is entirely valid and so are the two words either side of it.
__global__ void needs_v_kernel(const uint32_t *__restrict__ V, uint8_t *__restrict__ needs_v, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x;
    __shared__ uint32_t all_and;
    if (threadIdx.x == 0) all_and = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t v = V[(size_t)c * CHUNK_WORDS + threadIdx.x];
    if (threadIdx.x < 2) {
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

// ------------------------------------------------------------------------------------------------------
// Raw pileup ingestion: the reference's three pre-filters evaluated on the device (dataload.py:191-247, in the order
// of find_motifs_bin.py:399-414), then classification into the state planes and a compact list of the surviving
// confidently methylated rows (the input of window extraction, find_motifs_bin.py:625-661).
// ------------------------------------------------------------------------------------------------------
constexpr int NM_MAX_MOD_CODES = 8;

struct RawRows {
    uint64_t n;
    const uint32_t *contig;      // engine-local contig id, 0xFFFFFFFF = contig not resident (row ignored)
    const uint32_t *position;
    const int8_t *mod;           // 0..NM_MAX_MOD_CODES-1
    const uint8_t *strand;
    const double *frac;
    const int32_t *nvalid;
};

// (1) coverage filter + per (contig, mod code) counts for the frequency filter
__global__ void ingest_count_kernel(RawRows r, uint32_t n_contigs, const uint64_t *__restrict__ contig_len,
                                    int min_cov, double meth_thr, unsigned int *cnt /*[contig][mod][2]*/, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    const uint32_t c = r.contig[i];
    if (c == 0xFFFFFFFFu) return;
    const int m = r.mod[i];
    if (c >= n_contigs || r.position[i] >= contig_len[c] || m < 0 || m >= NM_MAX_MOD_CODES) { atomicOr(err, 1u); return; }
    if (r.nvalid[i] <= min_cov) return;                                     // dataload.py:199: Nvalid_cov > 5
    unsigned int *p = cnt + ((size_t)c * NM_MAX_MOD_CODES + m) * 2;
    atomicAdd(p, 1u);
    if (r.frac[i] > meth_thr) atomicAdd(p + 1, 1u);                         // dataload.py:215: fraction_mod > 0.7
}

// (2) frequency filter verdict per (contig, mod code): n_mod / n > 1e-4 and n_mod > 50 (dataload.py:218-219)
__global__ void ingest_group_kernel(uint32_t n_groups, const unsigned int *__restrict__ cnt, double min_freq,
                                    unsigned int min_mods, uint8_t *ok) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const unsigned int n = cnt[g * 2], nm = cnt[g * 2 + 1];
    ok[g] = n > 0 && ((double)nm / (double)n) > min_freq && nm > min_mods;
}

__device__ __forceinline__ bool ingest_row_alive(const RawRows &r, uint64_t i, int min_cov, const uint8_t *ok,
                                                 uint32_t *c_out, bool *plus_out) {
    const uint32_t c = r.contig[i];
    const int m = r.mod[i];
    if (c == 0xFFFFFFFFu || m < 0 || m >= NM_MAX_MOD_CODES || r.nvalid[i] <= min_cov) return false;
    if (!ok[(size_t)c * NM_MAX_MOD_CODES + m]) return false;
    const uint8_t st = r.strand[i];
    if (st != '+' && st != '-') return false;      // other strand labels form groups of their own and are never scored
    *c_out = c;
    *plus_out = st == '+';
    return true;
}

// (3a) adjacency filter, scatter: per strand the maximal fraction at every position (mod codes mixed, dataload.py:237)
__global__ void ingest_scatter_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                      const uint32_t *__restrict__ contig_chunk, unsigned long long *dense_plus,
                                      unsigned long long *dense_minus) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    uint32_t c;
    bool plus;
    if (!ingest_row_alive(r, i, min_cov, ok, &c, &plus)) return;
    const uint64_t g = (uint64_t)contig_chunk[c] * CHUNK_BP + r.position[i];
    // fractions are >= 0, so their IEEE bit patterns order like the values
    atomicMax((plus ? dense_plus : dense_minus) + g, (unsigned long long)__double_as_longlong(r.frac[i]));
}

// (3b) adjacency verdict + classification + confident-row list.  A row survives iff its fraction equals the maximum
// over positions p-d .. p+d of its contig and strand, or is below the threshold (dataload.py:244).
struct IngestSlots {
    int slot_of_mod[NM_MAX_MOD_CODES];        // -1: mod code not scored
    uint32_t *planes[NM_MAX_MOD_SLOTS][6];    // M U MP UP MM UM per slot
    uint32_t can_l[NM_MAX_MOD_SLOTS];         // canonical base C (1) or A (0)
};

__global__ void ingest_decide_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                     const uint32_t *__restrict__ contig_chunk, const unsigned long long *__restrict__ dense_plus,
                                     const unsigned long long *__restrict__ dense_minus, int adjacency, double meth_thr,
                                     double low, double high, IngestSlots sl, const uint32_t *__restrict__ H,
                                     const uint32_t *__restrict__ L, const uint32_t *__restrict__ V,
                                     unsigned int *kept /*[contig][mod]*/, unsigned long long *n_kept,
                                     unsigned long long *conf_count, uint32_t *conf_contig, uint32_t *conf_pos,
                                     uint8_t *conf_strand, int8_t *conf_mod, uint64_t conf_cap, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.n) return;
    uint32_t c;
    bool plus;
    if (!ingest_row_alive(r, i, min_cov, ok, &c, &plus)) return;
    const double f = r.frac[i];
    const uint64_t g = (uint64_t)contig_chunk[c] * CHUNK_BP + r.position[i];
    if (!(f < meth_thr)) {
        const unsigned long long *d = plus ? dense_plus : dense_minus;
        unsigned long long mx = 0;
        for (int k = -adjacency; k <= adjacency; ++k) mx = max(mx, d[g + k]);   // >= 64 zero positions around every contig
        if (mx != (unsigned long long)__double_as_longlong(f)) return;
    }
    const int m = r.mod[i];
    atomicAdd(kept + (size_t)c * NM_MAX_MOD_CODES + m, 1u);
    atomicAdd(n_kept, 1ull);
    const int slot = sl.slot_of_mod[m];
    if (slot < 0) return;
    const bool meth = f >= high, non = f <= low;
    if (meth) {
        const unsigned long long at = atomicAdd(conf_count, 1ull);
        if (at < conf_cap) {
            conf_contig[at] = c;
            conf_pos[at] = r.position[i];
            conf_strand[at] = plus ? '+' : '-';
            conf_mod[at] = (int8_t)m;
        }
    }
    if (!meth && !non) return;
    const size_t w = g >> 5;
    const uint32_t bit = 1u << (g & 31);
    uint32_t *const *pl = sl.planes[slot];
    uint32_t *gen = plus ? (meth ? pl[2] : pl[3]) : (meth ? pl[4] : pl[5]);
    const uint32_t old = atomicOr(gen + w, bit);
    if (old & bit) atomicOr(err, 4u);
    const uint32_t h = (H[w] & bit) != 0, l = (L[w] & bit) != 0, v = (V[w] & bit) != 0;
    const uint32_t want_h = plus ? 0u : 1u;
    if (v && h == want_h && l == sl.can_l[slot]) atomicOr((meth ? pl[0] : pl[1]) + w, bit);
}

// ------------------------------------------------------------------------------------------------------
// Window engine: the methylated-site windows of every (bin, mod type) search stay on the device as bit planes over
// WINDOWS (bit i of a plane = window i), so that the per-expansion work of the search — filter_sequence_matches
// (seq.py:499-524) + DNAarray.pssm (seq.py:526-537) — is the same AND / popcount pattern as the genome scan.
//   per task: plane1[col][b] = windows whose base at column col is exactly b (b in A, C, G, T),
//             planeN[col]   = windows with N there (one-hot 1111: counts for all four rows, matches only '.'),
//             alive         = windows not yet removed by an accepted / dead-end motif (find_motifs_bin.py:801-806).
// ------------------------------------------------------------------------------------------------------
constexpr int WIN_MAX_W = 64;                // window width limit (reference default 41)

struct WinTask {
    uint64_t plane_off;     // into the plane pool (words): [col][5][nw]
    uint64_t alive_off;     // into the alive pool (words): [nw]
    uint32_t n, nw, width, pad;
};

// windows given as base-set bytes [n][width] (bit0 A, bit1 C, bit2 G, bit3 T, 15 = N): one thread per (word, col)
__global__ void win_pack_kernel(WinTask t, const uint8_t *__restrict__ sets, uint32_t *__restrict__ planes,
                                uint32_t *__restrict__ alive) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;     // word of 32 windows
    const uint32_t col = blockIdx.y;
    if (w >= t.nw) return;
    uint32_t pa = 0, pc = 0, pg = 0, pt = 0, pn = 0, al = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint64_t i = (uint64_t)w * 32 + b;
        if (i >= t.n) break;
        const uint32_t m = sets[i * t.width + col];
        al |= 1u << b;
        if (m == 1) pa |= 1u << b;
        else if (m == 2) pc |= 1u << b;
        else if (m == 4) pg |= 1u << b;
        else if (m == 8) pt |= 1u << b;
        else pn |= 1u << b;
    }
    uint32_t *p = planes + t.plane_off + (uint64_t)col * 5 * t.nw + w;
    p[0] = pa; p[(uint64_t)t.nw] = pc; p[2ull * t.nw] = pg; p[3ull * t.nw] = pt; p[4ull * t.nw] = pn;
    if (col == 0) alive[t.alive_off + w] = al;
}

// One request = (task, motif as one base-set byte per column, kind).  kind 0: PSSM counts of the alive windows that
// match the motif (every window base must be inside the motif's set at that column; an N window only matches '.');
// kind 1: remove the matching windows from alive and report alive counts before / after.
// out[req] = { n_active | before, 0 | after, counts[4 rows A,T,G,C][width] }  (int32)
__global__ __launch_bounds__(256) void win_request_kernel(const WinTask *__restrict__ tasks, uint32_t n_req,
                                                          const uint32_t *__restrict__ req_task,
                                                          const uint8_t *__restrict__ req_kind,
                                                          const uint8_t *__restrict__ req_sets /*[n_req][WIN_MAX_W]*/,
                                                          const uint32_t *__restrict__ planes, uint32_t *alive,
                                                          int *__restrict__ out, uint32_t out_stride) {
    __shared__ int cnt[2 + 4 * WIN_MAX_W];
    __shared__ uint8_t mset[WIN_MAX_W];
    const uint32_t r = blockIdx.x;
    const WinTask t = tasks[req_task[r]];
    const uint32_t kind = req_kind[r];
    for (uint32_t i = threadIdx.x; i < 2 + 4 * WIN_MAX_W; i += blockDim.x) cnt[i] = 0;
    if (threadIdx.x < WIN_MAX_W) mset[threadIdx.x] = threadIdx.x < t.width ? req_sets[(size_t)r * WIN_MAX_W + threadIdx.x] : 15;
    __syncthreads();
    const uint32_t *pl = planes + t.plane_off;
    uint32_t *al = alive + t.alive_off;
    // word slices of this request are spread over gridDim.y workgroups
    for (uint32_t w = blockIdx.y * blockDim.x + threadIdx.x; w < t.nw; w += gridDim.y * blockDim.x) {
        uint32_t match = 0xFFFFFFFFu;
        for (uint32_t col = 0; col < t.width; ++col) {
            const uint32_t m = mset[col];
            if (m == 15) continue;
            const uint32_t *p = pl + (uint64_t)col * 5 * t.nw + w;
            uint32_t ok = 0;
            if (m & 1) ok |= p[0];
            if (m & 2) ok |= p[(uint64_t)t.nw];
            if (m & 4) ok |= p[2ull * t.nw];
            if (m & 8) ok |= p[3ull * t.nw];
            match &= ok;
        }
        const uint32_t a = al[w];
        if (kind == 1) {
            const uint32_t na = a & ~match;
            al[w] = na;
            atomicAdd(&cnt[0], __popc(a));
            atomicAdd(&cnt[1], __popc(na));
            continue;
        }
        const uint32_t active = a & match;
        if (!active) continue;
        atomicAdd(&cnt[0], __popc(active));
        for (uint32_t col = 0; col < t.width; ++col) {
            const uint32_t *p = pl + (uint64_t)col * 5 * t.nw + w;
            const int n_any = __popc(active & p[4ull * t.nw]);
            const int ca = __popc(active & p[0]) + n_any, cc = __popc(active & p[(uint64_t)t.nw]) + n_any;
            const int cg = __popc(active & p[2ull * t.nw]) + n_any, ct = __popc(active & p[3ull * t.nw]) + n_any;
            if (ca) atomicAdd(&cnt[2 + 0 * WIN_MAX_W + col], ca);      // row order A, T, G, C (constants.py:1)
            if (ct) atomicAdd(&cnt[2 + 1 * WIN_MAX_W + col], ct);
            if (cg) atomicAdd(&cnt[2 + 2 * WIN_MAX_W + col], cg);
            if (cc) atomicAdd(&cnt[2 + 3 * WIN_MAX_W + col], cc);
        }
    }
    __syncthreads();
    int *o = out + (size_t)r * out_stride;
    for (uint32_t i = threadIdx.x; i < 2 + 4 * WIN_MAX_W; i += blockDim.x)
        if (cnt[i]) atomicAdd(&o[i], cnt[i]);
}

// ------------------------------------------------------------------------------------------------------
// Window extraction on the device (find_motifs_bin.py:625-686): gather of the methylation windows from the resident
// sequence planes, and the background sample (seq.py:202-225) as "k-th position whose base is X" through a rank
// table (popcount prefix per 512-bp block, per contig).
// ------------------------------------------------------------------------------------------------------
constexpr int RANK_BLOCK_WORDS = 16;                          // 512 bp per rank entry
constexpr int RANK_PER_CHUNK = CHUNK_WORDS / RANK_BLOCK_WORDS;

__device__ __forceinline__ uint32_t base_word(uint32_t h, uint32_t l, uint32_t v, int b) {   // b: 0 A, 1 C, 2 G, 3 T
    const uint32_t hh = (b >= 2) ? h : ~h;                    // A=00 C=01 G=11 T=10 as (H, L)
    const uint32_t ll = (b == 1 || b == 2) ? l : ~l;
    return v & hh & ll;
}

// bits [g, g+n) of a plane as the low n bits of a uint64, n <= 64; touches only the words that hold them
__device__ __forceinline__ uint64_t plane_field(const uint32_t *__restrict__ P, uint64_t g, uint32_t n) {
    const size_t w0 = (size_t)(g >> 5);
    const uint32_t sh = (uint32_t)(g & 31);
    uint64_t x = P[w0];
    if (sh + n > 32) x |= (uint64_t)P[w0 + 1] << 32;
    x >>= sh;
    if (sh + n > 64) x |= (uint64_t)P[w0 + 2] << (64 - sh);
    return n >= 64 ? x : (x & ((1ull << n) - 1));
}

// one wave per contig: rank[block] = number of positions with base b in the contig before the block
// (b < 0: the set bits of `plane`, e.g. a slot's methylated-row plane, instead of a base)
__global__ __launch_bounds__(64) void rank_build_kernel(Planes s, const uint32_t *__restrict__ plane,
                                                        const uint32_t *__restrict__ contig_chunk,
                                                        const uint64_t *__restrict__ contig_len, int b,
                                                        uint32_t *__restrict__ rank, uint64_t *__restrict__ total) {
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    const uint32_t c0 = contig_chunk[ci];
    const uint32_t nblk = (uint32_t)((contig_len[ci] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < nblk; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t cnt = 0;
        if (j < nblk) {
            const size_t w = (size_t)c0 * CHUNK_WORDS + (size_t)j * RANK_BLOCK_WORDS;
            for (int k = 0; k < RANK_BLOCK_WORDS; ++k)
                cnt += __popc(b < 0 ? plane[w + k] : base_word(s.H[w + k], s.L[w + k], s.V[w + k], b));
        }
        uint32_t x = cnt;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if ((int)lane >= d) x += y;
        }
        if (j < nblk) rank[(size_t)c0 * RANK_PER_CHUNK + j] = carry + x - cnt;
        carry += __shfl(x, 63);
    }
    if (lane == 0) total[ci] = carry;
}

// n_valid[contig] = positions p in [pad, len - pad) with base b
__global__ void base_count_kernel(Planes s, const uint32_t *__restrict__ contig_chunk, const uint64_t *__restrict__ contig_len,
                                  uint32_t n_contigs, int b, uint32_t pad, const uint64_t *__restrict__ total,
                                  uint64_t *__restrict__ out) {
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_contigs) return;
    const uint64_t len = contig_len[ci], g0 = (uint64_t)contig_chunk[ci] * CHUNK_BP;
    if (len < 2ull * pad + 1) { out[ci] = 0; return; }
    uint64_t n = total[ci];
    if (pad) {
        const uint64_t hh = plane_field(s.H, g0, pad), hl = plane_field(s.L, g0, pad), hv = plane_field(s.V, g0, pad);
        const uint64_t th = plane_field(s.H, g0 + len - pad, pad), tl = plane_field(s.L, g0 + len - pad, pad),
                       tv = plane_field(s.V, g0 + len - pad, pad);
        const uint64_t head = hv & ((b >= 2) ? hh : ~hh) & ((b == 1 || b == 2) ? hl : ~hl);
        const uint64_t tail = tv & ((b >= 2) ? th : ~th) & ((b == 1 || b == 2) ? tl : ~tl);
        n -= __popcll(head) + __popcll(tail);
    }
    out[ci] = n;
}

// Global bit index of the k-th (0-based, ascending) set bit of a contig in `plane` (b < 0) or among the positions
// with base b; rk = the contig's slice of the rank table.  ~0 when k is beyond the contig's set bits.
__device__ __forceinline__ uint64_t select_kth(const Planes &s, const uint32_t *__restrict__ plane, int b,
                                               const uint32_t *__restrict__ rk, uint32_t c0, uint32_t nblk, uint32_t k) {
    uint32_t lo = 0, hi = nblk - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (rk[mid] <= k) lo = mid; else hi = mid - 1;
    }
    uint32_t r = k - rk[lo];
    const size_t w = (size_t)c0 * CHUNK_WORDS + (size_t)lo * RANK_BLOCK_WORDS;
    for (int q = 0; q < RANK_BLOCK_WORDS; ++q) {
        uint32_t x = b < 0 ? plane[w + q] : base_word(s.H[w + q], s.L[w + q], s.V[w + q], b);
        const uint32_t pc = __popc(x);
        if (r < pc) {
            for (; r; --r) x &= x - 1;
            return (uint64_t)(w + q) * 32 + (uint32_t)__builtin_ctz(x);
        }
        r -= pc;
    }
    return ~0ull;
}

// per contig and strand: rows (set bits of the methylated planes) with pad < pos < len - pad, and how many lie
// before that range — out[contig] = {n_plus, n_minus, head_plus, head_minus}
__global__ void meth_count_kernel(const uint32_t *__restrict__ MP, const uint32_t *__restrict__ MM,
                                  const uint32_t *__restrict__ contig_chunk, const uint64_t *__restrict__ contig_len,
                                  uint32_t n_contigs, uint32_t pad, const uint64_t *__restrict__ total_p,
                                  const uint64_t *__restrict__ total_m, uint64_t *__restrict__ out) {
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_contigs) return;
    const uint64_t len = contig_len[ci], g0 = (uint64_t)contig_chunk[ci] * CHUNK_BP;
    uint64_t *o = out + (size_t)ci * 4;
    if (len < 2ull * pad + 2) { o[0] = o[1] = o[2] = o[3] = 0; return; }
    const uint64_t hp = __popcll(plane_field(MP, g0, pad + 1)), hm = __popcll(plane_field(MM, g0, pad + 1));
    const uint64_t tp = pad ? __popcll(plane_field(MP, g0 + len - pad, pad)) : 0;
    const uint64_t tm = pad ? __popcll(plane_field(MM, g0 + len - pad, pad)) : 0;
    o[0] = total_p[ci] - hp - tp;
    o[1] = total_m[ci] - hm - tm;
    o[2] = hp;
    o[3] = hm;
}

struct BgBlock { uint32_t task, begin_lo, begin_hi, count; };

// Background sample: thread per sample finds the k-th valid centre of its contig (binary search in the rank table,
// scan of one 512-bp block, select in a word), reads the 2*pad+1 letters around it; per column the wave ballots the
// four letters and lane 0 adds the popcounts into the workgroup's LDS table; one workgroup serves one task.
__global__ __launch_bounds__(256) void bg_counts_kernel(Planes s, const uint32_t *__restrict__ rank,
                                                        const uint32_t *__restrict__ contig_chunk,
                                                        const uint64_t *__restrict__ contig_len,
                                                        const BgBlock *__restrict__ blocks,
                                                        const uint32_t *__restrict__ sample_contig,
                                                        const uint32_t *__restrict__ sample_rank, int b, uint32_t pad,
                                                        unsigned long long *__restrict__ out, unsigned int *err) {
    __shared__ uint32_t cnt[4 * WIN_MAX_W];
    const BgBlock blk = blocks[blockIdx.x];
    const uint64_t begin = ((uint64_t)blk.begin_hi << 32) | blk.begin_lo;
    const uint32_t W = 2 * pad + 1, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 4 * WIN_MAX_W; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < blk.count; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        bool on = i < blk.count;
        uint64_t fh = 0, fl = 0, fv = 0;
        if (on) {
            const uint32_t ci = sample_contig[begin + i];
            const uint32_t c0 = contig_chunk[ci];
            const uint64_t g0 = (uint64_t)c0 * CHUNK_BP;
            const uint32_t nblk = (uint32_t)((contig_len[ci] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
            uint32_t k = sample_rank[begin + i];
            if (pad) {
                const uint64_t hh = plane_field(s.H, g0, pad), hl = plane_field(s.L, g0, pad), hv = plane_field(s.V, g0, pad);
                k += __popcll(hv & ((b >= 2) ? hh : ~hh) & ((b == 1 || b == 2) ? hl : ~hl));
            }
            const uint64_t centre = select_kth(s, nullptr, b, rank + (size_t)c0 * RANK_PER_CHUNK, c0, nblk, k);
            if (centre == ~0ull || centre < g0 + pad) {
                atomicOr(err, 4u);                       // rank beyond the contig's valid starts
                on = false;
            } else {
                fh = plane_field(s.H, centre - pad, W);
                fl = plane_field(s.L, centre - pad, W);
                fv = plane_field(s.V, centre - pad, W);
            }
        }
        for (uint32_t col = 0; col < W; ++col) {
            const bool v = on && ((fv >> col) & 1), h = (fh >> col) & 1, l = (fl >> col) & 1;
            const unsigned long long ba = __ballot(v && !h && !l), bt = __ballot(v && h && !l);
            const unsigned long long bg = __ballot(v && h && l), bc = __ballot(v && !h && l);
            if (lane == 0) {                              // rows A, T, G, C (constants.py:1)
                if (ba) atomicAdd(&cnt[0 * WIN_MAX_W + col], (uint32_t)__popcll(ba));
                if (bt) atomicAdd(&cnt[1 * WIN_MAX_W + col], (uint32_t)__popcll(bt));
                if (bg) atomicAdd(&cnt[2 * WIN_MAX_W + col], (uint32_t)__popcll(bg));
                if (bc) atomicAdd(&cnt[3 * WIN_MAX_W + col], (uint32_t)__popcll(bc));
            }
        }
    }
    __syncthreads();
    unsigned long long *o = out + (size_t)blk.task * 4 * WIN_MAX_W;
    for (uint32_t i = threadIdx.x; i < 4 * WIN_MAX_W; i += blockDim.x)
        if (cnt[i]) atomicAdd(&o[i], (unsigned long long)cnt[i]);
}

// Methylation windows of one task from the sequence planes: a wave packs 64 windows; per column it ballots the five
// window planes (A, C, G, T, N) and lanes 0 / 1 store the two words.  Minus rows are reverse-complemented
// (complement = flip H in the (H, L) code; N stays N).
__device__ __forceinline__ void pack_windows(const WinTask &t, const Planes &s, uint32_t wave, uint32_t lane, bool on,
                                             uint64_t centre, bool minus, uint32_t pad, uint32_t *__restrict__ planes,
                                             uint32_t *__restrict__ alive) {
    const uint32_t W = t.width;
    uint64_t fh = 0, fl = 0, fv = 0;
    if (on) {
        const uint64_t g = centre - pad;
        fh = plane_field(s.H, g, W);
        fl = plane_field(s.L, g, W);
        fv = plane_field(s.V, g, W);
        if (minus) {
            fh = __brevll(fh) >> (64 - W);
            fl = __brevll(fl) >> (64 - W);
            fv = __brevll(fv) >> (64 - W);
            fh = ~fh & fv;
        }
    }
    const uint32_t w = wave * 2 + lane;                   // word written by lanes 0 and 1
    const bool writer = lane < 2 && w < t.nw;
    const uint32_t shift = 32 * (lane & 1);
    for (uint32_t col = 0; col < W; ++col) {
        const bool v = (fv >> col) & 1, h = (fh >> col) & 1, l = (fl >> col) & 1;
        const unsigned long long ba = __ballot(on && v && !h && !l), bc = __ballot(on && v && !h && l);
        const unsigned long long bg = __ballot(on && v && h && l), bt = __ballot(on && v && h && !l);
        const unsigned long long bn = __ballot(on && !v);
        if (writer) {
            uint32_t *p = planes + t.plane_off + (uint64_t)col * 5 * t.nw + w;
            p[0] = (uint32_t)(ba >> shift);
            p[(uint64_t)t.nw] = (uint32_t)(bc >> shift);
            p[2ull * t.nw] = (uint32_t)(bg >> shift);
            p[3ull * t.nw] = (uint32_t)(bt >> shift);
            p[4ull * t.nw] = (uint32_t)(bn >> shift);
        }
    }
    const unsigned long long bal = __ballot(on);
    if (writer) alive[t.alive_off + w] = (uint32_t)(bal >> shift);
}

// windows around explicit rows (global bit index of the centre, strand flag)
__global__ __launch_bounds__(256) void win_gather_kernel(WinTask t, Planes s, const uint64_t *__restrict__ row_centre,
                                                         const uint8_t *__restrict__ row_minus, uint32_t pad,
                                                         uint32_t *__restrict__ planes, uint32_t *__restrict__ alive) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t i = (uint64_t)wave * 64 + lane;
    if ((uint64_t)wave * 64 >= t.n) return;
    const bool on = i < t.n;
    pack_windows(t, s, wave, lane, on, on ? row_centre[i] : 0, on && row_minus[i], pad, planes, alive);
}

// windows around the methylated rows of a list of contigs, read from the slot's methylated-row planes: window i
// belongs to the segment (contig, strand) with the largest dst_start <= i and is that segment's
// (i - dst_start + head)-th set bit (head = rows before the edge-filtered range)
struct WinSegment { uint32_t dst_start, contig, minus, head; };

__global__ __launch_bounds__(256) void win_gather_contigs_kernel(WinTask t, Planes s, const uint32_t *__restrict__ MP,
                                                                 const uint32_t *__restrict__ MM,
                                                                 const uint32_t *__restrict__ rank_p,
                                                                 const uint32_t *__restrict__ rank_m,
                                                                 const uint32_t *__restrict__ contig_chunk,
                                                                 const uint64_t *__restrict__ contig_len,
                                                                 const WinSegment *__restrict__ seg, uint32_t n_seg,
                                                                 uint32_t pad, uint32_t *__restrict__ planes,
                                                                 uint32_t *__restrict__ alive, unsigned int *err) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t i = (uint64_t)wave * 64 + lane;
    if ((uint64_t)wave * 64 >= t.n) return;
    bool on = i < t.n, minus = false;
    uint64_t centre = 0;
    if (on) {
        uint32_t lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (seg[mid].dst_start <= (uint32_t)i) lo = mid; else hi = mid - 1;
        }
        const WinSegment sg = seg[lo];
        minus = sg.minus != 0;
        const uint32_t c0 = contig_chunk[sg.contig];
        const uint32_t nblk = (uint32_t)((contig_len[sg.contig] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
        centre = select_kth(s, minus ? MM : MP, -1, (minus ? rank_m : rank_p) + (size_t)c0 * RANK_PER_CHUNK, c0, nblk,
                            (uint32_t)i - sg.dst_start + sg.head);
        if (centre == ~0ull) {
            atomicOr(err, 8u);
            on = false;
        }
    }
    pack_windows(t, s, wave, lane, on, centre, minus, pad, planes, alive);
}

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// Read-only tables are addressed through the constant address space so that wave-uniform reads become scalar
// loads (s_load_dwordx16 straight into SGPRs) instead of vector loads + v_readfirstlane.
typedef const uint32_t __attribute__((address_space(4))) *cu32p;
typedef const uint8_t __attribute__((address_space(4))) *cu8p;

// Eight derived planes for the T main words of this lane plus GN halo words left and GP right.
template <int GN, int GP>
struct Tile {
    static constexpr int NW = T_WORDS + GN + GP;
    uint32_t w[8][NW];

    __device__ __forceinline__ void load(const Planes &s, uint32_t chunk, int lane, bool need_v) {
        const size_t base = (size_t)chunk * CHUNK_WORDS + (size_t)lane * T_WORDS;
        uint32_t h[NW], l[NW], v[NW];
        const uint4 h4 = *reinterpret_cast<const uint4 *>(s.H + base);
        const uint4 l4 = *reinterpret_cast<const uint4 *>(s.L + base);
        h[GN + 0] = h4.x; h[GN + 1] = h4.y; h[GN + 2] = h4.z; h[GN + 3] = h4.w;
        l[GN + 0] = l4.x; l[GN + 1] = l4.y; l[GN + 2] = l4.z; l[GN + 3] = l4.w;
#pragma unroll
        for (int j = 0; j < GN; ++j) { h[j] = s.H[base - GN + j]; l[j] = s.L[base - GN + j]; }
#pragma unroll
        for (int j = 0; j < GP; ++j) { h[GN + T_WORDS + j] = s.H[base + T_WORDS + j]; l[GN + T_WORDS + j] = s.L[base + T_WORDS + j]; }
        if (need_v) {                                                // wave-uniform
            const uint4 v4 = *reinterpret_cast<const uint4 *>(s.V + base);
            v[GN + 0] = v4.x; v[GN + 1] = v4.y; v[GN + 2] = v4.z; v[GN + 3] = v4.w;
#pragma unroll
            for (int j = 0; j < GN; ++j) v[j] = s.V[base - GN + j];
#pragma unroll
            for (int j = 0; j < GP; ++j) v[GN + T_WORDS + j] = s.V[base + T_WORDS + j];
        } else {
#pragma unroll
            for (int j = 0; j < NW; ++j) v[j] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const uint32_t hh = h[j], ll = l[j], vv = v[j];
            w[0][j] = vv & ~hh & ~ll;          // A = 00
            w[1][j] = vv & ~hh & ll;           // C = 01
            w[2][j] = vv & hh & ll;            // G = 11
            w[3][j] = vv & hh & ~ll;           // T = 10
            w[4][j] = vv & (hh | ll);          // valid, not A
            w[5][j] = vv & (hh | ~ll);         // valid, not C
            w[6][j] = vv & ~(hh & ll);         // valid, not G
            w[7][j] = vv & (~hh | ll);         // valid, not T
        }
    }
};

// One strand's constraint masks for the offset groups this kernel variant supports: m[(gi - (2 - GN)) * 8 + p].
template <int GN, int GP>
struct StrandMasks {
    static constexpr int N = (GN + GP) * 8;
    uint32_t m[N];
    __device__ __forceinline__ void load(cu32p prog_strand) {
#pragma unroll
        for (int i = 0; i < N; ++i) m[i] = prog_strand[i];
    }
};

// acc[t] &= plane p at offset d (d = 32 g + r) for every constraint bit of one strand's program.  The caller
// initialises acc (all ones, or the plane of the modified base: that constraint sits at offset 0 and needs no
// alignbit).  Two constraints of the same (word-group, plane) class are folded into one 3-input AND (v_bitop3).
template <int GN, int GP>
__device__ __forceinline__ void eval_strand(const StrandMasks<GN, GP> &sm, const Tile<GN, GP> &tile,
                                            uint32_t (&acc)[T_WORDS]) {
#pragma unroll
    for (int g = 0; g < GN + GP; ++g) {            // g = gi - (2 - GN): word pair (t + g, t + g + 1)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            uint32_t m = sm.m[g * 8 + p];
            while (m) {
                const uint32_t r = __builtin_ctz(m);
                m &= m - 1;
                if (false) {
                } else {
#pragma unroll
                    for (int t = 0; t < T_WORDS; ++t) acc[t] &= alignbit(tile.w[p][t + g + 1], tile.w[p][t + g], r);
                }
            }
        }
    }
}

// The candidates [k0, k0 + nb) of one mod-type slot against the tile this wave holds: match masks, site counts,
// per-lane counts into LDS rows lds_row0 + k.  CAN: canonical base of the slot, 0 = A (reverse-strand sites sit on T),
// 1 = C (reverse on G).
// State words of one slot for this lane's T words: compact = {M, U}, general = {MP, UP, MM, UM}.
template <bool COMPACT>
struct StateWords {
    uint32_t s[COMPACT ? 2 : 4][T_WORDS];
    __device__ __forceinline__ void load(const StatePlanes &st, size_t base) {
        const uint32_t *src[4] = {COMPACT ? st.M : st.MP, COMPACT ? st.U : st.UP, st.MM, st.UM};
#pragma unroll
        for (int i = 0; i < (COMPACT ? 2 : 4); ++i) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src[i] + base);
            s[i][0] = v.x; s[i][1] = v.y; s[i][2] = v.z; s[i][3] = v.w;
        }
    }
};

template <int GN, int GP, bool COMPACT, int CAN>
__device__ __forceinline__ void score_candidates(const ScoreArgs &a, const Tile<GN, GP> &tile, const StateWords<COMPACT> &sw,
                                                 uint32_t k0, uint32_t nb, uint32_t *lds_acc, uint32_t lds_row0, int lane) {
    constexpr int PF = CAN == 0 ? 0 : 1;   // plane of the canonical base: A or C
    constexpr int PR = CAN == 0 ? 3 : 2;   // plane of its complement:     T or G
    for (uint32_t k = 0; k < nb; ++k) {
        cu32p prog = (cu32p)(a.programs + (size_t)(k0 + k) * (2 * StrandMasks<GN, GP>::N));
        StrandMasks<GN, GP> mf, mr;
        mf.load(prog);
        mr.load(prog + StrandMasks<GN, GP>::N);
        uint32_t accf[T_WORDS], accr[T_WORDS];
#pragma unroll
        for (int t = 0; t < T_WORDS; ++t) {
            // compact batches: every candidate has the canonical literal at its modified position, the host leaves
            // that constraint out of the program and it becomes the accumulator's initial value
            accf[t] = COMPACT ? tile.w[PF][t + GN] : 0xFFFFFFFFu;
            accr[t] = COMPACT ? tile.w[PR][t + GN] : 0xFFFFFFFFu;
        }
        eval_strand<GN, GP>(mf, tile, accf);
        eval_strand<GN, GP>(mr, tile, accr);
        uint32_t n_mod = 0, n_non = 0;
#pragma unroll
        for (int t = 0; t < T_WORDS; ++t) {
            if (COMPACT) {
                const uint32_t sites = accf[t] | accr[t];
                n_mod += __popc(sites & sw.s[0][t]);
                n_non += __popc(sites & sw.s[1][t]);
            } else {
                n_mod += __popc(accf[t] & sw.s[0][t]) + __popc(accr[t] & sw.s[COMPACT ? 0 : 2][t]);
                n_non += __popc(accf[t] & sw.s[1][t]) + __popc(accr[t] & sw.s[COMPACT ? 1 : 3][t]);
            }
        }
        atomicAdd(&lds_acc[((lds_row0 + k) * 2 + 0) * 64 + lane], n_mod);
        atomicAdd(&lds_acc[((lds_row0 + k) * 2 + 1) * 64 + lane], n_non);
    }
}

// NS = mod-type slots fused into one workgroup: with NS > 1 a tile's sequence planes are loaded and expanded once
// and serve the candidates of all NS slots (each slot brings its own M / U planes); with NS = 1 the slot comes
// from blockIdx.y (batches that touch more than two classifications).  A pass handles up to BMAX / NS candidates
// per slot; LDS rows are [slot j][candidate k].
template <int GN, int GP, bool COMPACT, int NS>
__global__ __launch_bounds__(256, (GN + GP > 2 ? 2 : 4)) void score_kernel(ScoreArgs a) {
    __shared__ uint32_t lds_acc[BMAX * 2 * 64];
    constexpr uint32_t H = BMAX / NS;
    // XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), give every XCD a contiguous run
    // of segments so candidate programs and counters of one bin stay in one L2.
    const uint32_t nb = gridDim.x;
    const uint32_t per = (nb + 7) / 8;
    uint32_t seg = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (seg >= a.n_segments) return;
    uint4 sg = a.segments[seg];
    sg.x = __builtin_amdgcn_readfirstlane(sg.x);   // everything below is wave-uniform: keep it in SGPRs
    sg.y = __builtin_amdgcn_readfirstlane(sg.y);
    sg.z = __builtin_amdgcn_readfirstlane(sg.z);
    uint2 range[NS];
    uint32_t most = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const uint32_t slot_i = NS == 1 ? blockIdx.y : (uint32_t)j;
        range[j] = a.cand_range[(size_t)slot_i * a.n_bins + sg.z];
        range[j].x = __builtin_amdgcn_readfirstlane(range[j].x);
        range[j].y = __builtin_amdgcn_readfirstlane(range[j].y);
        most = max(most, range[j].y);
    }
    if (most == 0) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably uniform: chunk indices stay scalar
    // per-slot facts are read from the kernel arguments once, not per chunk
    StatePlanes stp[NS];
    bool is_c[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const uint32_t slot = a.active_slot[NS == 1 ? blockIdx.y : (uint32_t)j];
        stp[j] = a.st[slot];
        is_c[j] = a.slot_is_c[slot] != 0;
    }

    for (uint32_t pass0 = 0; pass0 < most; pass0 += H) {
        for (uint32_t i = threadIdx.x; i < BMAX * 128; i += 256) lds_acc[i] = 0;
        __syncthreads();
        for (uint32_t ck = wave; ck < sg.y; ck += 4) {
            const uint32_t chunk = sg.x + ck;
            const bool need_v = ((cu8p)a.seq.needs_v)[chunk] != 0;          // scalar load, issued first
            const size_t base = (size_t)chunk * CHUNK_WORDS + (size_t)lane * T_WORDS;
            // all global loads of this chunk are issued back to back (state planes first, then the sequence planes)
            // so that one memory latency covers them; the expansion into the eight derived planes follows
            StateWords<COMPACT> sw[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) sw[j].load(stp[j], base);          // unconditional: no control flow between loads
            Tile<GN, GP> tile;
            tile.load(a.seq, chunk, lane, need_v);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (range[j].y <= pass0) continue;                       // wave-uniform
                const uint32_t nbj = min(H, range[j].y - pass0);
                if (COMPACT && is_c[j])
                    score_candidates<GN, GP, COMPACT, 1>(a, tile, sw[j], range[j].x + pass0, nbj, lds_acc, j * H, lane);
                else
                    score_candidates<GN, GP, COMPACT, 0>(a, tile, sw[j], range[j].x + pass0, nbj, lds_acc, j * H, lane);
            }
        }
        __syncthreads();
        // 4 threads per counter, 16 lane-slots each, then a 4-lane butterfly; one 64-bit atomic per counter
        for (uint32_t idx = threadIdx.x; idx < BMAX * 8; idx += 256) {
            const uint32_t i = idx >> 2, q = idx & 3;               // i = counter row: (j * H + k) * 2 + which
            const uint32_t j = (i >> 1) / H, k = (i >> 1) % H;
            uint32_t s = 0;
#pragma unroll 4
            for (int jj = 0; jj < 16; ++jj) s += lds_acc[i * 64 + q * 16 + jj];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (q == 0 && s) {
                uint2 rj = range[0];
#pragma unroll
                for (int t = 1; t < NS; ++t)
                    if (j == (uint32_t)t) rj = range[t];
                const uint32_t orig = a.orig_index[rj.x + pass0 + k];
                atomicAdd(a.out + (size_t)orig * 2 + (i & 1), (unsigned long long)s);
            }
        }
        __syncthreads();
    }
}

// One candidate record as the host stages it (sorted by mod-type slot, then bin).
struct CandRec {
    uint32_t mask_off;   // into the staged mask bytes
    uint32_t orig;       // caller's index of this candidate
    uint8_t len, modpos, slot, pad;
};

// Compile the staged candidates into constraint programs ON THE DEVICE: one thread per candidate.  A literal is one
// constraint on an is-X plane, a 3-set one on a valid-not-X plane, a 2-set two of those; the reverse strand takes
// the complemented set at the negated offset (motif.py:260-266).  Program layout: [strand][word-group][plane] with
// the word-groups the launched variant reads (narrow: groups 1..2, wide: 0..3); bit r of a word = offset 32 g + r.
// fold_modpos: the modified position's own constraint is left out (compact batches start the accumulator from the
// canonical plane instead).
__global__ void compile_kernel(uint32_t n_prog, const CandRec *__restrict__ rec, const uint8_t *__restrict__ masks,
                               uint32_t *__restrict__ programs, int wide, int fold_modpos) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_prog) return;
    const int groups = wide ? 4 : 2, g0 = wide ? 0 : 1;
    const int pdw = 2 * groups * 8;
    uint32_t *prog = programs + (size_t)k * pdw;
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
            uint32_t *row = prog + (strand * groups + g) * 8;
            if (__popc(set) == 1) {
                row[__ffs(set) - 1] |= bit;
            } else {
                uint32_t missing = (~set) & 15u;
                while (missing) {
                    row[4 + __ffs(missing) - 1] |= bit;
                    missing &= missing - 1;
                }
            }
        }
    }
}

// Site masks of one candidate over the chunks of one contig (general planes) for nm_hit_positions.
template <int GN, int GP>
__global__ __launch_bounds__(256) void hits_kernel(Planes seq, StatePlanes st, uint32_t chunk0, uint32_t n_chunks,
                                                   const uint32_t *__restrict__ prog, int which,
                                                   uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ck = blockIdx.x * 4 + wave;
    if (ck >= n_chunks) return;
    const uint32_t chunk = chunk0 + ck;
    Tile<GN, GP> tile;
    tile.load(seq, chunk, lane, seq.needs_v[chunk] != 0);
    uint32_t acc[T_WORDS];
#pragma unroll
    for (int t = 0; t < T_WORDS; ++t) acc[t] = 0xFFFFFFFFu;
    StrandMasks<GN, GP> sm;
    sm.load((cu32p)(prog + (which >= 2 ? StrandMasks<GN, GP>::N : 0)));
    eval_strand<GN, GP>(sm, tile, acc);
    const uint32_t *plane = which == 0 ? st.MP : which == 1 ? st.UP : which == 2 ? st.MM : st.UM;
    const size_t base = (size_t)chunk * CHUNK_WORDS + (size_t)lane * T_WORDS;
#pragma unroll
    for (int t = 0; t < T_WORDS; ++t) out[(size_t)ck * CHUNK_WORDS + lane * T_WORDS + t] = acc[t] & plane[base + t];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
struct ModSlot {
    bool present = false;
    uint8_t canonical = 0;          // 'A' or 'C'
    double low = 0.3, high = 0.7;
    uint32_t *planes[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // M U MP UP MM UM
    uint64_t n_rows = 0;
    // window extraction: rank tables over the methylated-row planes MP / MM and the per-contig row counts for the
    // edge padding `meth_pad` ({n_plus, n_minus, head_plus, head_minus} per contig); dropped when the planes change
    uint32_t *rank[2] = {nullptr, nullptr};
    uint64_t *rank_total[2] = {nullptr, nullptr};
    std::vector<uint64_t> meth_counts;
    uint32_t meth_pad = 0xFFFFFFFFu;
};

static void drop_slot_ranks(ModSlot &ms) {
    for (int k = 0; k < 2; ++k) {
        if (ms.rank[k]) (void)hipFree(ms.rank[k]);
        if (ms.rank_total[k]) (void)hipFree(ms.rank_total[k]);
        ms.rank[k] = nullptr;
        ms.rank_total[k] = nullptr;
    }
    ms.meth_counts.clear();
    ms.meth_pad = 0xFFFFFFFFu;
}

struct nm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    std::vector<uint32_t> bucket;                     // per-call host scratch, kept to avoid reallocation
    // window engine
    std::vector<WinTask> win_tasks;
    uint32_t *d_win_planes = nullptr, *d_win_alive = nullptr;
    uint64_t win_planes_cap = 0, win_alive_cap = 0, win_planes_used = 0, win_alive_used = 0;
    WinTask *d_win_tasks = nullptr;
    size_t d_win_tasks_cap = 0;
    bool win_tasks_dirty = false;
    // results of the last nm_ingest_pileup
    std::vector<uint32_t> ing_kept;                   // kept rows per (contig, mod code)
    uint32_t *d_ing_contig = nullptr, *d_ing_pos = nullptr;   // confident rows stay on the device until asked for
    uint8_t *d_ing_strand = nullptr;
    int8_t *d_ing_mod = nullptr;
    uint64_t ing_nconf = 0;
    uint32_t *d_programs = nullptr;                   // compiled constraint programs of the current batch
    size_t prog_cap_dw = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // per-launch event pairs since the last nm_timing_reset (bounded pool, summed lazily: no sync per launch)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    bool ev_collect = false;
    // assembly
    uint32_t n_contigs = 0, n_bins = 0, n_chunks = 0;
    uint64_t total_bp = 0;
    std::vector<uint64_t> contig_len;
    std::vector<uint32_t> contig_chunk, contig_bin, contig_nchunks;
    std::vector<uint32_t> bin_chunk0, bin_nchunks;
    uint32_t *dH = nullptr, *dL = nullptr, *dV = nullptr;
    uint8_t *d_needs_v = nullptr;
    uint32_t *d_contig_chunk = nullptr;
    uint64_t *d_contig_len = nullptr;
    uint4 *d_segments = nullptr;
    uint32_t n_segments = 0;
    ModSlot slots[NM_MAX_MOD_SLOTS];
    // per-call staging: ring of two (device, pinned host) buffer pairs so that compiling the next batch on the
    // host overlaps the previous launch; `busy` marks the last device work that read the pair.
    struct Stage {
        void *d = nullptr, *h = nullptr;
        size_t bytes = 0;
        hipEvent_t busy = nullptr;
        bool pending = false;
    } stage[2];
    int stage_next = 0;
    void *d_stage = nullptr, *h_stage = nullptr;   // the pair acquired by the current call
    Stage *cur_stage = nullptr;
    unsigned long long *d_counts = nullptr;
    size_t counts_cap = 0;
    unsigned int *d_err = nullptr;
    // window extraction: per-base rank tables over the sequence planes (built on first use), other-letter count
    uint32_t *d_rank[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_base_total[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long *d_other = nullptr;
    uint64_t other_letters = 0;
    uint64_t launches = 0, last_wgs = 0, last_compact = 0, last_general = 0;
};

namespace {

size_t plane_words(const nm_ctx *c) { return (size_t)c->n_chunks * CHUNK_WORDS; }

int ensure_stage(nm_ctx *c, size_t bytes) {
    nm_ctx::Stage &st = c->stage[c->stage_next];
    c->stage_next ^= 1;
    if (!st.busy) HIP_TRY(hipEventCreateWithFlags(&st.busy, hipEventDisableTiming));
    if (st.pending) {
        HIP_TRY(hipEventSynchronize(st.busy));
        st.pending = false;
    }
    if (bytes > st.bytes) {
        const size_t nb = std::max(bytes, st.bytes * 2);
        if (st.d) (void)hipFree(st.d);
        if (st.h) (void)hipHostFree(st.h);
        st.d = st.h = nullptr;
        st.bytes = 0;
        HIP_TRY(hipMalloc(&st.d, nb));
        HIP_TRY(hipHostMalloc(&st.h, nb, hipHostMallocDefault));
        st.bytes = nb;
    }
    c->d_stage = st.d;
    c->h_stage = st.h;
    c->cur_stage = &st;
    return NM_OK;
}

int release_stage(nm_ctx *c) {   // call after the last device work that reads the acquired pair was enqueued
    HIP_TRY(hipEventRecord(c->cur_stage->busy, c->stream));
    c->cur_stage->pending = true;
    return NM_OK;
}

inline uint32_t comp_mask(uint32_t m) { return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3); }

// Compile one stripped motif into the per-strand constraint masks.  Layout: prog[strand*32 + gi*8 + plane],
// gi = floor(d/32) + 2, bit r = d mod 32.
int compile_program(const uint8_t *masks, uint32_t len, uint32_t modpos, uint32_t *prog, bool *wide,
                    uint32_t *modpos_mask) {
    if (len == 0 || len > NM_MAX_MOTIF_LEN) return fail(NM_ERANGE, "motif length %u outside 1..%d", len, NM_MAX_MOTIF_LEN);
    if (modpos >= len) return fail(NM_EINVAL, "mod_position %u outside motif of length %u", modpos, len);
    memset(prog, 0, PROG_DW * sizeof(uint32_t));
    bool any = false;
    *wide = false;
    for (uint32_t j = 0; j < len; ++j) {
        const uint32_t m = masks[j] & 15u;
        if (m == 0) return fail(NM_EINVAL, "empty base set at motif position %u", j);
        if (m == 15u) continue;
        any = true;
        if (j == modpos) continue;      // offset 0: folded into the accumulator init, or added by add_modpos_constraint
        for (int strand = 0; strand < 2; ++strand) {
            const int d = strand == 0 ? (int)j - (int)modpos : (int)modpos - (int)j;
            const uint32_t set = strand == 0 ? m : comp_mask(m);
            const int gi = (d >> 5) + 2;                            // arithmetic shift = floor
            const uint32_t r = (uint32_t)d & 31u;
            if (gi < 0 || gi > 3) return fail(NM_ERANGE, "offset %d from the modified base is outside [-64, 63]", d);
            if (gi == 0 || gi == 3) *wide = true;
            uint32_t *row = prog + strand * 32 + gi * 8;
            const int nbits = __builtin_popcount(set);
            if (nbits == 1) {
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
    *modpos_mask = masks[modpos] & 15u;
    return NM_OK;
}

// The constraint of the modified position itself (offset 0 -> word-group 2, shift 0), for programs that run on the
// general path where the accumulator starts from all ones.
void add_modpos_constraint(uint32_t *prog, uint32_t modpos_mask) {
    if (modpos_mask == 15u) return;
    for (int strand = 0; strand < 2; ++strand) {
        const uint32_t set = strand == 0 ? modpos_mask : comp_mask(modpos_mask);
        uint32_t *row = prog + strand * 32 + 2 * 8;
        if (__builtin_popcount(set) == 1) {
            row[__builtin_ctz(set)] |= 1u;
        } else {
            uint32_t missing = (~set) & 15u;
            while (missing) {
                row[4 + __builtin_ctz(missing)] |= 1u;
                missing &= missing - 1;
            }
        }
    }
}

template <int GN, int GP, bool COMPACT>
void launch_score(const ScoreArgs &a, uint32_t gx, uint32_t n_active, bool fuse, hipStream_t s) {
    // two classifications (the usual 6mA + 5mC batch) share one pass over the sequence planes when the batch is
    // light (HBM-bound: a greedy round); heavy batches are VALU-bound and run better as twice as many workgroups
    if (n_active == 2 && fuse) hipLaunchKernelGGL((score_kernel<GN, GP, COMPACT, 2>), dim3(gx, 1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((score_kernel<GN, GP, COMPACT, 1>), dim3(gx, std::max(n_active, 1u)), dim3(256), 0, s, a);
}

int score_impl(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
               const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
               const uint8_t *cand_masks, unsigned long long *d_out, int64_t *h_out) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    if (n_cand == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    // ---- one pass over the candidates: validate, classify the batch, count per (slot, bin) bucket.
    // Candidates whose bin has no contig on this device (multi-GPU shard) contribute zero and get no program.
    const uint32_t n_bins = c->n_bins;
    std::vector<uint32_t> &bucket = c->bucket;          // counting sort by (slot, bin)
    bucket.assign((size_t)NM_MAX_MOD_SLOTS * n_bins + 1, 0);
    bool any_wide = false, all_compact = true;
    uint32_t n_prog = 0;
    uint64_t mask_bytes = 0, mask_lo = ~0ull;   // byte range of cand_masks the resident candidates reference
    bool slot_used[NM_MAX_MOD_SLOTS] = {};
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t slot = cand_mod_slot[k], bin = cand_bin[k];
        if (slot >= NM_MAX_MOD_SLOTS || !c->slots[slot].present)
            return fail(NM_ESTATE, "candidate %u uses mod slot %u with no pileup uploaded", k, slot);
        if (bin >= n_bins) return fail(NM_EINVAL, "candidate %u: bin %u >= n_bins %u", k, bin, n_bins);
        if (c->bin_nchunks[bin] == 0) continue;         // not resident here: the rank that holds the bin validates it
        const uint32_t len = cand_len[k], mp = cand_modpos[k];
        if (len == 0 || len > NM_MAX_MOTIF_LEN) return fail(NM_ERANGE, "candidate %u: motif length %u outside 1..%d", k, len, NM_MAX_MOTIF_LEN);
        if (mp >= len) return fail(NM_EINVAL, "candidate %u: mod_position %u outside motif of length %u", k, mp, len);
        const uint8_t *m = cand_masks + cand_mask_offset[k];
        uint32_t all_and = 15u, any_zero = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const uint32_t v = m[j] & 15u;
            all_and &= v;
            any_zero |= (v == 0);
        }
        if (any_zero) return fail(NM_EINVAL, "candidate %u: empty base set in the motif", k);
        if (all_and == 15u) return fail(NM_EINVAL, "candidate %u: motif has no specified position", k);
        mask_lo = std::min<uint64_t>(mask_lo, cand_mask_offset[k]);
        mask_bytes = std::max<uint64_t>(mask_bytes, (uint64_t)cand_mask_offset[k] + len);
        // offsets relative to the modified base span [-mp, len-1-mp] forward and the mirror image in reverse
        if (mp > 31 || len - 1 - mp > 31) any_wide = true;
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
    const uint32_t pdw = any_wide ? 64u : 32u;
    const size_t rec_bytes = (size_t)n_prog * sizeof(CandRec);
    const size_t off_orig = (rec_bytes + 15) & ~(size_t)15;
    const size_t off_masks = (off_orig + (size_t)n_prog * 4 + 15) & ~(size_t)15;
    const size_t off_range = (off_masks + (mask_bytes - mask_lo) + 15) & ~(size_t)15;
    const size_t range_bytes = (size_t)std::max(n_active, 1u) * n_bins * sizeof(uint2);
    const size_t total = off_range + range_bytes;
    int rc = ensure_stage(c, total);
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage);
    CandRec *h_rec = reinterpret_cast<CandRec *>(hs);
    uint32_t *h_orig = reinterpret_cast<uint32_t *>(hs + off_orig);
    uint2 *h_range = reinterpret_cast<uint2 *>(hs + off_range);
    memcpy(hs + off_masks, cand_masks + mask_lo, mask_bytes - mask_lo);
    memset(h_range, 0, range_bytes);
    uint32_t n_groups = 0;
    // exclusive prefix over the buckets -> first sorted index of each (slot, bin); stable within a bucket
    for (size_t i = 1; i < bucket.size(); ++i) bucket[i] += bucket[i - 1];
    for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) {
        if (slot_to_active[sl] < 0) continue;
        for (uint32_t b = 0; b < n_bins; ++b) {
            const uint32_t lo = bucket[(size_t)sl * n_bins + b], hi = bucket[(size_t)sl * n_bins + b + 1];
            if (hi > lo) {
                h_range[(size_t)slot_to_active[sl] * n_bins + b] = make_uint2(lo, hi - lo);
                n_groups += 1;
            }
        }
    }
    for (uint32_t k = 0; k < n_cand; ++k) {
        const uint32_t slot = cand_mod_slot[k], bin = cand_bin[k];
        if (c->bin_nchunks[bin] == 0) continue;
        const uint32_t at = bucket[(size_t)slot * n_bins + bin]++;
        h_rec[at] = CandRec{(uint32_t)(cand_mask_offset[k] - mask_lo), k, cand_len[k], cand_modpos[k], (uint8_t)slot, 0};
        h_orig[at] = k;
    }
    // device-side program buffer
    const size_t need_dw = (size_t)std::max(n_prog, 1u) * pdw;
    if (c->prog_cap_dw < need_dw) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_programs) (void)hipFree(c->d_programs);
        c->d_programs = nullptr;
        c->prog_cap_dw = 0;
        HIP_TRY(hipMalloc(&c->d_programs, need_dw * 4 * 2));
        c->prog_cap_dw = need_dw * 2;
    }
    // staged tables travel on the copy stream so that the upload of batch k+1 overlaps the kernel of batch k
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, total, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipEventRecord(c->copy_done, c->copy_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->copy_done, 0));
    uint8_t *ds = static_cast<uint8_t *>(c->d_stage);
    if (n_prog) {
        hipLaunchKernelGGL(compile_kernel, dim3((n_prog + 255) / 256), dim3(256), 0, c->stream, n_prog,
                           reinterpret_cast<const CandRec *>(ds), ds + off_masks, c->d_programs, any_wide ? 1 : 0,
                           all_compact ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    // ---- output counters
    unsigned long long *out = d_out;
    if (!out) {
        if (c->counts_cap < n_cand) {
            if (c->d_counts) (void)hipFree(c->d_counts);
            c->d_counts = nullptr;
            c->counts_cap = 0;
            HIP_TRY(hipMalloc(&c->d_counts, (size_t)n_cand * 2 * sizeof(unsigned long long) * 2));
            c->counts_cap = (size_t)n_cand * 2;
        }
        out = c->d_counts;
    }
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_cand * 2 * sizeof(unsigned long long), c->stream));
    // ---- launch
    ScoreArgs a{};
    a.seq = Planes{c->dH, c->dL, c->dV, c->d_needs_v};
    for (int s = 0; s < NM_MAX_MOD_SLOTS; ++s) {
        const ModSlot &ms = c->slots[s];
        a.st[s] = StatePlanes{ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4], ms.planes[5]};
    }
    a.segments = c->d_segments;
    a.n_segments = c->n_segments;
    a.n_bins = c->n_bins;
    a.programs = c->d_programs;
    a.orig_index = reinterpret_cast<uint32_t *>(ds + off_orig);
    a.cand_range = reinterpret_cast<uint2 *>(ds + off_range);
    a.out = out;
    for (uint32_t i = 0; i < n_active; ++i) a.active_slot[i] = active[i];
    for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) a.slot_is_c[sl] = c->slots[sl].canonical == 'C';
    const uint32_t gx = ((c->n_segments + 7) / 8) * 8;
    // measured crossover (profiles/): up to ~6 candidates per (slot, bin) group the launch is HBM-bound
    const bool fuse = n_active == 2 && (uint64_t)n_prog <= 6ull * n_groups;

    hipEvent_t e0 = c->ev0, e1 = c->ev1;
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
    HIP_TRY(hipEventRecord(e0, c->stream));
    if (n_prog == 0) { /* nothing resident for this batch: the zeroed table is the answer */ }
    else if (!any_wide && all_compact) launch_score<1, 1, true>(a, gx, n_active, fuse, c->stream);
    else if (!any_wide) launch_score<1, 1, false>(a, gx, n_active, fuse, c->stream);
    else if (all_compact) launch_score<2, 2, true>(a, gx, n_active, fuse, c->stream);
    else launch_score<2, 2, false>(a, gx, n_active, fuse, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, c->stream));
    rc = release_stage(c);
    if (rc) return rc;
    c->timed = !c->ev_collect;
    c->launches += 1;
    c->last_wgs = (uint64_t)gx * (fuse ? 1 : std::max(n_active, 1u));
    c->last_compact = all_compact ? n_prog : 0;
    c->last_general = all_compact ? 0 : n_prog;
    if (h_out) {
        HIP_TRY(hipMemcpyAsync(h_out, out, (size_t)n_cand * 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
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

const char *nm_last_error(void) { return g_err.c_str(); }

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
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipMalloc(&c->d_err, sizeof(unsigned int)));
    HIP_TRY(hipMalloc(&c->d_other, sizeof(unsigned long long)));
    *out = c;
    return NM_OK;
}

static void drop_ingest_rows(nm_ctx *c) {
    void *ptrs[] = {c->d_ing_contig, c->d_ing_pos, c->d_ing_strand, c->d_ing_mod};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    c->d_ing_contig = c->d_ing_pos = nullptr;
    c->d_ing_strand = nullptr;
    c->d_ing_mod = nullptr;
    c->ing_nconf = 0;
}

static void free_assembly(nm_ctx *c) {
    drop_ingest_rows(c);
    void *ptrs[] = {c->dH, c->dL, c->dV, c->d_needs_v, c->d_contig_chunk, c->d_contig_len, c->d_segments};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    c->dH = c->dL = c->dV = nullptr;
    c->d_needs_v = nullptr;
    c->d_contig_chunk = nullptr;
    c->d_contig_len = nullptr;
    c->d_segments = nullptr;
    for (int b = 0; b < 4; ++b) {
        if (c->d_rank[b]) (void)hipFree(c->d_rank[b]);
        if (c->d_base_total[b]) (void)hipFree(c->d_base_total[b]);
        c->d_rank[b] = nullptr;
        c->d_base_total[b] = nullptr;
    }
    for (auto &s : c->slots) {
        for (auto &p : s.planes) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        s.present = false;
        s.n_rows = 0;
        drop_slot_ranks(s);
    }
}

int nm_ctx_destroy(nm_ctx *c) {
    if (!c) return NM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_assembly(c);
    for (auto &st : c->stage) {
        if (st.d) (void)hipFree(st.d);
        if (st.h) (void)hipHostFree(st.h);
        if (st.busy) (void)hipEventDestroy(st.busy);
    }
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_programs) (void)hipFree(c->d_programs);
    if (c->d_win_planes) (void)hipFree(c->d_win_planes);
    if (c->d_win_alive) (void)hipFree(c->d_win_alive);
    if (c->d_win_tasks) (void)hipFree(c->d_win_tasks);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_other) (void)hipFree(c->d_other);
    for (auto &pr : c->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return NM_OK;
}

int nm_set_stream(nm_ctx *c, void *hip_stream) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    c->timed = false;
    return NM_OK;
}

static int upload_contigs_impl(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id,
                               uint32_t n_bins, const uint8_t *seq_ascii, bool on_device) {
    if (!c || !offsets || !bin_id || !seq_ascii) return fail(NM_EINVAL, "NULL argument");
    if (n_contigs == 0 || n_bins == 0) return fail(NM_EINVAL, "need at least one contig and one bin");
    if (offsets[0] != 0) return fail(NM_EINVAL, "offsets[0] must be 0");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_assembly(c);
    c->n_contigs = n_contigs;
    c->n_bins = n_bins;
    c->contig_len.assign(n_contigs, 0);
    c->contig_bin.assign(bin_id, bin_id + n_contigs);
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
    uint64_t next = 1;
    std::vector<uint32_t> chunk_contig(1, 0xFFFFFFFFu);
    for (uint32_t oi = 0; oi < n_contigs; ++oi) {
        const uint32_t i = order[oi];
        const uint64_t nch = (c->contig_len[i] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP;
        if (next + nch + 1 >= 0xFFFFFFFFull / CHUNK_WORDS * 8) return fail(NM_ERANGE, "assembly too large for one device context");
        c->contig_chunk[i] = (uint32_t)next;
        c->contig_nchunks[i] = (uint32_t)nch;
        const uint32_t b = bin_id[i];
        if (c->bin_nchunks[b] == 0) c->bin_chunk0[b] = (uint32_t)next;
        c->bin_nchunks[b] += (uint32_t)nch;
        chunk_contig.insert(chunk_contig.end(), nch, i);
        next += nch;
    }
    chunk_contig.push_back(0xFFFFFFFFu);
    c->n_chunks = (uint32_t)(next + 1);
    const size_t words = plane_words(c);
    HIP_TRY(hipMalloc(&c->dH, words * 4));
    HIP_TRY(hipMalloc(&c->dL, words * 4));
    HIP_TRY(hipMalloc(&c->dV, words * 4));
    HIP_TRY(hipMalloc(&c->d_needs_v, c->n_chunks));
    HIP_TRY(hipMemsetAsync(c->dH, 0, words * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->dL, 0, words * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->dV, 0, words * 4, c->stream));
    HIP_TRY(hipMalloc(&c->d_contig_chunk, (size_t)n_contigs * 4));
    HIP_TRY(hipMalloc(&c->d_contig_len, (size_t)n_contigs * 8));
    HIP_TRY(hipMemcpyAsync(c->d_contig_chunk, c->contig_chunk.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_contig_len, c->contig_len.data(), (size_t)n_contigs * 8, hipMemcpyHostToDevice, c->stream));
    // temporaries for the pack pass
    uint8_t *d_ascii = nullptr;
    uint64_t *d_off = nullptr;
    uint32_t *d_chunk_contig = nullptr;
    if (on_device) d_ascii = const_cast<uint8_t *>(seq_ascii);
    else HIP_TRY(hipMalloc(&d_ascii, c->total_bp));
    HIP_TRY(hipMalloc(&d_off, (size_t)(n_contigs + 1) * 8));
    HIP_TRY(hipMalloc(&d_chunk_contig, (size_t)c->n_chunks * 4));
    if (!on_device) HIP_TRY(hipMemcpyAsync(d_ascii, seq_ascii, c->total_bp, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_off, offsets, (size_t)(n_contigs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_chunk_contig, chunk_contig.data(), (size_t)c->n_chunks * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_other, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(pack_kernel, dim3(c->n_chunks), dim3(256), 0, c->stream, d_ascii, d_off, d_chunk_contig,
                       c->d_contig_chunk, c->dH, c->dL, c->dV, c->d_other);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(needs_v_kernel, dim3(c->n_chunks), dim3(CHUNK_WORDS), 0, c->stream, c->dV, c->d_needs_v, c->n_chunks);
    HIP_TRY(hipGetLastError());
    // static segment table: every bin's chunk range cut into pieces of SEG_CHUNKS
    std::vector<uint4> segs;
    for (uint32_t b = 0; b < n_bins; ++b)
        for (uint32_t k = 0; k < c->bin_nchunks[b]; k += SEG_CHUNKS)
            segs.push_back(make_uint4(c->bin_chunk0[b] + k, std::min<uint32_t>(SEG_CHUNKS, c->bin_nchunks[b] - k), b, 0));
    c->n_segments = (uint32_t)segs.size();
    HIP_TRY(hipMalloc(&c->d_segments, segs.size() * sizeof(uint4)));
    HIP_TRY(hipMemcpyAsync(c->d_segments, segs.data(), segs.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
    unsigned long long other = 0;
    HIP_TRY(hipMemcpyAsync(&other, c->d_other, sizeof other, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->other_letters = other;
    if (!on_device) (void)hipFree(d_ascii);
    (void)hipFree(d_off);
    (void)hipFree(d_chunk_contig);
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
    ModSlot &ms = c->slots[mod_slot];
    drop_slot_ranks(ms);
    const size_t words = plane_words(c);
    if (!ms.present || !append) {
        for (auto &p : ms.planes) {
            if (!p) HIP_TRY(hipMalloc(&p, words * 4));
            HIP_TRY(hipMemsetAsync(p, 0, words * 4, c->stream));
        }
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
    ms.n_rows += n_rows;
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

int nm_ingest_pileup(nm_ctx *c, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_code,
                     const uint8_t *strand, const double *fraction_mod, const int32_t *nvalid_cov,
                     const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8], double low, double high,
                     int rows_on_device, uint64_t *n_kept, uint64_t *n_confident) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (!slot_of_mod || !canonical_of_mod || !n_kept || !n_confident) return fail(NM_EINVAL, "NULL argument");
    if (n_rows && (!contig_id || !position || !mod_code || !strand || !fraction_mod || !nvalid_cov)) return fail(NM_EINVAL, "NULL column");
    if (!(high > low)) return fail(NM_EINVAL, "high threshold must exceed low");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t words = plane_words(c);
    IngestSlots sl{};
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m) {
        sl.slot_of_mod[m] = slot_of_mod[m];
        if (slot_of_mod[m] < 0) continue;
        if (slot_of_mod[m] >= NM_MAX_MOD_SLOTS) return fail(NM_EINVAL, "slot %d >= %d", slot_of_mod[m], NM_MAX_MOD_SLOTS);
        if (canonical_of_mod[m] != 'A' && canonical_of_mod[m] != 'C') return fail(NM_EINVAL, "canonical base must be 'A' or 'C'");
        ModSlot &ms = c->slots[slot_of_mod[m]];
        drop_slot_ranks(ms);
        for (auto &p : ms.planes) {
            if (!p) HIP_TRY(hipMalloc(&p, words * 4));
            HIP_TRY(hipMemsetAsync(p, 0, words * 4, c->stream));
        }
        ms.present = true;
        ms.canonical = canonical_of_mod[m];
        ms.low = low;
        ms.high = high;
        ms.n_rows = 0;
        for (int k = 0; k < 6; ++k) sl.planes[slot_of_mod[m]][k] = ms.planes[k];
        sl.can_l[slot_of_mod[m]] = canonical_of_mod[m] == 'C' ? 1u : 0u;
    }
    // device copies of the raw columns
    std::vector<void *> owned;
    auto cleanup = [&]() { for (void *p : owned) (void)hipFree(p); };
    RawRows r{};
    r.n = n_rows;
    if (rows_on_device) {
        r.contig = contig_id; r.position = position; r.mod = mod_code; r.strand = strand; r.frac = fraction_mod; r.nvalid = nvalid_cov;
    } else {
        const void *src[6] = {contig_id, position, mod_code, strand, fraction_mod, nvalid_cov};
        const size_t esz[6] = {4, 4, 1, 1, 8, 4};
        void *dst[6];
        for (int k = 0; k < 6; ++k) {
            dst[k] = nullptr;
            if (hipMalloc(&dst[k], std::max<size_t>(n_rows * esz[k], 16)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory for the raw pileup"); }
            owned.push_back(dst[k]);
            if (n_rows && hipMemcpyAsync(dst[k], src[k], n_rows * esz[k], hipMemcpyHostToDevice, c->stream) != hipSuccess) { cleanup(); return fail(NM_EHIP, "H2D copy of the raw pileup failed"); }
        }
        r.contig = (const uint32_t *)dst[0]; r.position = (const uint32_t *)dst[1]; r.mod = (const int8_t *)dst[2];
        r.strand = (const uint8_t *)dst[3]; r.frac = (const double *)dst[4]; r.nvalid = (const int32_t *)dst[5];
    }
    const size_t n_groups = (size_t)c->n_contigs * NM_MAX_MOD_CODES;
    const uint64_t npos = (uint64_t)c->n_chunks * CHUNK_BP;
    unsigned int *d_cnt = nullptr, *d_kept = nullptr;
    uint8_t *d_ok = nullptr;
    unsigned long long *d_dense = nullptr, *d_scalars = nullptr;
    uint32_t *d_cc = nullptr, *d_cp = nullptr;
    uint8_t *d_cs = nullptr;
    int8_t *d_cm = nullptr;
    const uint64_t conf_cap = std::max<uint64_t>(n_rows, 1);
#define ING_ALLOC(ptr, bytes) do { void *q_ = nullptr; if (hipMalloc(&q_, (bytes)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory in nm_ingest_pileup (%zu bytes)", (size_t)(bytes)); } owned.push_back(q_); ptr = (decltype(ptr))q_; } while (0)
    ING_ALLOC(d_cnt, n_groups * 2 * 4);
    ING_ALLOC(d_kept, n_groups * 4);
    ING_ALLOC(d_ok, n_groups);
    ING_ALLOC(d_dense, npos * 8 * 2);
    ING_ALLOC(d_scalars, 16);
    ING_ALLOC(d_cc, conf_cap * 4);
    ING_ALLOC(d_cp, conf_cap * 4);
    ING_ALLOC(d_cs, conf_cap);
    ING_ALLOC(d_cm, conf_cap);
#undef ING_ALLOC
    hipError_t e = hipSuccess;
    e = hipMemsetAsync(d_cnt, 0, n_groups * 2 * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_kept, 0, n_groups * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_dense, 0, npos * 8 * 2, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_scalars, 0, 16, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream);
    if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "memset failed: %s", hipGetErrorString(e)); }
    if (n_rows) {
        const dim3 grid((unsigned)((n_rows + 255) / 256)), blk(256);
        hipLaunchKernelGGL(ingest_count_kernel, grid, blk, 0, c->stream, r, c->n_contigs, c->d_contig_len, 5, 0.7, d_cnt, c->d_err);
        hipLaunchKernelGGL(ingest_group_kernel, dim3((unsigned)((n_groups + 255) / 256)), blk, 0, c->stream, (uint32_t)n_groups, d_cnt, 0.0001, 50u, d_ok);
        hipLaunchKernelGGL(ingest_scatter_kernel, grid, blk, 0, c->stream, r, 5, d_ok, c->d_contig_chunk, d_dense, d_dense + npos);
        hipLaunchKernelGGL(ingest_decide_kernel, grid, blk, 0, c->stream, r, 5, d_ok, c->d_contig_chunk, d_dense, d_dense + npos, 8, 0.7,
                           low, high, sl, c->dH, c->dL, c->dV, d_kept, d_scalars, d_scalars + 1, d_cc, d_cp, d_cs, d_cm, conf_cap, c->d_err);
        e = hipGetLastError();
        if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "ingest launch failed: %s", hipGetErrorString(e)); }
    }
    unsigned long long scal[2] = {0, 0};
    unsigned int err = 0;
    e = hipMemcpyAsync(scal, d_scalars, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "ingest failed: %s", hipGetErrorString(e)); }
    const uint64_t nconf = scal[1];
    drop_ingest_rows(c);
    c->ing_kept.resize(n_groups);
    if (nconf) {        // exact-size copies; the n_rows-sized scratch goes away with cleanup()
        e = hipMalloc(&c->d_ing_contig, nconf * 4);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_pos, nconf * 4);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_strand, nconf);
        if (e == hipSuccess) e = hipMalloc(&c->d_ing_mod, nconf);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_contig, d_cc, nconf * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_pos, d_cp, nconf * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_strand, d_cs, nconf, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_ing_mod, d_cm, nconf, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { cleanup(); drop_ingest_rows(c); return fail(NM_EHIP, "keeping the confident rows failed: %s", hipGetErrorString(e)); }
        c->ing_nconf = nconf;
    }
    (void)hipMemcpy(c->ing_kept.data(), d_kept, n_groups * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (err & 1u) return fail(NM_EINVAL, "pileup row with contig_id / position / mod code outside the uploaded assembly");
    if (err & 4u) return fail(NM_EINVAL, "duplicate (contig, position, strand) rows within one modification type: the reference's np.isin(assume_unique=True) requires unique positions (find_motifs_bin.py:1258)");
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m)
        if (slot_of_mod[m] >= 0) c->slots[slot_of_mod[m]].n_rows = scal[0];
    *n_kept = scal[0];
    *n_confident = nconf;
    return NM_OK;
}

int nm_ingest_results(nm_ctx *c, uint32_t *conf_contig, uint32_t *conf_position, uint8_t *conf_strand, int8_t *conf_mod,
                      uint64_t capacity, uint32_t *kept_per_contig_mod /*[n_contigs][8]*/) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    const size_t n = c->ing_nconf;
    if (conf_contig || conf_position || conf_strand || conf_mod) {          // all NULL: only the kept table is wanted
        if (capacity < n) return fail(NM_ERANGE, "capacity %llu < %zu confident rows", (unsigned long long)capacity, n);
        if (n) {
            if (!conf_contig || !conf_position || !conf_strand || !conf_mod) return fail(NM_EINVAL, "NULL argument");
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipMemcpy(conf_contig, c->d_ing_contig, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_position, c->d_ing_pos, n * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_strand, c->d_ing_strand, n, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(conf_mod, c->d_ing_mod, n, hipMemcpyDeviceToHost));
        }
    }
    if (kept_per_contig_mod) memcpy(kept_per_contig_mod, c->ing_kept.data(), c->ing_kept.size() * 4);
    return NM_OK;
}

int nm_score_batch(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                   const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                   const uint8_t *cand_masks, int64_t *out_counts) {
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !out_counts))
        return fail(NM_EINVAL, "NULL argument");
    return score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks, nullptr, out_counts);
}

int nm_score_batch_device(nm_ctx *c, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                          const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                          const uint8_t *cand_masks, int64_t *d_out_counts) {
    if (n_cand && (!cand_bin || !cand_mod_slot || !cand_len || !cand_modpos || !cand_mask_offset || !cand_masks || !d_out_counts))
        return fail(NM_EINVAL, "NULL argument");
    return score_impl(c, n_cand, cand_bin, cand_mod_slot, cand_len, cand_modpos, cand_mask_offset, cand_masks,
                      reinterpret_cast<unsigned long long *>(d_out_counts), nullptr);
}

int nm_hit_positions(nm_ctx *c, uint32_t contig_id, uint32_t mod_slot, uint8_t len, uint8_t modpos, const uint8_t *masks,
                     int which, int64_t *out, uint64_t capacity, uint64_t *n_out) {
    if (!c || !masks || !n_out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs has not been called");
    if (contig_id >= c->n_contigs) return fail(NM_EINVAL, "contig_id %u >= %u", contig_id, c->n_contigs);
    if (mod_slot >= NM_MAX_MOD_SLOTS || !c->slots[mod_slot].present) return fail(NM_ESTATE, "mod slot %u has no pileup", mod_slot);
    if (which < 0 || which > 3) return fail(NM_EINVAL, "which must be 0..3");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t full[PROG_DW], prog[PROG_DW];
    bool wide = false;
    uint32_t mpm = 0;
    int rc = compile_program(masks, len, modpos, full, &wide, &mpm);
    if (rc) return rc;
    add_modpos_constraint(full, mpm);
    if (wide) {
        memcpy(prog, full, sizeof full);
    } else {
        memset(prog, 0, sizeof prog);
        memcpy(prog, full + 8, 16 * 4);
        memcpy(prog + 16, full + 32 + 8, 16 * 4);
    }
    const uint32_t nch = c->contig_nchunks[contig_id];
    const size_t out_words = (size_t)nch * CHUNK_WORDS;
    rc = ensure_stage(c, sizeof prog + out_words * 4);
    if (rc) return rc;
    memcpy(c->h_stage, prog, sizeof prog);
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, sizeof prog, hipMemcpyHostToDevice, c->stream));
    uint32_t *d_prog = static_cast<uint32_t *>(c->d_stage);
    uint32_t *d_out = d_prog + PROG_DW;
    const ModSlot &ms = c->slots[mod_slot];
    Planes seq{c->dH, c->dL, c->dV, c->d_needs_v};
    StatePlanes st{ms.planes[0], ms.planes[1], ms.planes[2], ms.planes[3], ms.planes[4], ms.planes[5]};
    dim3 grid((nch + 3) / 4);
    if (wide) hipLaunchKernelGGL((hits_kernel<2, 2>), grid, dim3(256), 0, c->stream, seq, st, c->contig_chunk[contig_id], nch, d_prog, which, d_out);
    else hipLaunchKernelGGL((hits_kernel<1, 1>), grid, dim3(256), 0, c->stream, seq, st, c->contig_chunk[contig_id], nch, d_prog, which, d_out);
    HIP_TRY(hipGetLastError());
    uint32_t *h_out = static_cast<uint32_t *>(c->h_stage) + PROG_DW;
    HIP_TRY(hipMemcpyAsync(h_out, d_out, out_words * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t n = 0;
    for (size_t w = 0; w < out_words; ++w) {
        uint32_t m = h_out[w];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (out && n < capacity) out[n] = (int64_t)(w * 32 + b);
            ++n;
        }
    }
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

// ---- window engine host side ------------------------------------------------------------------------
static int win_grow(nm_ctx *c, uint32_t **buf, uint64_t *cap, uint64_t used, uint64_t need_words) {
    if (used + need_words <= *cap) return NM_OK;
    const uint64_t ncap = std::max<uint64_t>((used + need_words) * 3 / 2, 1u << 20);
    uint32_t *nb = nullptr;
    HIP_TRY(hipMalloc(&nb, ncap * 4));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*buf && used) HIP_TRY(hipMemcpy(nb, *buf, used * 4, hipMemcpyDeviceToDevice));
    if (*buf) (void)hipFree(*buf);
    *buf = nb;
    *cap = ncap;
    return NM_OK;
}

int nm_win_clear(nm_ctx *c) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->win_tasks.clear();
    c->win_planes_used = c->win_alive_used = 0;
    c->win_tasks_dirty = true;
    return NM_OK;
}

int nm_win_add_task(nm_ctx *c, uint32_t n_windows, uint32_t width, const uint8_t *sets, uint32_t *task_id) {
    if (!c || !task_id || (n_windows && !sets)) return fail(NM_EINVAL, "NULL argument");
    if (width == 0 || width > WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", width, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    WinTask t{};
    t.n = n_windows;
    t.nw = (n_windows + 31) / 32;
    t.width = width;
    t.plane_off = c->win_planes_used;
    t.alive_off = c->win_alive_used;
    int rc = win_grow(c, &c->d_win_planes, &c->win_planes_cap, c->win_planes_used, (uint64_t)width * 5 * t.nw);
    if (rc) return rc;
    rc = win_grow(c, &c->d_win_alive, &c->win_alive_cap, c->win_alive_used, t.nw);
    if (rc) return rc;
    if (n_windows) {
        uint8_t *d_sets = nullptr;
        HIP_TRY(hipMalloc(&d_sets, (size_t)n_windows * width));
        HIP_TRY(hipMemcpyAsync(d_sets, sets, (size_t)n_windows * width, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(win_pack_kernel, dim3((t.nw + 255) / 256, width), dim3(256), 0, c->stream, t, d_sets,
                           c->d_win_planes, c->d_win_alive);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(d_sets);
    }
    c->win_planes_used += (uint64_t)width * 5 * t.nw;
    c->win_alive_used += t.nw;
    *task_id = (uint32_t)c->win_tasks.size();
    c->win_tasks.push_back(t);
    c->win_tasks_dirty = true;
    return NM_OK;
}

int nm_win_batch(nm_ctx *c, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                 int32_t *out) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (n_req == 0) return NM_OK;
    if (!req_task || !req_kind || !req_sets || !out) return fail(NM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t max_nw = 1;
    for (uint32_t r = 0; r < n_req; ++r) {
        if (req_task[r] >= c->win_tasks.size()) return fail(NM_EINVAL, "request %u: window task %u does not exist", r, req_task[r]);
        if (req_kind[r] > 1) return fail(NM_EINVAL, "request %u: kind must be 0 (pssm) or 1 (remove)", r);
        max_nw = std::max(max_nw, c->win_tasks[req_task[r]].nw);
    }
    if (c->win_tasks_dirty) {
        if (c->d_win_tasks_cap < c->win_tasks.size()) {
            if (c->d_win_tasks) (void)hipFree(c->d_win_tasks);
            c->d_win_tasks = nullptr;
            c->d_win_tasks_cap = 0;
            HIP_TRY(hipMalloc(&c->d_win_tasks, c->win_tasks.size() * 2 * sizeof(WinTask)));
            c->d_win_tasks_cap = c->win_tasks.size() * 2;
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(c->d_win_tasks, c->win_tasks.data(), c->win_tasks.size() * sizeof(WinTask), hipMemcpyHostToDevice));
        c->win_tasks_dirty = false;
    }
    const uint32_t stride = 2 + 4 * WIN_MAX_W;
    const size_t o_kind = (size_t)n_req * 4, o_sets = (o_kind + n_req + 15) & ~(size_t)15;
    const size_t o_out = (o_sets + (size_t)n_req * WIN_MAX_W + 15) & ~(size_t)15;
    const size_t total = o_out + (size_t)n_req * stride * 4;
    int rc = ensure_stage(c, total);
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
    memcpy(hs, req_task, (size_t)n_req * 4);
    memcpy(hs + o_kind, req_kind, n_req);
    memcpy(hs + o_sets, req_sets, (size_t)n_req * WIN_MAX_W);
    HIP_TRY(hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(ds + o_out, 0, (size_t)n_req * stride * 4, c->stream));
    const uint32_t gy = std::min<uint32_t>(64, (max_nw + 255) / 256);
    hipLaunchKernelGGL(win_request_kernel, dim3(n_req, gy), dim3(256), 0, c->stream, c->d_win_tasks, n_req,
                       reinterpret_cast<const uint32_t *>(ds), ds + o_kind, ds + o_sets, c->d_win_planes, c->d_win_alive,
                       reinterpret_cast<int *>(ds + o_out), stride);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hs + o_out, ds + o_out, (size_t)n_req * stride * 4, hipMemcpyDeviceToHost, c->stream));
    rc = release_stage(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, hs + o_out, (size_t)n_req * stride * 4);
    return NM_OK;
}

// ------------------------------------------------------------------------------------------------------
// window extraction on the device
// ------------------------------------------------------------------------------------------------------
static int base_index(uint8_t base) {
    switch (base) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

static Planes seq_planes(const nm_ctx *c) {
    Planes p;
    p.H = c->dH;
    p.L = c->dL;
    p.V = c->dV;
    p.needs_v = c->d_needs_v;
    return p;
}

static int ensure_rank(nm_ctx *c, int b) {
    if (c->d_rank[b]) return NM_OK;
    HIP_TRY(hipMalloc(&c->d_rank[b], (size_t)c->n_chunks * RANK_PER_CHUNK * 4));
    HIP_TRY(hipMalloc(&c->d_base_total[b], (size_t)c->n_contigs * 8));
    hipLaunchKernelGGL(rank_build_kernel, dim3(c->n_contigs), dim3(64), 0, c->stream, seq_planes(c),
                       static_cast<const uint32_t *>(nullptr), c->d_contig_chunk, c->d_contig_len, b, c->d_rank[b],
                       c->d_base_total[b]);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

int nm_assembly_other_letters(nm_ctx *c, uint64_t *n) {
    if (!c || !n) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    *n = c->other_letters;
    return NM_OK;
}

int nm_contig_base_counts(nm_ctx *c, uint8_t base, uint32_t pad, uint64_t *out) {
    if (!c || !out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    const int b = base_index(base);
    if (b < 0) return fail(NM_EINVAL, "base must be one of A C G T");
    if (2 * pad + 1 > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", 2 * pad + 1, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rank(c, b);
    if (rc) return rc;
    uint64_t *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, (size_t)c->n_contigs * 8));
    hipLaunchKernelGGL(base_count_kernel, dim3((c->n_contigs + 255) / 256), dim3(256), 0, c->stream, seq_planes(c),
                       c->d_contig_chunk, c->d_contig_len, c->n_contigs, b, pad, c->d_base_total[b], d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)c->n_contigs * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(d_out);
    return NM_OK;
}

int nm_bg_counts(nm_ctx *c, uint8_t base, uint32_t pad, uint64_t n_samples, const uint32_t *sample_contig,
                 const uint32_t *sample_rank, uint32_t n_tasks, const uint64_t *task_begin, int64_t *out) {
    if (!c || !task_begin || !out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    const int b = base_index(base);
    if (b < 0) return fail(NM_EINVAL, "base must be one of A C G T");
    const uint32_t W = 2 * pad + 1;
    if (W > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", W, WIN_MAX_W);
    if (n_tasks == 0) return NM_OK;
    if (n_samples && (!sample_contig || !sample_rank)) return fail(NM_EINVAL, "NULL sample column");
    if (task_begin[0] != 0 || task_begin[n_tasks] != n_samples) return fail(NM_EINVAL, "task_begin must run from 0 to n_samples");
    constexpr uint32_t SPB = 2048;                       // samples per workgroup
    std::vector<BgBlock> blocks;
    for (uint32_t t = 0; t < n_tasks; ++t) {
        if (task_begin[t + 1] < task_begin[t]) return fail(NM_EINVAL, "task_begin must be non-decreasing");
        for (uint64_t s0 = task_begin[t]; s0 < task_begin[t + 1]; s0 += SPB)
            blocks.push_back(BgBlock{t, (uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)std::min<uint64_t>(SPB, task_begin[t + 1] - s0)});
    }
    for (uint64_t i = 0; i < n_samples; ++i)
        if (sample_contig[i] >= c->n_contigs) return fail(NM_EINVAL, "sample %llu: contig %u >= %u", (unsigned long long)i, sample_contig[i], c->n_contigs);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rank(c, b);
    if (rc) return rc;
    const size_t o_rank = (size_t)n_samples * 4, o_blk = (o_rank + (size_t)n_samples * 4 + 15) & ~(size_t)15;
    const size_t o_out = (o_blk + blocks.size() * sizeof(BgBlock) + 15) & ~(size_t)15;
    const size_t out_bytes = (size_t)n_tasks * 4 * WIN_MAX_W * 8;
    rc = ensure_stage(c, o_out + out_bytes);
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
    if (n_samples) {
        memcpy(hs, sample_contig, (size_t)n_samples * 4);
        memcpy(hs + o_rank, sample_rank, (size_t)n_samples * 4);
    }
    if (!blocks.empty()) memcpy(hs + o_blk, blocks.data(), blocks.size() * sizeof(BgBlock));
    HIP_TRY(hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(ds + o_out, 0, out_bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
    if (!blocks.empty()) {
        hipLaunchKernelGGL(bg_counts_kernel, dim3((unsigned)blocks.size()), dim3(256), 0, c->stream, seq_planes(c),
                           c->d_rank[b], c->d_contig_chunk, c->d_contig_len, reinterpret_cast<const BgBlock *>(ds + o_blk),
                           reinterpret_cast<const uint32_t *>(ds), reinterpret_cast<const uint32_t *>(ds + o_rank), b, pad,
                           reinterpret_cast<unsigned long long *>(ds + o_out), c->d_err);
        HIP_TRY(hipGetLastError());
    }
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(hs + o_out, ds + o_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    rc = release_stage(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err) return fail(NM_EINVAL, "a sample rank is not below the contig's number of valid starts (nm_contig_base_counts)");
    const uint64_t *ho = reinterpret_cast<const uint64_t *>(hs + o_out);
    for (uint32_t t = 0; t < n_tasks; ++t)
        for (uint32_t r = 0; r < 4; ++r)
            for (uint32_t col = 0; col < W; ++col)
                out[((size_t)t * 4 + r) * W + col] = (int64_t)ho[((size_t)t * 4 + r) * WIN_MAX_W + col];
    return NM_OK;
}

int nm_win_add_task_rows(nm_ctx *c, uint32_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                         const uint8_t *minus, uint32_t pad, uint32_t *task_id) {
    if (!c || !task_id || (n_rows && (!contig_id || !position || !minus))) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    const uint32_t width = 2 * pad + 1;
    if (width > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", width, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    WinTask t{};
    t.n = n_rows;
    t.nw = (n_rows + 31) / 32;
    t.width = width;
    t.plane_off = c->win_planes_used;
    t.alive_off = c->win_alive_used;
    int rc = win_grow(c, &c->d_win_planes, &c->win_planes_cap, c->win_planes_used, (uint64_t)width * 5 * t.nw);
    if (rc) return rc;
    rc = win_grow(c, &c->d_win_alive, &c->win_alive_cap, c->win_alive_used, t.nw);
    if (rc) return rc;
    if (n_rows) {
        const size_t o_minus = (size_t)n_rows * 8;
        rc = ensure_stage(c, o_minus + n_rows);
        if (rc) return rc;
        uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
        uint64_t *centre = reinterpret_cast<uint64_t *>(hs);
        for (uint32_t i = 0; i < n_rows; ++i) {
            const uint32_t ci = contig_id[i];
            if (ci >= c->n_contigs) return fail(NM_EINVAL, "row %u: contig %u >= %u", i, ci, c->n_contigs);
            if (!(position[i] > pad && (uint64_t)position[i] + pad < c->contig_len[ci]))
                return fail(NM_EINVAL, "row %u: position %u is within %u bp of an end of contig %u (seq.py:186)", i, position[i], pad, ci);
            centre[i] = (uint64_t)c->contig_chunk[ci] * CHUNK_BP + position[i];
        }
        memcpy(hs + o_minus, minus, n_rows);
        HIP_TRY(hipMemcpyAsync(ds, hs, o_minus + n_rows, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(win_gather_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, c->stream, t, seq_planes(c),
                           reinterpret_cast<const uint64_t *>(ds), ds + o_minus, pad, c->d_win_planes, c->d_win_alive);
        HIP_TRY(hipGetLastError());
        rc = release_stage(c);
        if (rc) return rc;
    }
    c->win_planes_used += (uint64_t)width * 5 * t.nw;
    c->win_alive_used += t.nw;
    *task_id = (uint32_t)c->win_tasks.size();
    c->win_tasks.push_back(t);
    c->win_tasks_dirty = true;
    return NM_OK;
}

// rank tables over a slot's methylated-row planes + the per-contig row counts inside the edge-filtered range
static int ensure_slot_counts(nm_ctx *c, uint32_t slot, uint32_t pad) {
    ModSlot &ms = c->slots[slot];
    if (!ms.rank[0]) {
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipMalloc(&ms.rank[k], (size_t)c->n_chunks * RANK_PER_CHUNK * 4));
            HIP_TRY(hipMalloc(&ms.rank_total[k], (size_t)c->n_contigs * 8));
            hipLaunchKernelGGL(rank_build_kernel, dim3(c->n_contigs), dim3(64), 0, c->stream, seq_planes(c),
                               static_cast<const uint32_t *>(ms.planes[k == 0 ? 2 : 4]), c->d_contig_chunk, c->d_contig_len, -1,
                               ms.rank[k], ms.rank_total[k]);
            HIP_TRY(hipGetLastError());
        }
        ms.meth_pad = 0xFFFFFFFFu;
    }
    if (ms.meth_pad == pad && ms.meth_counts.size() == (size_t)c->n_contigs * 4) return NM_OK;
    uint64_t *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, (size_t)c->n_contigs * 4 * 8));
    hipLaunchKernelGGL(meth_count_kernel, dim3((c->n_contigs + 255) / 256), dim3(256), 0, c->stream, ms.planes[2], ms.planes[4],
                       c->d_contig_chunk, c->d_contig_len, c->n_contigs, pad, ms.rank_total[0], ms.rank_total[1], d_out);
    HIP_TRY(hipGetLastError());
    ms.meth_counts.assign((size_t)c->n_contigs * 4, 0);
    HIP_TRY(hipMemcpyAsync(ms.meth_counts.data(), d_out, (size_t)c->n_contigs * 4 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(d_out);
    ms.meth_pad = pad;
    return NM_OK;
}

int nm_methylated_row_counts(nm_ctx *c, uint32_t mod_slot, uint32_t pad, uint64_t *out) {
    if (!c || !out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (mod_slot >= NM_MAX_MOD_SLOTS || !c->slots[mod_slot].present) return fail(NM_ESTATE, "mod slot %u holds no pileup", mod_slot);
    if (2 * pad + 1 > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", 2 * pad + 1, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_slot_counts(c, mod_slot, pad);
    if (rc) return rc;
    const std::vector<uint64_t> &mc = c->slots[mod_slot].meth_counts;
    for (uint32_t i = 0; i < c->n_contigs; ++i) {
        out[2 * (size_t)i] = mc[4 * (size_t)i];
        out[2 * (size_t)i + 1] = mc[4 * (size_t)i + 1];
    }
    return NM_OK;
}

int nm_win_add_task_contigs(nm_ctx *c, uint32_t mod_slot, uint32_t n_contigs, const uint32_t *contig_id, uint32_t pad,
                            uint32_t *task_id, uint64_t *n_windows) {
    if (!c || !task_id || !n_windows || (n_contigs && !contig_id)) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (mod_slot >= NM_MAX_MOD_SLOTS || !c->slots[mod_slot].present) return fail(NM_ESTATE, "mod slot %u holds no pileup", mod_slot);
    const uint32_t width = 2 * pad + 1;
    if (width > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", width, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_slot_counts(c, mod_slot, pad);
    if (rc) return rc;
    ModSlot &ms = c->slots[mod_slot];
    std::vector<WinSegment> segs;
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_contigs; ++k) {
        const uint32_t ci = contig_id[k];
        if (ci >= c->n_contigs) return fail(NM_EINVAL, "contig %u >= %u", ci, c->n_contigs);
        for (uint32_t strand = 0; strand < 2; ++strand) {       // plus rows, then minus rows (find_motifs_bin.py:640-659)
            const uint64_t n = ms.meth_counts[4 * (size_t)ci + strand];
            if (!n) continue;
            segs.push_back(WinSegment{(uint32_t)total, ci, strand, (uint32_t)ms.meth_counts[4 * (size_t)ci + 2 + strand]});
            total += n;
        }
    }
    if (total >= 0xFFFFFFFFull) return fail(NM_ERANGE, "more than 4G windows in one task");
    WinTask t{};
    t.n = (uint32_t)total;
    t.nw = (t.n + 31) / 32;
    t.width = width;
    t.plane_off = c->win_planes_used;
    t.alive_off = c->win_alive_used;
    rc = win_grow(c, &c->d_win_planes, &c->win_planes_cap, c->win_planes_used, (uint64_t)width * 5 * t.nw);
    if (rc) return rc;
    rc = win_grow(c, &c->d_win_alive, &c->win_alive_cap, c->win_alive_used, t.nw);
    if (rc) return rc;
    if (total) {
        rc = ensure_stage(c, segs.size() * sizeof(WinSegment));
        if (rc) return rc;
        memcpy(c->h_stage, segs.data(), segs.size() * sizeof(WinSegment));
        HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, segs.size() * sizeof(WinSegment), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(win_gather_contigs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, c->stream, t, seq_planes(c),
                           static_cast<const uint32_t *>(ms.planes[2]), static_cast<const uint32_t *>(ms.planes[4]),
                           static_cast<const uint32_t *>(ms.rank[0]), static_cast<const uint32_t *>(ms.rank[1]), c->d_contig_chunk,
                           c->d_contig_len, reinterpret_cast<const WinSegment *>(c->d_stage), (uint32_t)segs.size(), pad,
                           c->d_win_planes, c->d_win_alive, c->d_err);
        HIP_TRY(hipGetLastError());
        rc = release_stage(c);
        if (rc) return rc;
    }
    c->win_planes_used += (uint64_t)width * 5 * t.nw;
    c->win_alive_used += t.nw;
    *task_id = (uint32_t)c->win_tasks.size();
    *n_windows = total;
    c->win_tasks.push_back(t);
    c->win_tasks_dirty = true;
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
    c->ev_used = 0;
    c->ev_collect = enable != 0;
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
