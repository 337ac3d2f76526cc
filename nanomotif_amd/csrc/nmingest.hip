// libnmscan — raw pileup ingestion: the reference's three pre-filters on the device, classification into the state
// planes, kept-row counts (C ABI: nm_ingest_pileup / nm_ingest_results, include/nmscan.h).
#include <chrono>
#include "nmscan_internal.h"

using namespace nmdetail;

namespace {

// ------------------------------------------------------------------------------------------------------
// Raw pileup ingestion: the reference's three pre-filters evaluated on the device (dataload.py:191-247, in the order
// of find_motifs_bin.py:399-414), then classification into the state planes and a compact list of the surviving
// confidently methylated rows (the input of window extraction, find_motifs_bin.py:625-661).
// ------------------------------------------------------------------------------------------------------

struct RawRows {
    uint64_t n;
    const uint32_t *contig;      // engine-local contig id, 0xFFFFFFFF = contig not resident (row ignored)
    const uint32_t *position;
    const int8_t *mod;           // 0..NM_CODE_STRIDE-1; only codes < NM_MAX_MOD_CODES can have a slot
    const uint8_t *strand;
    const double *frac;
    const int32_t *nvalid;
};

// Per-(contig, mod code) counters.  Pileup rows arrive sorted by contig, so (1) the lanes of a wave almost always share
// their key: the rows of one wave iteration are grouped by key (leader election over the ballot) — a billion same-address
// atomics were 0.75 s of the 1 Gbp ingest; and (2) a wave that walks a CONTIGUOUS range of rows meets the same two or
// three keys for thousands of iterations: the sums stay in a four-entry cache of wave-uniform registers and go to memory
// when a key is evicted or the wave is done — with one atomic per wave iteration every wave of the device was still
// hammering the same few addresses (3e7 serialised atomics = most of the 24 ms the classification pass took).
#ifndef NM_DECIDE_MEMO
#define NM_DECIDE_MEMO 1   /* the contig a wave walks through: verdicts / chunk / part membership in wave-uniform registers: 7.45 -> 6.7 ms at 1e9 rows
                              (tools/gpu_r4m.sh; requesting the next turn's columns ahead as well changed nothing at 1, 2 or 3 pieces: 6.7 / 6.7 / 7.4 ms —
                              the pass is bound by the instructions it issues per row, not by the loads in flight) */
#endif
#ifndef NM_DECIDE_U
#define NM_DECIDE_U 2      /* same-device A/B at 1e9 rows (tools/gpu_r4e.sh): 4 pieces 8.46 ms (101 VGPRs, 4 waves), 2 pieces 7.52 (78, 6 waves), 1 piece 8.22 */
#endif
struct KeyCache {                       // four entries, most recently used first; every member is wave-uniform
    uint32_t k0 = ~0u, k1 = ~0u, k2 = ~0u, k3 = ~0u;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;      // counter [key * stride]
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;      // counter [key * stride + 1]
};

__device__ __forceinline__ void cache_flush_one(uint32_t key, uint32_t a, uint32_t b, unsigned int *counters, uint32_t stride, uint32_t lane) {
    if (lane == 0 && key != ~0u) {
        if (a) atomicAdd(counters + (size_t)key * stride, a);
        if (stride > 1 && b) atomicAdd(counters + (size_t)key * stride + 1, b);
    }
}

// all arguments but `lane` are wave-uniform.  The entries are plain scalars moved to the front on use (no indexed array: a
// `next` index made the compiler keep the round-3 cache in private memory — 56 bytes of scratch per lane and a dependent
// scratch round trip in every call, in the two classification kernels).
__device__ __forceinline__ void cache_add(KeyCache &kc, unsigned int *counters, uint32_t stride, uint32_t key, uint32_t a, uint32_t b, uint32_t lane) {
    if (kc.k0 == key) {
        kc.a0 += a;
        kc.b0 += b;
        return;
    }
    uint32_t k = key, x = a, y = b;               // what goes to the front
    if (kc.k1 == key) {
        x += kc.a1; y += kc.b1;
    } else if (kc.k2 == key) {
        x += kc.a2; y += kc.b2;
        kc.k2 = kc.k1; kc.a2 = kc.a1; kc.b2 = kc.b1;
    } else {
        if (kc.k3 == key) { x += kc.a3; y += kc.b3; }
        else cache_flush_one(kc.k3, kc.a3, kc.b3, counters, stride, lane);      // the least recently used leaves
        kc.k3 = kc.k2; kc.a3 = kc.a2; kc.b3 = kc.b2;
        kc.k2 = kc.k1; kc.a2 = kc.a1; kc.b2 = kc.b1;
    }
    kc.k1 = kc.k0; kc.a1 = kc.a0; kc.b1 = kc.b0;
    kc.k0 = k; kc.a0 = x; kc.b0 = y;
}

__device__ __forceinline__ void cache_flush(const KeyCache &kc, unsigned int *counters, uint32_t stride, uint32_t lane) {
    cache_flush_one(kc.k0, kc.a0, kc.b0, counters, stride, lane);
    cache_flush_one(kc.k1, kc.a1, kc.b1, counters, stride, lane);
    cache_flush_one(kc.k2, kc.a2, kc.b2, counters, stride, lane);
    cache_flush_one(kc.k3, kc.a3, kc.b3, counters, stride, lane);
}

// counters[key * stride] += lanes with pred0, counters[key * stride + 1] += lanes with pred0 && pred1, through the cache.
// All active lanes must call this together.
__device__ __forceinline__ void wave_add_keyed(KeyCache &kc, unsigned int *counters, uint32_t key, bool pred0, bool pred1, uint32_t stride) {
    const uint32_t lane = __lane_id();
    unsigned long long todo = __ballot(pred0);
    while (todo) {                                           // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const bool mine = pred0 && key == k;
        const unsigned long long same = __ballot(mine);
        const unsigned long long same1 = __ballot(mine && pred1);
        cache_add(kc, counters, stride, k, (uint32_t)__popcll(same), (uint32_t)__popcll(same1), lane);
        todo &= ~same;
    }
}

// The rows [row_begin, row_end) one wave walks, 64 at a time: the device's waves cut the pileup into contiguous ranges.
__device__ __forceinline__ void wave_row_range(uint64_t n, uint64_t *row_begin, uint64_t *row_end) {
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6), id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t per = (((n + waves - 1) / waves) + 63) & ~(uint64_t)63;
    *row_begin = std::min<uint64_t>(n, id * per);
    *row_end = std::min<uint64_t>(n, (id + 1) * per);
}

// (1) coverage filter + per (contig, mod code) counts for the frequency filter
// Also establishes, for free, whether the rows come the way modkit writes them — every contig's rows in ONE run, positions
// non-decreasing inside it (order bit 1 = a position decreases, runs[contig] = number of runs): the classification pass
// then assembles complete plane words and writes them with plain stores instead of read-modify-write atomics.
__global__ __launch_bounds__(256) void ingest_count_kernel(RawRows r, uint32_t n_contigs, const uint64_t *__restrict__ contig_len,
                                    int min_cov, double meth_thr, unsigned int *cnt /*[contig][mod][2]*/, unsigned int *err,
                                    unsigned int *order /*[0]: flags*/, unsigned int *runs /*[contig]*/,
                                    unsigned int *wave_cand /*[waves]: rows that can enter the adjacency test (upper bound)*/) {
    uint64_t row_begin, row_end;
    wave_row_range(r.n, &row_begin, &row_end);
    const uint32_t lane = threadIdx.x & 63;
    KeyCache kc;
    uint32_t n_cand = 0;                                             // wave-uniform
    // 256 rows per turn: the loads of four consecutive 64-row pieces are all issued before the first value is looked at —
    // no test sits between two loads (the short-circuit tests used to make three dependent round trips of them), and a
    // wave asks memory for 1 KB of consecutive addresses per column at a time instead of 256 B
    constexpr int U = 4;
    uint32_t prev_c = 0xFFFFFFFEu, prev_p = 0;                       // the row before this wave's first (0xFFFFFFFE: none)
    if (row_begin > 0 && row_begin < row_end) {
        prev_c = r.contig[row_begin - 1];
        prev_p = r.position[row_begin - 1];
    }
    for (uint64_t i0 = row_begin; i0 < row_end; i0 += 64 * U) {
        uint32_t c[U], p[U];
        int m[U], nv[U];
        double f[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * 64 + lane;
            in[u] = i < row_end;
            const uint64_t ii = in[u] ? i : row_begin;
            c[u] = r.contig[ii];
            p[u] = r.position[ii];
            m[u] = r.mod[ii];
            nv[u] = r.nvalid[ii];
            f[u] = r.frac[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + (uint64_t)u * 64 >= row_end) break;             // wave-uniform
            const uint32_t cc = in[u] ? c[u] : 0xFFFFFFFFu;
            uint32_t pc = __shfl_up(cc, 1), pp = __shfl_up(p[u], 1);
            if (lane == 0) { pc = prev_c; pp = prev_p; }
            if (in[u] && cc != 0xFFFFFFFFu && cc < n_contigs) {
                if (pc == cc) { if (p[u] < pp) atomicOr(order, 1u); }
                else atomicAdd(runs + cc, 1u);
            }
            prev_c = __shfl(cc, 63);                                 // (beyond the wave's last row: never used again)
            prev_p = __shfl(p[u], 63);
            bool counted = cc != 0xFFFFFFFFu;
            if (counted && (cc >= n_contigs || p[u] >= contig_len[cc] || m[u] < 0)) {
                atomicOr(err, 1u);
                counted = false;
            }
            counted = counted && nv[u] > min_cov;                    // dataload.py:199: Nvalid_cov > 5
            const bool is_mod = counted && f[u] > meth_thr;          // dataload.py:215: fraction_mod > 0.7
            // (a row whose percentage is null — fraction < 0 — counts as a position, never as a modified one: pl.count() vs
            //  (fraction_mod > thr).sum(), dataload.py:216-217)
            wave_add_keyed(kc, cnt, counted ? cc * NM_CODE_STRIDE + (uint32_t)m[u] : 0u, counted, is_mod, 2);
            n_cand += (uint32_t)__popcll(__ballot(counted && f[u] >= meth_thr));
        }
    }
    cache_flush(kc, cnt, 2, lane);
    if (lane == 0) wave_cand[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = n_cand;
}

// exclusive prefix of n <= 2^20 counters by ONE workgroup: out[i] = sum of in[0 .. i), out[n] = the total
__global__ __launch_bounds__(1024) void ingest_scan_kernel(const unsigned int *__restrict__ in, uint32_t n, unsigned long long *out) {
    __shared__ unsigned long long part[1024];
    const uint32_t per = (n + 1023) / 1024, a = threadIdx.x * per, b = min(n, a + per);
    unsigned long long sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += in[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
        out[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint32_t i = a; i < b; ++i) { out[i] = run; run += in[i]; }
}

// (3a') adjacency filter on rows in modkit's order: the rows that can decide a verdict (the ones ingest_scatter_kernel
// would scatter) are COMPACTED IN ROW ORDER instead — contig, position, strand, fraction of ~1.5 % of a pileup — and a
// tested row looks its window up in that list (binary search + the few entries within 8 positions).  No table of one
// maximum per position and strand: at 1 Gbp that table is 16 GB, 98.5 % of it the zeros it was cleared to.
// The classification pass (ingest_decide_kernel<true>) compacts the candidates of every wave's row range into the block the
// counting pass sized for it (an upper bound: the frequency verdicts were not known then); ingest_pack_kernel closes the gaps.
struct CandList {
    unsigned long long *key;       // contig << 32 | position: ascending inside a contig's run (the runs come in file order)
    unsigned long long *fbits;     // IEEE bits of the fraction (>= 0: they order like the values)
    uint8_t *strand;
    int8_t *mod;
    unsigned long long *range;     // [contig][2]: the contig's entries are [first, end)
};

// where every contig's entries begin and end in the packed list (a contig without candidates keeps the cleared 0, 0)
__global__ __launch_bounds__(256) void ingest_ranges_kernel(CandList list, const unsigned long long *__restrict__ list_n) {
    const unsigned long long n = list_n[0];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(list.key[i] >> 32);
        if (i == 0 || (uint32_t)(list.key[i - 1] >> 32) != c) list.range[2 * (size_t)c] = i;
        if (i + 1 == n || (uint32_t)(list.key[i + 1] >> 32) != c) list.range[2 * (size_t)c + 1] = i + 1;
    }
}

__global__ __launch_bounds__(256) void ingest_pack_kernel(uint32_t n_waves, const unsigned long long *__restrict__ wave_first,
                                   const unsigned int *__restrict__ wave_n, const unsigned long long *__restrict__ packed_first,
                                   CandList src, CandList dst) {
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const unsigned long long a = wave_first[w], b = packed_first[w];
    for (uint32_t i = lane; i < wave_n[w]; i += 64) {
        dst.key[b + i] = src.key[a + i];
        dst.fbits[b + i] = src.fbits[a + i];
        dst.strand[b + i] = src.strand[a + i];
        dst.mod[b + i] = src.mod[a + i];
    }
}

// (2) frequency filter verdict per (contig, mod code): n_mod / n > 1e-4 and n_mod > 50 (dataload.py:218-219)
__global__ void ingest_group_kernel(uint32_t n_groups, const unsigned int *__restrict__ cnt, double min_freq,
                                    unsigned int min_mods, uint8_t *ok, const unsigned int *__restrict__ runs, unsigned int *order) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    if (g % NM_CODE_STRIDE == 0 && runs[g / NM_CODE_STRIDE] > 1) atomicOr(order, 2u);       // a contig's rows in several places
    const unsigned int n = cnt[g * 2], nm = cnt[g * 2 + 1];
    ok[g] = n > 0 && ((double)nm / (double)n) > min_freq && nm > min_mods;
}

// (3a) adjacency filter, scatter: per strand the maximal fraction at every position (mod codes mixed, dataload.py:237).
// Only rows the verdict below tests (fraction >= meth_thr) can decide it — a tested row fails iff a row in its window has
// a LARGER fraction, and that row is then above the threshold itself — so only those (~1 % of a pileup) are scattered;
// every other position keeps the 0 the arrays were cleared to.
__global__ __launch_bounds__(256) void ingest_scatter_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                      const uint64_t *__restrict__ dense_off, unsigned long long *dense_plus,
                                      unsigned long long *dense_minus, double meth_thr, unsigned int *err) {
    // waves walk contiguous row ranges, 256 rows per turn with every load issued before the first value is looked at
    // (with early returns between the loads the kernel ran at a third of the rate the 18 bytes per row allow)
    uint64_t row_begin, row_end;
    wave_row_range(r.n, &row_begin, &row_end);
    const uint32_t lane = threadIdx.x & 63;
    constexpr int U = 4;
    for (uint64_t i0 = row_begin; i0 < row_end; i0 += 64 * U) {
        double f[U];
        uint32_t c[U];
        int m[U], nv[U];
        uint8_t st[U];
        uint64_t idx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * 64 + lane;
            idx[u] = i < row_end ? i : ~0ull;
            const uint64_t ii = i < row_end ? i : row_begin;
            f[u] = r.frac[ii];
            c[u] = r.contig[ii];
            m[u] = r.mod[ii];
            nv[u] = r.nvalid[ii];
            st[u] = r.strand[ii];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx[u] == ~0ull || !(f[u] >= meth_thr) || c[u] == 0xFFFFFFFFu || m[u] < 0 || nv[u] <= min_cov) continue;   // (~1 % of the rows go on)
            if (!ok[(size_t)c[u] * NM_CODE_STRIDE + m[u]]) continue;
            if (st[u] != '+' && st[u] != '-') continue;
            const uint64_t off = dense_off[c[u]];
            if (off == ~0ull) { atomicOr(err, 8u); continue; }        // contig not listed for this part
            // fractions are >= 0, so their IEEE bit patterns order like the values
            atomicMax((st[u] == '+' ? dense_plus : dense_minus) + off + r.position[idx[u]], (unsigned long long)__double_as_longlong(f[u]));
        }
    }
}

// (3b) adjacency verdict + classification + confident-row list.  A row survives iff its fraction equals the maximum
// over positions p-d .. p+d of its contig and strand, or is below the threshold (dataload.py:244).
struct IngestSlots {
    int slot_of_mod[NM_MAX_MOD_CODES];        // -1: mod code not scored
    uint32_t *planes[NM_MAX_MOD_SLOTS][6];    // M U MP UP MM UM per slot
    uint32_t can_l[NM_MAX_MOD_SLOTS];         // canonical base C (1) or A (0)
};

// One row's columns, as loaded (nothing tested yet).
struct RowCols {
    uint32_t c, pos;
    int m, nv;
    uint8_t st;
    double f;
};

__device__ __forceinline__ RowCols load_row(const RawRows &r, uint64_t i) {
    RowCols x;
    x.c = r.contig[i];
    x.pos = r.position[i];
    x.m = r.mod[i];
    x.nv = r.nvalid[i];
    x.st = r.strand[i];
    x.f = r.frac[i];
    return x;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}

// Every wave walks a contiguous range of rows, 64 per iteration.
//  * No test sits between two loads of a row, and four iterations' columns are requested at once (the kernel was bound by
//    the round trip of its own loads: 8 waves per SIMD x one dependent chain each, 256 B per request).
//  * Rows at or above the adjacency threshold (~1.5 % of a pileup) need 17 scattered loads of the dense maxima: they are
//    QUEUED (row index, LDS) and judged 64 at a time — one dependent round trip per 64 candidates instead of one in two
//    of three iterations.  Their classification bits (the methylated planes: sparse) go to memory as atomicOr.
//  * The bits of all other classified rows (the unmethylated planes: nearly every row) are collected per wave in a circular
//    LDS window of 32 plane words per (slot, strand).  When the input is in modkit's order (order[0] == 0: verified by the
//    counting pass) a word is COMPLETE once the wave has moved past it, and leaves as a plain store, whole runs of
//    consecutive words per plane at a time; only the first word a wave touches and what it holds at its end can be shared
//    with a neighbouring wave and go out as atomicOr.  Unordered input: the window is flushed with atomicOr every iteration.
template <bool LIST, int U>      // LIST: the candidate rows are ingest_judge_kernel's — skipped here, no queue; U: 64-row pieces requested together
__global__ __launch_bounds__(256) void ingest_decide_kernel(RawRows r, int min_cov, const uint8_t *__restrict__ ok,
                                     const uint32_t *__restrict__ contig_chunk, const uint64_t *__restrict__ dense_off,
                                     const unsigned long long *__restrict__ dense_plus,
                                     const unsigned long long *__restrict__ dense_minus, int adjacency, double meth_thr,
                                     double low, double high, IngestSlots sl, const unsigned int *__restrict__ order,
                                     unsigned int *kept /*[contig][mod]*/, unsigned long long *n_kept,
                                     unsigned long long *n_classified, const unsigned long long *__restrict__ wave_first, CandList out,
                                     unsigned int *wave_n, unsigned int *err) {
    // LIST: this pass also COMPACTS the candidate rows, in row order, into the block the counting pass sized for the wave
    // (an upper bound: the frequency verdicts were not known then; ingest_pack_kernel closes the gaps)
    constexpr uint32_t WIN = 32, UP_PLANES = NM_MAX_MOD_SLOTS * 2, QCAP = LIST ? 1 : 64 + 4 * 64;
    __shared__ uint32_t tab[4][UP_PLANES * WIN];          // [slot * 2 + minus][word & 31]
    __shared__ uint32_t *u_plane[UP_PLANES];
    __shared__ unsigned long long queue[4][QCAP];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t e = threadIdx.x; e < 4 * UP_PLANES * WIN; e += blockDim.x) (&tab[0][0])[e] = 0;
    if (threadIdx.x < UP_PLANES) u_plane[threadIdx.x] = sl.planes[threadIdx.x >> 1][(threadIdx.x & 1) ? 5 : 3];   // UP / UM
    __syncthreads();
    // plain stores need ordered input AND candidates that can never set an unmethylated bit later (low < the adjacency threshold)
    const bool sorted = __builtin_amdgcn_readfirstlane(order[0]) == 0 && low < meth_thr;
    uint64_t row_begin, row_end;
    wave_row_range(r.n, &row_begin, &row_end);
    unsigned long long my_kept = 0, my_cls = 0;
    KeyCache kc;
    uint32_t *const mytab = tab[wave];
    unsigned long long *const myq = queue[wave];
    const size_t wid = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const unsigned long long list_base = LIST ? wave_first[wid] : 0ull;
    uint32_t list_n = 0;                                  // wave-uniform
    uint32_t qn = 0;                                      // wave-uniform
    uint64_t w_base = 0, first_word = ~0ull;              // wave-uniform
    uint32_t cur_contig = 0xFFFFFFFFu;
    bool have = false;
    uint32_t memo_c = 0xFFFFFFFFu, memo_chunk = 0;        // wave-uniform: what hangs on the contig the wave is walking through
    unsigned long long memo_ok = 0;
    bool memo_part = false;

    // words [lo, hi) of the window -> memory.  Lanes run along the words of one plane: consecutive stores.
    auto flush_range = [&](uint64_t lo, uint64_t hi, bool all_atomic) {
        if (!have || hi <= lo) return;
        const uint32_t nw = (uint32_t)(hi - lo);
        __asm__ volatile("" ::: "memory");
        for (uint32_t e = lane; e < UP_PLANES * nw; e += 64) {
            const uint32_t pid = e / nw;
            const uint64_t w = lo + e % nw;
            const uint32_t bits = mytab[pid * WIN + (uint32_t)(w & (WIN - 1))];
            if (bits) {
                if (all_atomic || w == first_word) atomicOr(u_plane[pid] + w, bits);
                else u_plane[pid][w] = bits;
                mytab[pid * WIN + (uint32_t)(w & (WIN - 1))] = 0;
            }
        }
        __asm__ volatile("" ::: "memory");
    };

    // the adjacency verdict of up to 64 queued rows, one per lane (n <= 64 of them)
    auto judge = [&](uint32_t n) {
        const bool on = lane < n;
        const uint64_t i = on ? myq[lane] : row_begin;
        const RowCols x = load_row(r, i);
        const bool plus = x.st == '+';
        unsigned long long mx = 0;                        // >= 64 zero positions around every contig
        if (on) {
            const uint64_t doff = dense_off[x.c];
            const unsigned long long *d = (plus ? dense_plus : dense_minus) + doff + x.pos;
            if (adjacency == 8) {
                unsigned long long v[17];                 // all 17 loads are issued before the first is waited for
#pragma unroll
                for (int k = 0; k < 17; ++k) v[k] = d[k - 8];
#pragma unroll
                for (int k = 0; k < 17; ++k) mx = max(mx, v[k]);
            } else {
                for (int k = -adjacency; k <= adjacency; ++k) mx = max(mx, d[k]);
            }
        }
        const bool alive = on && mx == (unsigned long long)__double_as_longlong(x.f);
        const bool listed = alive && x.m < NM_MAX_MOD_CODES;
        wave_add_keyed(kc, kept, listed ? x.c * NM_MAX_MOD_CODES + (uint32_t)x.m : 0u, listed, false, 1);
        my_kept += alive;
        const int slot = listed ? sl.slot_of_mod[x.m] : -1;
        const bool meth = slot >= 0 && x.f >= high, non = slot >= 0 && x.f <= low;
        my_cls += meth || non;
        if (meth || non) {
            const uint64_t g = (uint64_t)contig_chunk[x.c] * CHUNK_BP + x.pos;
            // M U MP UP MM UM: methylated plus / minus = 2 / 4, unmethylated = 3 / 5 (non only with low >= the threshold:
            // then the kernel runs unordered, every unmethylated bit is an atomicOr)
            atomicOr(sl.planes[slot][(plus ? 2 : 4) + (meth ? 0 : 1)] + (g >> 5), 1u << (g & 31));
        }
    };

    // 256 rows per turn: the columns of four consecutive 64-row pieces are requested together (1 KB of consecutive addresses
    // per column and wave — the counting pass reaches the streaming rate of the device this way), then the pieces are
    // worked on one after the other
    for (uint64_t i00 = row_begin; i00 < row_end; i00 += 64 * U) {
      RowCols cols[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cols[u] = load_row(r, min(i00 + (uint64_t)u * 64 + lane, row_end - 1));
      // what hangs on the contig / mod code of a row (frequency-filter verdict, dense offset, first chunk) is fetched for
      // all four pieces together: one dependent round trip per 256 rows, not one per 64
      bool pre[U], in_part[U];
      uint8_t okv[U];
      uint32_t chunkv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const RowCols &x = cols[u];
        pre[u] = i00 + (uint64_t)u * 64 + lane < row_end && x.c != 0xFFFFFFFFu && x.m >= 0 && x.nv > min_cov && (x.st == '+' || x.st == '-') && !(x.f < 0);
        // a piece's rows nearly always share ONE contig (a pileup is grouped by contig): its verdicts, chunk and part
        // membership stay in wave-uniform registers from piece to piece — no second, dependent round trip for the piece
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.c);
        if (NM_DECIDE_MEMO && c0 != 0xFFFFFFFFu && __ballot(pre[u] && (x.c != c0 || x.m >= 8)) == 0) {
            if (c0 != memo_c) {
                memo_c = c0;
                memo_ok = *reinterpret_cast<const unsigned long long *>(ok + (size_t)c0 * NM_CODE_STRIDE);      // codes 0..7
                memo_chunk = contig_chunk[c0];
                memo_part = dense_off[c0] != ~0ull;
            }
            okv[u] = (uint8_t)(memo_ok >> (8 * (x.m & 7)));
            in_part[u] = memo_part;
            chunkv[u] = memo_chunk;
        } else {
            const uint32_t c = pre[u] ? x.c : 0u;
            okv[u] = ok[(size_t)c * NM_CODE_STRIDE + (pre[u] ? x.m : 0)];
            in_part[u] = dense_off[c] != ~0ull;                     // (~0: the contig is not listed for this part; flagged by the scatter pass)
            chunkv[u] = contig_chunk[c];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t i0 = i00 + (uint64_t)u * 64;
        if (i0 >= row_end) break;                                // wave-uniform
        const uint64_t i = i0 + lane;
        const RowCols x = cols[u];
        const bool alive = pre[u] && okv[u] && in_part[u];
        const uint32_t c = alive ? x.c : 0u;
        const bool plus = x.st == '+';
        // candidates for the adjacency test are queued; everybody else survives the filter
        const bool cand = alive && !(x.f < meth_thr);
        const unsigned long long cmask = __ballot(cand);
        if (!LIST && cmask) {                                    // (judged after the four pieces: at most 63 + 256 are waiting)
            if (cand) myq[qn + (uint32_t)__popcll(cmask & ((1ull << lane) - 1ull))] = i;
            qn += (uint32_t)__popcll(cmask);
        }
        if (LIST) {
            if (pre[u] && okv[u] && !in_part[u] && x.f >= meth_thr) atomicOr(err, 8u);     // contig not listed for this part
            const bool take = cand && x.f >= meth_thr;           // (a NaN fraction is a candidate without an entry: it fails like against the table)
            const unsigned long long tmask = __ballot(take);
            if (take) {
                const unsigned long long at = list_base + list_n + (uint32_t)__popcll(tmask & ((1ull << lane) - 1ull));
                out.key[at] = ((unsigned long long)x.c << 32) | x.pos;
                out.fbits[at] = (unsigned long long)__double_as_longlong(x.f);
                out.strand[at] = x.st;
                out.mod[at] = (int8_t)x.m;
            }
            list_n += (uint32_t)__popcll(tmask);
        }
        const bool pass = alive && !cand;
        const int m = pass ? x.m : 0;
        const bool listed = pass && m < NM_MAX_MOD_CODES;        // codes beyond the ABI's eight are filtered with the rest, never reported
        wave_add_keyed(kc, kept, listed ? c * NM_MAX_MOD_CODES + (uint32_t)m : 0u, listed, false, 1);     // all lanes take part
        my_kept += pass;
        const int slot = listed ? sl.slot_of_mod[m] : -1;
        const bool non = slot >= 0 && x.f <= low;                // (below the adjacency threshold: never methylated unless high < threshold)
        const bool meth = slot >= 0 && x.f >= high;
        // classified rows are counted; the host compares the total with the population count of the general planes
        // afterwards — a duplicate (contig, position, strand) row sets a bit twice and shows up there
        my_cls += meth || non;
        const uint64_t g = (meth || non) ? (uint64_t)chunkv[u] * CHUNK_BP + x.pos : 0;
        if (meth && !non) atomicOr(sl.planes[slot][plus ? 2 : 4] + (g >> 5), 1u << (g & 31));      // only with high < threshold
        unsigned long long todo = __ballot(non);
        const uint64_t w = g >> 5;
        const uint32_t pid = non ? (uint32_t)slot * 2u + (plus ? 0u : 1u) : 0u;
        const uint32_t bit = 1u << (g & 31);
        while (todo) {                                           // wave-uniform; one turn unless a contig ends in this iteration
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)c, leader);
            const uint64_t lw = readlane64(w, leader);
            if (!have || lc != cur_contig || lw < w_base) {      // (lw < w_base: unordered input only)
                flush_range(w_base, w_base + WIN, !sorted);
                cur_contig = lc;
                w_base = lw;
                have = true;
                if (first_word == ~0ull) first_word = lw;
            } else if (lw >= w_base + WIN / 2) {                 // the words before the leader's are complete
                flush_range(w_base, min(lw, w_base + WIN), !sorted);
                w_base = lw;
            }
            const bool mine = non && c == lc && w >= w_base && w - w_base < WIN;
            if (mine) atomicOr(&mytab[pid * WIN + (uint32_t)(w & (WIN - 1))], bit);
            const unsigned long long done = __ballot(mine);
            todo &= ~done;
            if (!sorted) {                                       // unordered: whatever does not fit this window goes straight to memory
                if (non && !mine && ((todo >> lane) & 1)) atomicOr(u_plane[pid] + w, bit);
                todo = 0;
            }
        }
        if (!sorted) flush_range(w_base, w_base + WIN, true);
      }
      while (qn >= 64) {                                         // wave-uniform
        __asm__ volatile("" ::: "memory");
        judge(64);
        for (uint32_t k = lane; k + 64 < qn; k += 64) {          // the rest moves down (reads of a turn before its writes)
            const unsigned long long moved = myq[k + 64];
            __asm__ volatile("" ::: "memory");
            myq[k] = moved;
        }
        qn -= 64;
      }
    }
    while (qn) {                                                 // the rest of the queue
        const uint32_t n = min(qn, 64u);
        __asm__ volatile("" ::: "memory");
        judge(n);
        const unsigned long long moved = lane + 64 < qn ? myq[lane + 64] : 0ull;
        __asm__ volatile("" ::: "memory");
        if (lane + 64 < qn) myq[lane] = moved;
        qn -= n;
    }
    flush_range(w_base, w_base + WIN, true);                     // what is left may share its words with the next wave
    if (LIST && lane == 0) wave_n[wid] = list_n;
    cache_flush(kc, kept, 1, lane);
    for (int d = 32; d; d >>= 1) {
        my_kept += __shfl_xor(my_kept, d);
        my_cls += __shfl_xor(my_cls, d);
    }
    // (per wave, not per workgroup: a barrier here keeps the finished waves of a workgroup resident until its slowest one is
    // done — measured 6.7 -> 7.45 ms)
    if ((threadIdx.x & 63) == 0) {
        if (my_kept) atomicAdd(n_kept, my_kept);
        if (my_cls) atomicAdd(n_classified, my_cls);
    }
}

// (3b') the candidate rows of an ordered pileup, judged where they sit in the list: an entry survives iff no entry of its
// strand within `adjacency` positions has a larger fraction (its neighbours are the entries next to it), and a survivor is
// what ingest_decide_kernel makes of a passing row — counted as kept, its bit set in the methylated plane when its fraction
// reaches `high` (it cannot be unmethylated: the list path needs low < the adjacency threshold).  The classification pass
// then skips these rows altogether: no queue, no scattered loads there.
__global__ __launch_bounds__(256) void ingest_judge_kernel(CandList list, const unsigned long long *__restrict__ list_n, int adjacency, double high,
                                    const uint32_t *__restrict__ contig_chunk, IngestSlots sl, unsigned int *kept /*[contig][mod]*/,
                                    unsigned long long *n_kept, unsigned long long *n_classified) {
    uint64_t e_begin, e_end;
    wave_row_range(list_n[0], &e_begin, &e_end);
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long my_kept = 0, my_cls = 0;
    KeyCache kc;
    for (uint64_t e0 = e_begin; e0 < e_end; e0 += 64) {
        const uint64_t e = e0 + lane;
        const bool on = e < e_end;
        bool alive = false;
        uint32_t c = 0, pos = 0;
        int m = 0;
        uint8_t st = 0;
        unsigned long long fb = 0;
        if (on) {
            const unsigned long long key = list.key[e];
            c = (uint32_t)(key >> 32);
            pos = (uint32_t)key;
            st = list.strand[e];
            m = list.mod[e];
            fb = list.fbits[e];
            const unsigned long long hi_c = key & 0xFFFFFFFF00000000ull;
            const unsigned long long c_first = list.range[2 * (size_t)c], c_end = list.range[2 * (size_t)c + 1];
            const unsigned long long kmin = hi_c | (pos > (uint32_t)adjacency ? pos - (uint32_t)adjacency : 0u);
            const unsigned long long kmax = hi_c | (unsigned long long)min((unsigned long long)pos + (unsigned long long)adjacency, 0xFFFFFFFFull);
            unsigned long long mx = fb;
            for (unsigned long long sx = e; sx > c_first && list.key[sx - 1] >= kmin; --sx)
                if (list.strand[sx - 1] == st) mx = max(mx, list.fbits[sx - 1]);
            for (unsigned long long sx = e + 1; sx < c_end && list.key[sx] <= kmax; ++sx)
                if (list.strand[sx] == st) mx = max(mx, list.fbits[sx]);
            alive = mx == fb;
        }
        const bool listed = alive && m < NM_MAX_MOD_CODES;
        wave_add_keyed(kc, kept, listed ? c * NM_MAX_MOD_CODES + (uint32_t)m : 0u, listed, false, 1);
        my_kept += alive;
        const int slot = listed ? sl.slot_of_mod[m] : -1;
        const bool meth = slot >= 0 && __longlong_as_double((long long)fb) >= high;
        my_cls += meth;
        if (meth) {
            const uint64_t g = (uint64_t)contig_chunk[c] * CHUNK_BP + pos;
            atomicOr(sl.planes[slot][st == '+' ? 2 : 4] + (g >> 5), 1u << (g & 31));
        }
    }
    cache_flush(kc, kept, 1, lane);
    for (int d = 32; d; d >>= 1) {
        my_kept += __shfl_xor(my_kept, d);
        my_cls += __shfl_xor(my_cls, d);
    }
    if (lane == 0) {
        if (my_kept) atomicAdd(n_kept, my_kept);
        if (my_cls) atomicAdd(n_classified, my_cls);
    }
}

// compact planes of a slot from its general planes: a row counts on the compact path only when the base under it is the
// one its strand implies ('+' rows on the canonical base, '-' rows on its complement) — word-parallel, no atomics.  The same pass
// counts the bits of the four general planes (the confident rows; the duplicate-row check of nm_ingest_pileup: every classified row
// must have set its own bit) — four separate count launches per slot read the planes once more at 1 TB/s each (1.0 of the 13.7 ms
// of a 1e9-row ingest).
__global__ __launch_bounds__(256) void compact_planes_kernel(const uint32_t *__restrict__ H, const uint32_t *__restrict__ L,
                                                             const uint32_t *__restrict__ V, const uint32_t *__restrict__ MP,
                                                             const uint32_t *__restrict__ UP, const uint32_t *__restrict__ MM,
                                                             const uint32_t *__restrict__ UM, uint32_t can_l,
                                                             uint32_t *__restrict__ M, uint32_t *__restrict__ U, size_t n_words,
                                                             unsigned long long *n_meth_bits, unsigned long long *n_unmeth_bits) {
    unsigned long long meth = 0, unmeth = 0;
    // four words per lane and turn (a plane is a whole number of 256-word chunks): 16-byte requests, 32-bit partial counts
    const size_t n4 = n_words / 4;
    const uint4 *H4 = reinterpret_cast<const uint4 *>(H), *L4 = reinterpret_cast<const uint4 *>(L), *V4 = reinterpret_cast<const uint4 *>(V);
    const uint4 *MP4 = reinterpret_cast<const uint4 *>(MP), *UP4 = reinterpret_cast<const uint4 *>(UP), *MM4 = reinterpret_cast<const uint4 *>(MM),
                *UM4 = reinterpret_cast<const uint4 *>(UM);
    uint4 *M4 = reinterpret_cast<uint4 *>(M), *U4 = reinterpret_cast<uint4 *>(U);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 h = H4[i], l = L4[i], v = V4[i], mp = MP4[i], up = UP4[i], mm = MM4[i], um = UM4[i];
        uint4 m, u;
        uint32_t cm = 0, cu = 0;
#define NM_COMPACT_WORD(f)                                                                                              \
        {                                                                                                               \
            const uint32_t lsel = can_l ? l.f : ~l.f;         /* A = 00 / C = 01 on '+', T = 10 / G = 11 on '-' */       \
            const uint32_t fwd = v.f & ~h.f & lsel, rev = v.f & h.f & lsel;                                              \
            m.f = (mp.f & fwd) | (mm.f & rev);                                                                           \
            u.f = (up.f & fwd) | (um.f & rev);                                                                           \
            cm += __popc(mp.f) + __popc(mm.f);                                                                           \
            cu += __popc(up.f) + __popc(um.f);                                                                           \
        }
        NM_COMPACT_WORD(x) NM_COMPACT_WORD(y) NM_COMPACT_WORD(z) NM_COMPACT_WORD(w)
#undef NM_COMPACT_WORD
        M4[i] = m;
        U4[i] = u;
        meth += cm;
        unmeth += cu;
    }
    for (int d = 32; d; d >>= 1) {
        meth += __shfl_xor(meth, d);
        unmeth += __shfl_xor(unmeth, d);
    }
    // one pair of atomics per WORKGROUP: the waves of the whole grid adding to the same two addresses one after the other
    // was most of this kernel's time (32 768 same-address atomics ~ 0.25 ms)
    __shared__ unsigned long long part[2][4];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = meth; part[1][threadIdx.x >> 6] = unmeth; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long a = part[0][0] + part[0][1] + part[0][2] + part[0][3], b = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        if (a) atomicAdd(n_meth_bits, a);
        if (b) atomicAdd(n_unmeth_bits, b);
    }
}

}  // namespace

extern "C" {

// One part of a pileup (or all of it): `part_contigs` lists the engine contig ids whose rows are in this call (NULL: all
// contigs); the dense adjacency arrays cover only those.  `first`: clear the slots and the accumulated tables.
static int ingest_impl(nm_ctx *c, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_code,
                       const uint8_t *strand, const double *fraction_mod, const int32_t *nvalid_cov,
                       const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8], double low, double high,
                       int rows_on_device, int first, uint32_t n_part_contigs, const uint32_t *part_contigs,
                       uint64_t *n_kept, uint64_t *n_confident) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (!slot_of_mod || !canonical_of_mod || !n_kept || !n_confident) return fail(NM_EINVAL, "NULL argument");
    if (n_rows && (!contig_id || !position || !mod_code || !strand || !fraction_mod || !nvalid_cov)) return fail(NM_EINVAL, "NULL column");
    if (!(high > low)) return fail(NM_EINVAL, "high threshold must exceed low");
    HIP_TRY(hipSetDevice(c->device));
    // NM_INGEST_TIMING=1: where the host's time in this call goes (stderr, one line)
    const bool timing = getenv("NM_INGEST_TIMING") != nullptr;
    double t_marks[8] = {};
    int n_marks = 0;
    auto mark = [&] { if (timing && n_marks < 8) t_marks[n_marks++] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    mark();
    HIP_TRY(hipStreamSynchronize(c->stream));      // an asynchronous scoring launch may still read the planes replaced below
    {
        const int rcj = nmdetail::join_lanes(c);
        if (rcj) return rcj;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    mark();                                        // [1] what was queued before (the assembly's pack kernel) has finished
    const size_t words = plane_words(c);
    const size_t n_groups = (size_t)c->n_contigs * NM_MAX_MOD_CODES;        // reported: kept rows per (contig, code < 8)
    const size_t n_fgroups = (size_t)c->n_contigs * NM_CODE_STRIDE;         // frequency-filter groups: every code
    IngestSlots sl{};
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m) {
        sl.slot_of_mod[m] = slot_of_mod[m];
        if (slot_of_mod[m] < 0) continue;
        if (slot_of_mod[m] >= NM_MAX_MOD_SLOTS) return fail(NM_EINVAL, "slot %d >= %d", slot_of_mod[m], NM_MAX_MOD_SLOTS);
        if (canonical_of_mod[m] != 'A' && canonical_of_mod[m] != 'C') return fail(NM_EINVAL, "canonical base must be 'A' or 'C'");
        ModSlot &ms = c->slots[slot_of_mod[m]];
        drop_slot_ranks(ms);
        if (first) {
            HIP_TRY(alloc_slot_planes(ms, words));
            HIP_TRY(hipMemsetAsync(ms.planes[0], 0, words * 4 * 6, c->stream));
            ms.present = true;
            ms.canonical = canonical_of_mod[m];
            ms.low = low;
            ms.high = high;
            ms.n_rows = 0;
        } else if (!ms.present || ms.canonical != canonical_of_mod[m] || ms.low != low || ms.high != high || c->ing_slot_of_mod[m] != slot_of_mod[m]) {
            return fail(NM_ESTATE, "a later part must use the slots, canonical bases and thresholds of the first part");
        }
        for (int k = 0; k < 6; ++k) sl.planes[slot_of_mod[m]][k] = ms.planes[k];
        sl.can_l[slot_of_mod[m]] = canonical_of_mod[m] == 'C' ? 1u : 0u;
    }
    if (first) {
        drop_ingest_rows(c);
        c->ing_kept.assign(n_groups, 0);
        c->ing_total_kept = c->ing_classified = 0;
        for (int m = 0; m < NM_MAX_MOD_CODES; ++m) c->ing_slot_of_mod[m] = slot_of_mod[m];
    } else if (c->ing_kept.size() != n_groups) {
        return fail(NM_ESTATE, "no first part was ingested for this assembly");
    }
    // dense coordinates of this part: the listed contigs back to back, 128 zero positions between them
    std::vector<uint64_t> dense_off(c->n_contigs, ~0ull);
    uint64_t npos = 64;
    if (part_contigs) {
        for (uint32_t k = 0; k < n_part_contigs; ++k) {
            const uint32_t ci = part_contigs[k];
            if (ci >= c->n_contigs) return fail(NM_EINVAL, "part contig %u >= %u", ci, c->n_contigs);
            if (dense_off[ci] != ~0ull) continue;
            dense_off[ci] = npos;
            npos += c->contig_len[ci] + 128;
        }
    } else {
        for (uint32_t ci = 0; ci < c->n_contigs; ++ci) {
            dense_off[ci] = npos;
            npos += c->contig_len[ci] + 128;
        }
    }
    // device copies of the raw columns
    std::vector<void *> owned;
    // (a registered pool allocator does not synchronise like hipFree: nothing queued on the stream may still use the blocks)
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        for (void *p : owned) (void)nmdetail::dev_free(p);
    };
    // a failed ingest leaves half-written planes behind: the slots go back to "no pileup" so that scoring refuses them
    auto invalidate = [&]() {
        for (int m = 0; m < NM_MAX_MOD_CODES; ++m)
            if (slot_of_mod[m] >= 0 && slot_of_mod[m] < NM_MAX_MOD_SLOTS) {
                c->slots[slot_of_mod[m]].present = false;
                c->slots[slot_of_mod[m]].n_rows = 0;
            }
        drop_ingest_rows(c);
    };
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
            if (nmdetail::dev_malloc(&dst[k], std::max<size_t>(n_rows * esz[k], 16)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory for the raw pileup"); }
            owned.push_back(dst[k]);
            if (n_rows && hipMemcpyAsync(dst[k], src[k], n_rows * esz[k], hipMemcpyHostToDevice, c->stream) != hipSuccess) { cleanup(); return fail(NM_EHIP, "H2D copy of the raw pileup failed"); }
        }
        r.contig = (const uint32_t *)dst[0]; r.position = (const uint32_t *)dst[1]; r.mod = (const int8_t *)dst[2];
        r.strand = (const uint8_t *)dst[3]; r.frac = (const double *)dst[4]; r.nvalid = (const int32_t *)dst[5];
    }
    unsigned int *d_cnt = nullptr, *d_kept = nullptr;
    uint8_t *d_ok = nullptr;
    unsigned long long *d_dense = nullptr, *d_scalars = nullptr;
    uint64_t *d_dense_off = nullptr;
#define ING_ALLOC(ptr, bytes) do { void *q_ = nullptr; if (nmdetail::dev_malloc(&q_, (bytes)) != hipSuccess) { cleanup(); return fail(NM_ENOMEM, "out of device memory in nm_ingest_pileup (%zu bytes)", (size_t)(bytes)); } owned.push_back(q_); ptr = (decltype(ptr))q_; } while (0)
    ING_ALLOC(d_cnt, std::max<size_t>(n_fgroups, 1) * 2 * 4);
    ING_ALLOC(d_kept, std::max<size_t>(n_groups, 1) * 4);
    ING_ALLOC(d_ok, std::max<size_t>(n_fgroups, 1));
    ING_ALLOC(d_dense_off, (size_t)std::max(c->n_contigs, 1u) * 8);
    unsigned int *d_order = nullptr;   // [0]: order flags of the rows (1: a position decreases inside a run, 2: a contig in several runs); [1..]: runs per contig
    ING_ALLOC(d_order, ((size_t)c->n_contigs + 1) * 4);
    ING_ALLOC(d_scalars, 32);          // n_kept, n_classified (this part), population of the methylated / unmethylated general planes (all parts)
    const dim3 blk(256);
    const dim3 walk((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_rows + 255) / 256, 256 * 32)));      // waves walk contiguous row ranges
    const uint32_t n_waves = walk.x * 4;
    unsigned int *d_wave_cand = nullptr, *d_wave_n = nullptr;
    unsigned long long *d_wave_first = nullptr, *d_packed_first = nullptr;
    ING_ALLOC(d_wave_cand, (size_t)n_waves * 4);
    ING_ALLOC(d_wave_n, (size_t)n_waves * 4);
    ING_ALLOC(d_wave_first, ((size_t)n_waves + 1) * 8);
    ING_ALLOC(d_packed_first, ((size_t)n_waves + 1) * 8);
    hipError_t e = hipSuccess;
    mark();                                        // [2] slot planes, tables, scratch allocated
    nmdetail::busy_begin(c);
    e = hipMemsetAsync(d_cnt, 0, n_fgroups * 2 * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_kept, 0, n_groups * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_scalars, 0, 32, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_order, 0, ((size_t)c->n_contigs + 1) * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_wave_first, 0, ((size_t)n_waves + 1) * 8, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dense_off, dense_off.data(), (size_t)c->n_contigs * 8, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { cleanup(); return fail(NM_EHIP, "memset failed: %s", hipGetErrorString(e)); }
    if (n_rows && n_groups) {              // (a shard without contigs ignores every row)
        hipLaunchKernelGGL(ingest_count_kernel, walk, blk, 0, c->stream, r, c->n_contigs, c->d_contig_len, 5, 0.7, d_cnt, c->d_err, d_order, d_order + 1,
                           d_wave_cand);
        hipLaunchKernelGGL(ingest_group_kernel, dim3((unsigned)((n_fgroups + 255) / 256)), blk, 0, c->stream, (uint32_t)n_fgroups, d_cnt, 0.0001, 50u, d_ok, d_order + 1, d_order);
        // Rows in modkit's order (verified by the counting pass) and thresholds the way the reference has them: the adjacency
        // test runs on the compacted list of candidates.  Otherwise (or NM_INGEST_DENSE=1): one maximum per position and strand.
        bool use_list = low < 0.7 && getenv("NM_INGEST_DENSE") == nullptr && getenv("NM_INGEST_ATOMIC") == nullptr;
        unsigned long long cap = 0;
        if (use_list) {
            hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, c->stream, d_wave_cand, n_waves, d_wave_first);
            unsigned int order_flags = 0;
            e = hipMemcpyAsync(&order_flags, d_order, 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&cap, d_wave_first + n_waves, 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { nmdetail::busy_end(c); cleanup(); invalidate(); return fail(NM_EHIP, "ingest failed: %s", hipGetErrorString(e)); }
            use_list = order_flags == 0;
            mark();                                // [3] counting pass done (the host waits for the row order and the candidate count)
        }
        CandList lists[2] = {};
        if (use_list) {
            for (int k = 0; k < 2; ++k) {
                ING_ALLOC(lists[k].key, (size_t)(cap + 2) * 8);
                ING_ALLOC(lists[k].fbits, (size_t)(cap + 2) * 8);
                ING_ALLOC(lists[k].strand, (size_t)(cap + 16));
                ING_ALLOC(lists[k].mod, (size_t)(cap + 16));
            }
            // one pass over the rows classifies everything below the adjacency threshold and compacts the rest; the list is
            // then packed, indexed per contig and judged
            hipLaunchKernelGGL((ingest_decide_kernel<true, NM_DECIDE_U>), walk, blk, 0, c->stream, r, 5, d_ok, c->d_contig_chunk, d_dense_off, nullptr, nullptr, 8, 0.7, low,
                               high, sl, d_order, d_kept, d_scalars, d_scalars + 1, d_wave_first, lists[0], d_wave_n, c->d_err);
            hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, c->stream, d_wave_n, n_waves, d_packed_first);
            hipLaunchKernelGGL(ingest_pack_kernel, dim3((n_waves + 3) / 4), blk, 0, c->stream, n_waves, d_wave_first, d_wave_n, d_packed_first, lists[0], lists[1]);
            ING_ALLOC(lists[1].range, (size_t)std::max(c->n_contigs, 1u) * 16);
            e = hipMemsetAsync(lists[1].range, 0, (size_t)std::max(c->n_contigs, 1u) * 16, c->stream);
            if (e != hipSuccess) { nmdetail::busy_end(c); cleanup(); invalidate(); return fail(NM_EHIP, "memset failed: %s", hipGetErrorString(e)); }
            const dim3 over_list((unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>((cap + 255) / 256, 8192)));
            hipLaunchKernelGGL(ingest_ranges_kernel, over_list, blk, 0, c->stream, lists[1], d_packed_first + n_waves);
            hipLaunchKernelGGL(ingest_judge_kernel, over_list, blk, 0, c->stream, lists[1], d_packed_first + n_waves, 8, high, c->d_contig_chunk, sl, d_kept,
                               d_scalars, d_scalars + 1);
        } else {
            ING_ALLOC(d_dense, npos * 8 * 2);
            e = hipMemsetAsync(d_dense, 0, npos * 8 * 2, c->stream);
            if (e != hipSuccess) { nmdetail::busy_end(c); cleanup(); invalidate(); return fail(NM_EHIP, "memset failed: %s", hipGetErrorString(e)); }
            hipLaunchKernelGGL(ingest_scatter_kernel, walk, blk, 0, c->stream, r, 5, d_ok, d_dense_off, d_dense, d_dense + npos, 0.7, c->d_err);
            if (getenv("NM_INGEST_ATOMIC")) (void)hipMemsetAsync(d_order, 0xFF, 4, c->stream);       // A/B switch: never take the store path
            hipLaunchKernelGGL((ingest_decide_kernel<false, 4>), walk, blk, 0, c->stream, r, 5, d_ok,
                               c->d_contig_chunk, d_dense_off, d_dense, d_dense + npos, 8, 0.7, low, high, sl, d_order, d_kept, d_scalars, d_scalars + 1,
                               nullptr, CandList{}, nullptr, c->d_err);
        }
    }
#undef ING_ALLOC
    // population of the general planes (all parts so far): the confident rows, and the duplicate check — every
    // classified row must have set its own bit; then the compact planes
    bool seen[NM_MAX_MOD_SLOTS] = {};
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m) {
        const int slot = slot_of_mod[m];
        if (slot < 0 || seen[slot]) continue;
        seen[slot] = true;
        uint32_t *const *pl = c->slots[slot].planes;
        // bits of MP, MM -> scalar 2 (the confident rows), of UP, UM -> scalar 3
        hipLaunchKernelGGL(compact_planes_kernel, dim3(2048), blk, 0, c->stream, c->dH, c->dL, c->dV, pl[2], pl[3], pl[4], pl[5],
                           sl.can_l[slot], pl[0], pl[1], words, d_scalars + 2, d_scalars + 3);
    }
    e = hipGetLastError();
    nmdetail::busy_end(c);
    mark();                                        // [4] everything else queued
    if (e != hipSuccess) { cleanup(); invalidate(); return fail(NM_EHIP, "ingest launch failed: %s", hipGetErrorString(e)); }
    unsigned long long scal[4] = {0, 0, 0, 0};
    unsigned int err = 0;
    std::vector<uint32_t> part_kept(n_groups);
    e = hipMemcpyAsync(scal, d_scalars, 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(part_kept.data(), d_kept, n_groups * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    mark();                                        // [5] the device is done
    cleanup();
    mark();                                        // [6] scratch given back
    if (timing && n_marks == 7)
        fprintf(stderr, "[nm_ingest] %llu rows: waiting for earlier work %.2f ms, allocations %.2f ms, clears + counting pass + wait %.2f ms, "
                        "queueing the rest %.2f ms, waiting for it %.2f ms, freeing scratch %.2f ms\n", (unsigned long long)n_rows,
                (t_marks[1] - t_marks[0]) * 1e3, (t_marks[2] - t_marks[1]) * 1e3, (t_marks[3] - t_marks[2]) * 1e3, (t_marks[4] - t_marks[3]) * 1e3,
                (t_marks[5] - t_marks[4]) * 1e3, (t_marks[6] - t_marks[5]) * 1e3);
    if (e != hipSuccess) { invalidate(); return fail(NM_EHIP, "ingest failed: %s", hipGetErrorString(e)); }
    if (err & 1u) { invalidate(); return fail(NM_EINVAL, "pileup row with contig_id / position / mod code outside the uploaded assembly"); }
    if (err & 8u) { invalidate(); return fail(NM_EINVAL, "pileup row of a contig that is not listed in part_contigs"); }
    c->ing_total_kept += scal[0];
    c->ing_classified += scal[1];
    c->ing_nconf = scal[2];
    for (size_t i = 0; i < n_groups; ++i) c->ing_kept[i] += part_kept[i];
    if ((err & 4u) || c->ing_classified != scal[2] + scal[3]) { invalidate(); return fail(NM_EINVAL, "duplicate (contig, position, strand) rows within one modification type: the reference's np.isin(assume_unique=True) requires unique positions (find_motifs_bin.py:1258)"); }
    for (int m = 0; m < NM_MAX_MOD_CODES; ++m)
        if (slot_of_mod[m] >= 0) c->slots[slot_of_mod[m]].n_rows = c->ing_total_kept;
    *n_kept = c->ing_total_kept;
    *n_confident = c->ing_nconf;
    return NM_OK;
}

int nm_ingest_pileup(nm_ctx *c, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_code,
                     const uint8_t *strand, const double *fraction_mod, const int32_t *nvalid_cov,
                     const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8], double low, double high,
                     int rows_on_device, uint64_t *n_kept, uint64_t *n_confident) {
    return ingest_impl(c, n_rows, contig_id, position, mod_code, strand, fraction_mod, nvalid_cov, slot_of_mod, canonical_of_mod,
                       low, high, rows_on_device, 1, 0, nullptr, n_kept, n_confident);
}

int nm_ingest_pileup_part(nm_ctx *c, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_code,
                          const uint8_t *strand, const double *fraction_mod, const int32_t *nvalid_cov,
                          const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8], double low, double high,
                          int rows_on_device, int first, uint32_t n_part_contigs, const uint32_t *part_contigs,
                          uint64_t *n_kept, uint64_t *n_confident) {
    if (n_part_contigs && !part_contigs) return fail(NM_EINVAL, "part_contigs is NULL");
    static const uint32_t none = 0;
    return ingest_impl(c, n_rows, contig_id, position, mod_code, strand, fraction_mod, nvalid_cov, slot_of_mod, canonical_of_mod,
                       low, high, rows_on_device, first, n_part_contigs, part_contigs ? part_contigs : &none, n_kept, n_confident);
}

int nm_ingest_results(nm_ctx *c, uint32_t *conf_contig, uint32_t *conf_position, uint8_t *conf_strand, int8_t *conf_mod,
                      uint64_t capacity, uint32_t *kept_per_contig_mod /*[n_contigs][8]*/) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    const size_t n = c->ing_nconf;
    if (conf_contig || conf_position || conf_strand || conf_mod) {          // all NULL: only the kept table is wanted
        if (capacity < n) return fail(NM_ERANGE, "capacity %llu < %zu confident rows", (unsigned long long)capacity, n);
        if (n) {
            if (!conf_contig || !conf_position || !conf_strand || !conf_mod) return fail(NM_EINVAL, "NULL argument");
            // the confident rows ARE the set bits of the slots' MP ('+') and MM ('-') planes: enumerate them on the host
            // (rare path: the device-side window extraction never needs the list)
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const size_t words = plane_words(c);
            std::vector<uint32_t> plane(words);
            size_t at = 0;
            for (int m = 0; m < NM_MAX_MOD_CODES; ++m) {
                const int slot = c->ing_slot_of_mod[m];
                if (slot < 0 || !c->slots[slot].present) continue;
                for (int strand = 0; strand < 2; ++strand) {
                    HIP_TRY(hipMemcpy(plane.data(), c->slots[slot].planes[strand ? 4 : 2], words * 4, hipMemcpyDeviceToHost));
                    for (uint32_t ci = 0; ci < c->n_contigs; ++ci) {
                        const size_t w0 = (size_t)c->contig_chunk[ci] * CHUNK_WORDS, nw = (size_t)((c->contig_len[ci] + 31) / 32);
                        for (size_t w = 0; w < nw; ++w) {
                            uint32_t x = plane[w0 + w];
                            while (x) {
                                if (at >= n) return fail(NM_ESTATE, "the slot's planes changed since nm_ingest_pileup");
                                conf_contig[at] = ci;
                                conf_position[at] = (uint32_t)(w * 32 + (uint32_t)__builtin_ctz(x));
                                conf_strand[at] = strand ? '-' : '+';
                                conf_mod[at] = (int8_t)m;
                                ++at;
                                x &= x - 1;
                            }
                        }
                    }
                }
            }
            if (at != n) return fail(NM_ESTATE, "the slot's planes changed since nm_ingest_pileup");
        }
    }
    if (kept_per_contig_mod) memcpy(kept_per_contig_mod, c->ing_kept.data(), c->ing_kept.size() * 4);
    return NM_OK;
}

}  // extern "C"
