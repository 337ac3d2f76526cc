// libnmscan — window engine and window extraction: the per-expansion work of the greedy search on bit planes over
// windows, the gather of those windows from the resident planes and the background sample (C ABI: nm_win_*,
// nm_contig_base_counts, nm_bg_counts, nm_methylated_row_counts; include/nmscan.h).
#include <chrono>
#include <cmath>
#include <thread>

#include "nmscan_device.h"

using namespace nmdetail;

void nm_mt_outputs(uint32_t mt_state[625], uint64_t n_draws, uint32_t *out);   // nmhost.cpp: genrand_uint32 outputs, state advanced in place

namespace {

// ------------------------------------------------------------------------------------------------------
// Window engine: the methylated-site windows of every (bin, mod type) search stay on the device as bit planes over
// WINDOWS (bit i of a plane = window i), so that the per-expansion work of the search — filter_sequence_matches
// (seq.py:499-524) + DNAarray.pssm (seq.py:526-537) — is the same AND / popcount pattern as the genome scan.
//   per task: plane1[col][b] = windows whose base at column col is exactly b (b in A, C, G, T),
//             planeN[col]   = windows with N there (one-hot 1111: counts for all four rows, matches only '.'),
//             alive         = windows not yet removed by an accepted / dead-end motif (find_motifs_bin.py:801-806).
// ------------------------------------------------------------------------------------------------------
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
// Work split: blockIdx.x = request, blockIdx.y = group of WIN_COL_GROUP columns whose counters this workgroup keeps in
// registers (a lane owns 32 windows per word it visits; the 4 x WIN_COL_GROUP per-lane counters are reduced over the wave
// ONCE, after the last word — reducing 164 counters per visited word, as the first version did, was the largest kernel
// of the whole search), blockIdx.z = slice of the task's words.  Every column group recomputes the match mask (a few
// plane words per specified motif position); group 0 also counts the active windows and serves the removals.
constexpr int WIN_COL_GROUP = 8;

__global__ __launch_bounds__(256) void win_request_kernel(const WinTask *__restrict__ tasks, uint32_t n_req,
                                                          const uint32_t *__restrict__ req_task,
                                                          const uint8_t *__restrict__ req_kind,
                                                          const uint8_t *__restrict__ req_sets /*[n_req][WIN_MAX_W]*/,
                                                          const uint32_t *__restrict__ planes, uint32_t *alive,
                                                          int *__restrict__ out, uint32_t ws /*columns per row of sets / counts*/) {
    __shared__ uint8_t mset[WIN_MAX_W];
    const uint32_t r = blockIdx.x;
    const WinTask t = tasks[req_task[r]];
    const uint32_t kind = req_kind[r];
    const uint32_t col0 = blockIdx.y * WIN_COL_GROUP;
    if (col0 >= t.width || (kind == 1 && blockIdx.y != 0)) return;
    const uint32_t out_stride = 2 + 4 * ws;
    if (threadIdx.x < WIN_MAX_W) mset[threadIdx.x] = threadIdx.x < t.width ? req_sets[(size_t)r * ws + threadIdx.x] : 15;
    __syncthreads();
    const uint32_t *pl = planes + t.plane_off;
    uint32_t *al = alive + t.alive_off;
    const uint64_t nw = t.nw;
    int cnt[WIN_COL_GROUP][4];
#pragma unroll
    for (int c = 0; c < WIN_COL_GROUP; ++c)
#pragma unroll
        for (int b = 0; b < 4; ++b) cnt[c][b] = 0;
    int n_a = 0, n_b = 0;
    for (uint32_t w = blockIdx.z * blockDim.x + threadIdx.x; w < t.nw; w += gridDim.z * blockDim.x) {
        uint32_t match = 0xFFFFFFFFu;
        for (uint32_t col = 0; col < t.width; ++col) {
            const uint32_t m = mset[col];
            if (m == 15) continue;
            const uint32_t *p = pl + (uint64_t)col * 5 * nw + w;
            uint32_t ok = 0;
            if (m & 1) ok |= p[0];
            if (m & 2) ok |= p[nw];
            if (m & 4) ok |= p[2 * nw];
            if (m & 8) ok |= p[3 * nw];
            match &= ok;
        }
        const uint32_t a = al[w];
        if (kind == 1) {
            const uint32_t na = a & ~match;
            al[w] = na;
            n_a += __popc(a);
            n_b += __popc(na);
            continue;
        }
        const uint32_t active = a & match;
        if (!active) continue;
        n_a += __popc(active);
#pragma unroll
        for (int c = 0; c < WIN_COL_GROUP; ++c) {
            if (col0 + c >= t.width) break;                            // uniform
            const uint32_t *p = pl + (uint64_t)(col0 + c) * 5 * nw + w;
            const uint32_t any = p[4 * nw];                            // an N window counts for all four letters
            cnt[c][0] += __popc(active & (p[0] | any));
            cnt[c][1] += __popc(active & (p[nw] | any));
            cnt[c][2] += __popc(active & (p[2 * nw] | any));
            cnt[c][3] += __popc(active & (p[3 * nw] | any));
        }
    }
    // one reduction per counter and wave, one atomic per non-zero sum
    int *o = out + (size_t)r * out_stride;
    const int lane = threadIdx.x & 63;
    auto wave_sum = [](int v) {
#pragma unroll
        for (int s = 32; s; s >>= 1) v += __shfl_xor(v, s);
        return v;
    };
    if (blockIdx.y == 0) {
        n_a = wave_sum(n_a);
        n_b = wave_sum(n_b);
        if (lane == 0 && n_a) atomicAdd(&o[0], n_a);
        if (lane == 0 && n_b) atomicAdd(&o[1], n_b);
    }
    if (kind == 1) return;
#pragma unroll
    for (int c = 0; c < WIN_COL_GROUP; ++c) {
        if (col0 + c >= t.width) break;
        const int ca = wave_sum(cnt[c][0]), cc = wave_sum(cnt[c][1]), cg = wave_sum(cnt[c][2]), ct = wave_sum(cnt[c][3]);
        if (lane == 0) {                                               // row order A, T, G, C (constants.py:1)
            if (ca) atomicAdd(&o[2 + 0 * ws + col0 + c], ca);
            if (ct) atomicAdd(&o[2 + 1 * ws + col0 + c], ct);
            if (cg) atomicAdd(&o[2 + 2 * ws + col0 + c], cg);
            if (cc) atomicAdd(&o[2 + 3 * ws + col0 + c], cc);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Window extraction on the device (find_motifs_bin.py:625-686): gather of the methylation windows from the resident
// sequence planes, and the background sample (seq.py:202-225) as "k-th position whose base is X" through a rank
// table (popcount prefix per 512-bp block, per contig).
// ------------------------------------------------------------------------------------------------------
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

// W <= 192 bits starting at global bit g, as three 64-bit pieces
struct Field3 { uint64_t w[3]; };

__device__ __forceinline__ Field3 plane_field3(const uint32_t *__restrict__ P, uint64_t g, uint32_t n) {
    Field3 f;
#pragma unroll
    for (int k = 0; k < 3; ++k) f.w[k] = n > 64u * k ? plane_field(P, g + 64u * k, min(64u, n - 64u * k)) : 0ull;
    return f;
}

__device__ __forceinline__ uint32_t field_bit(const Field3 &f, uint32_t j) {
    const uint64_t w = j < 64 ? f.w[0] : (j < 128 ? f.w[1] : f.w[2]);
    return (uint32_t)(w >> (j & 63)) & 1u;
}

// set bits among the n (<= 192) bits of a plane from bit g on
__device__ __forceinline__ uint32_t plane_popc(const uint32_t *__restrict__ P, uint64_t g, uint32_t n) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < n; k += 64) c += (uint32_t)__popcll(plane_field(P, g + k, min(64u, n - k)));
    return c;
}

// positions among the n (<= 192) from bit g on whose base is b
__device__ __forceinline__ uint32_t base_popc(const Planes &s, uint64_t g, uint32_t n, int b) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < n; k += 64) {
        const uint32_t m = min(64u, n - k);
        const uint64_t h = plane_field(s.H, g + k, m), l = plane_field(s.L, g + k, m), v = plane_field(s.V, g + k, m);
        c += (uint32_t)__popcll(v & ((b >= 2) ? h : ~h) & ((b == 1 || b == 2) ? l : ~l));
    }
    return c;
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
    if (pad) n -= base_popc(s, g0, pad, b) + base_popc(s, g0 + len - pad, pad, b);
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
    const uint64_t hp = plane_popc(MP, g0, pad + 1), hm = plane_popc(MM, g0, pad + 1);
    const uint64_t tp = pad ? plane_popc(MP, g0 + len - pad, pad) : 0;
    const uint64_t tm = pad ? plane_popc(MM, g0 + len - pad, pad) : 0;
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
                                                        unsigned long long *__restrict__ out, uint32_t ws /*columns per output row*/,
                                                        unsigned int *err) {
    __shared__ uint32_t cnt[4 * WIN_MAX_W];
    const BgBlock blk = blocks[blockIdx.x];
    const uint64_t begin = ((uint64_t)blk.begin_hi << 32) | blk.begin_lo;
    const uint32_t W = 2 * pad + 1, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 4 * WIN_MAX_W; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < blk.count; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        bool on = i < blk.count;
        Field3 fh{}, fl{}, fv{};
        if (on) {
            const uint32_t ci = sample_contig[begin + i];
            const uint32_t c0 = contig_chunk[ci];
            const uint64_t g0 = (uint64_t)c0 * CHUNK_BP;
            const uint32_t nblk = (uint32_t)((contig_len[ci] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
            uint32_t k = sample_rank[begin + i];
            if (pad) k += base_popc(s, g0, pad, b);
            const uint64_t centre = select_kth(s, nullptr, b, rank + (size_t)c0 * RANK_PER_CHUNK, c0, nblk, k);
            if (centre == ~0ull || centre < g0 + pad) {
                atomicOr(err, 4u);                       // rank beyond the contig's valid starts
                on = false;
            } else {
                fh = plane_field3(s.H, centre - pad, W);
                fl = plane_field3(s.L, centre - pad, W);
                fv = plane_field3(s.V, centre - pad, W);
            }
        }
        for (uint32_t col = 0; col < W; ++col) {
            const bool v = on && field_bit(fv, col), h = field_bit(fh, col), l = field_bit(fl, col);
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
    unsigned long long *o = out + (size_t)blk.task * 4 * ws;
    for (uint32_t i = threadIdx.x; i < 4 * WIN_MAX_W; i += blockDim.x)
        if (cnt[i]) atomicAdd(&o[(i / WIN_MAX_W) * ws + i % WIN_MAX_W], (unsigned long long)cnt[i]);
}

// Methylation windows of one task from the sequence planes: a wave packs 64 windows; per column it ballots the five
// window planes (A, C, G, T, N) and lanes 0 / 1 store the two words.  Minus rows are reverse-complemented
// (complement = flip H in the (H, L) code; N stays N).
__device__ __forceinline__ void pack_windows(const WinTask &t, const Planes &s, uint32_t wave, uint32_t lane, bool on,
                                             uint64_t centre, bool minus, uint32_t pad, uint32_t *__restrict__ planes,
                                             uint32_t *__restrict__ alive) {
    const uint32_t W = t.width;
    Field3 fh{}, fl{}, fv{};
    if (on) {
        const uint64_t g = centre - pad;
        fh = plane_field3(s.H, g, W);
        fl = plane_field3(s.L, g, W);
        fv = plane_field3(s.V, g, W);
    }
    const uint32_t w = wave * 2 + lane;                   // word written by lanes 0 and 1
    const bool writer = lane < 2 && w < t.nw;
    const uint32_t shift = 32 * (lane & 1);
    for (uint32_t col = 0; col < W; ++col) {
        // minus rows: the window is read backwards and complemented (flip H in the (H, L) code; N stays N)
        const uint32_t j = minus ? W - 1 - col : col;
        const bool v = field_bit(fv, j), l = field_bit(fl, j);
        const bool h = (field_bit(fh, j) != 0) != (minus && v);
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

// The methylation windows of ALL tasks in one launch (nm_plan_windows): workgroup b serves the 256 windows
// [first, first + 256) of task blocks[b].task; the task's (contig, strand) segments are segs[seg_begin .. seg_begin + n_seg).
struct WinBlock { uint32_t task, first, seg_begin, n_seg, slot; };
struct SlotRows { const uint32_t *MP, *MM, *rank_p, *rank_m; };
struct AllSlotRows { SlotRows s[NM_MAX_MOD_SLOTS]; };

__global__ __launch_bounds__(256) void win_gather_all_kernel(const WinTask *__restrict__ tasks, const WinBlock *__restrict__ blocks,
                                                             Planes s, AllSlotRows rows, const uint32_t *__restrict__ contig_chunk,
                                                             const uint64_t *__restrict__ contig_len, const WinSegment *__restrict__ segs,
                                                             uint32_t pad, uint32_t *__restrict__ planes, uint32_t *__restrict__ alive,
                                                             unsigned int *err) {
    const WinBlock blk = blocks[blockIdx.x];
    const WinTask t = tasks[blk.task];
    const SlotRows sr = rows.s[blk.slot];
    const WinSegment *seg = segs + blk.seg_begin;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blk.first / 64 + (threadIdx.x >> 6);
    const uint64_t i = (uint64_t)wave * 64 + lane;
    if ((uint64_t)wave * 64 >= t.n) return;
    bool on = i < t.n, minus = false;
    uint64_t centre = 0;
    if (on) {
        uint32_t lo = 0, hi = blk.n_seg - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (seg[mid].dst_start <= (uint32_t)i) lo = mid; else hi = mid - 1;
        }
        const WinSegment sg = seg[lo];
        minus = sg.minus != 0;
        const uint32_t c0 = contig_chunk[sg.contig];
        const uint32_t nblk = (uint32_t)((contig_len[sg.contig] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
        centre = select_kth(s, minus ? sr.MM : sr.MP, -1, (minus ? sr.rank_m : sr.rank_p) + (size_t)c0 * RANK_PER_CHUNK, c0, nblk,
                            (uint32_t)i - sg.dst_start + sg.head);
        if (centre == ~0ull) {
            atomicOr(err, 8u);
            on = false;
        }
    }
    pack_windows(t, s, wave, lane, on, centre, minus, pad, planes, alive);
}

}  // namespace

extern "C" {

// ---- window engine host side ------------------------------------------------------------------------
static int win_grow(nm_ctx *c, uint32_t **buf, uint64_t *cap, uint64_t used, uint64_t need_words) {
    if (used + need_words <= *cap) return NM_OK;
    const uint64_t ncap = std::max<uint64_t>((used + need_words) * 3 / 2, 1u << 20);
    uint32_t *nb = nullptr;
    HIP_TRY(nmdetail::dev_malloc(&nb, ncap * 4));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));       // window batches in flight (batch_stream)
    HIP_TRY(nmdetail::sync_flight_streams(c));
    if (*buf && used) HIP_TRY(hipMemcpy(nb, *buf, used * 4, hipMemcpyDeviceToDevice));
    if (*buf) (void)nmdetail::dev_free(*buf);
    *buf = nb;
    *cap = ncap;
    return NM_OK;
}

int nm_win_clear(nm_ctx *c) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
    HIP_TRY(nmdetail::sync_flight_streams(c));
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
        HIP_TRY(nmdetail::dev_malloc(&d_sets, (size_t)n_windows * width));
        HIP_TRY(hipMemcpyAsync(d_sets, sets, (size_t)n_windows * width, hipMemcpyHostToDevice, c->stream));
        if (t.nw) hipLaunchKernelGGL(win_pack_kernel, dim3((t.nw + 255) / 256, width), dim3(256), 0, c->stream, t, d_sets,
                           c->d_win_planes, c->d_win_alive);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)nmdetail::dev_free(d_sets);
    }
    c->win_planes_used += (uint64_t)width * 5 * t.nw;
    c->win_alive_used += t.nw;
    *task_id = (uint32_t)c->win_tasks.size();
    c->win_tasks.push_back(t);
    c->win_tasks_dirty = true;
    return NM_OK;
}

// The stream window batches run on.  A round of the search enqueues its window batch and its scoring batch back to back and
// collects the windows first: on ONE stream the scoring chain (copy, compile, clear, kernel, copy back) queued behind the
// window chain (copy, clear, kernel, copy back) — some ten dependent commands of 5-8 us each per round; the two batches ask
// about different tasks and touch different memory (the staging pairs are separate, the window planes belong to the window
// engine), so the window chain goes to the context's second stream and the two overlap.  Window state is only ever changed
// by these batches (in order, on that stream) and by the set-up calls, which finish on the ctx stream before a batch can
// start (win_tasks_dirty / nm_plan_windows synchronise) and wait for this stream before they touch the planes.
// NM_WIN_STREAM=0: everything on the ctx stream (A/B, tools/gpu_r4p.sh).
static hipStream_t batch_stream(nm_ctx *c, int flight = 0) {
    static const bool second = getenv("NM_WIN_STREAM") == nullptr || atoi(getenv("NM_WIN_STREAM")) != 0;
    if (flight >= 1 && flight < NM_FLIGHTS && second) {       // the further flights of the search: a stream of their own each, made on first use
        hipStream_t &fs = c->flight_stream[flight - 1];
        if (!fs) {
            // A plain stream.  NM_FLIGHT_PRIORITY=1 gives it a priority — and with it a hardware queue — of its own (A/B, tools/gpu_r5l.sh):
            // the search gets SLOWER (window batches 12.8 -> 22 ms at 1 Gbp, as with GPU_MAX_HW_QUEUES=8): two chains of small kernels
            // that run truly side by side lengthen each other more than taking turns on one queue costs
            int least = 0, greatest = 0;
            const bool prio = getenv("NM_FLIGHT_PRIORITY") != nullptr && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
            const hipError_t e = prio ? hipStreamCreateWithPriority(&fs, hipStreamNonBlocking, greatest)
                                      : hipStreamCreateWithFlags(&fs, hipStreamNonBlocking);
            if (e != hipSuccess) fs = nullptr;
        }
        if (fs) return fs;
    }
    return second && c->copy_stream ? c->copy_stream : c->stream;
}

// (nmdetail::busy_begin / busy_end bracket work on the ctx stream; the same bookkeeping for a batch on another stream)
static void busy_begin_on(nm_ctx *c, hipStream_t st) {
    if (st == c->stream) { nmdetail::busy_begin(c); return; }
    if (!c->ev_collect_all) return;
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= 65536) return;
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        c->ev_pool.emplace_back(a, b);
    }
    (void)hipEventRecord(c->ev_pool[c->ev_used].first, st);
    c->busy_open = true;
}

static void busy_end_on(nm_ctx *c, hipStream_t st) {
    if (st == c->stream) { nmdetail::busy_end(c); return; }
    if (!c->ev_collect_all || !c->busy_open) return;
    (void)hipEventRecord(c->ev_pool[c->ev_used].second, st);
    c->ev_used += 1;
    c->busy_open = false;
}


// ------------------------------------------------------------------------------------------------------
// Speculative children of the search (round 5).  A task of the lock-step search alternates two dependent round trips: the
// window counts of its current motif, then — once the host has picked the column of the largest KL divergence against the
// bin's background and the bases that pass the two frequency filters (find_motifs_bin.py:957-1023) — the counts of the
// up-to-four children there.  This kernel makes the same pick ON THE DEVICE right behind win_request_kernel and writes the
// children as candidate records of a scoring batch that runs in the same chain of launches (nmdetail::score_batch_spec_begin):
// when the host's own, exact pick agrees (it recomputes it the way scipy does and stays authoritative), the children's counts
// are already there and the second round trip is not made.  A pick that differs — double-precision log of the device against
// the host's on a near-tie — only costs the round trip it would have cost anyway.
// One wave per request; lane j holds column j.
// ------------------------------------------------------------------------------------------------------
struct SpecParams {
    uint32_t n_req, ws, width, pad;
    double min_kl, freq_threshold;
};

__global__ __launch_bounds__(256) void spec_children_kernel(SpecParams P, const uint8_t *__restrict__ req_kind, const uint8_t *__restrict__ req_sets,
                                                            const uint32_t *__restrict__ req_search_task, const uint32_t *__restrict__ req_entry,
                                                            const int *__restrict__ win_out, const double *__restrict__ bg /*[task][4][width]*/,
                                                            int *__restrict__ spec_info /*[n_req][2]: column (-1: none), base rows as bits*/,
                                                            nmdetail::SpecCompile cc, uint32_t mask_stride) {
    __shared__ uint32_t lds_prog[4][4 * 64];                         // per wave: its four children's programs while they are built
    const uint32_t wave = threadIdx.x >> 6, r = blockIdx.x * 4 + wave, lane = threadIdx.x & 63;
    if (r >= P.n_req) return;
    const uint32_t W = P.width;
    const int *o = win_out + (size_t)r * (2 + 4 * P.ws);
    const uint8_t *sets = req_sets + (size_t)r * P.ws;
    const int n_active = o[0];
    const double *q = bg + (size_t)req_search_task[r] * 4 * W;
    // KL(column j) = sum_r rel_entr(p_r / sum p, q_r / sum q)  (scipy.stats.entropy), specified columns count as 0
    double kl = 0.0;
    bool is_nan = false;
    const uint32_t my_set = lane < W ? sets[lane] : 0u;
    if (lane < W && my_set == 15u && req_kind[r] == 0 && n_active > 0) {
        double p[4], qq[4];
        for (int k = 0; k < 4; ++k) { p[k] = (double)o[2 + k * P.ws + lane] / (double)n_active; qq[k] = q[k * W + lane]; }
        const double sp = ((p[0] + p[1]) + p[2]) + p[3], sq = ((qq[0] + qq[1]) + qq[2]) + qq[3];
        double e[4];
        for (int k = 0; k < 4; ++k) {
            const double x = p[k] / sp, y = qq[k] / sq;
            e[k] = (x > 0 && y > 0) ? x * log(x / y) : (x == 0 && y >= 0) ? 0.0 : INFINITY;
            if (x != x || y != y) e[k] = x + y;                       // NaN in, NaN out
        }
        kl = ((e[0] + e[1]) + e[2]) + e[3];
        is_nan = kl != kl;
    }
    // np.argmax: a NaN wins, otherwise the first maximum
    const unsigned long long nan_lanes = __ballot(is_nan);
    int col;
    double best;
    if (nan_lanes) {
        col = __ffsll((long long)nan_lanes) - 1;
        best = 0.0;
    } else {
        best = kl;
        col = (int)lane;
        for (int s = 32; s; s >>= 1) {
            const double ob = __shfl_xor(best, s);
            const int oc = __shfl_xor(col, s);
            if (ob > best || (ob == best && oc < col)) { best = ob; col = oc; }
        }
    }
    // the bases at that column: frequency above half the background's and above the threshold (rows A, T, G, C)
    uint32_t bases = 0;
    const bool any_dot = __ballot(lane < W && my_set == 15u) != 0;
    if (req_kind[r] == 0 && n_active > 0 && any_dot && (nan_lanes || !(best < P.min_kl))) {
        for (int k = 0; k < 4; ++k) {
            const double m = (double)o[2 + k * P.ws + col] / (double)n_active;
            if (m > q[k * W + col] * 0.5 && m > P.freq_threshold) bases |= 1u << k;
        }
    }
    const uint32_t nv = __popc(bases);
    const uint32_t entry = req_entry[r];
    if (lane == 0) {
        spec_info[2 * r] = nv ? col : -1;
        spec_info[2 * r + 1] = (int)bases;
        cc.range[entry].y = nv;                                       // the group's candidates: its valid children, packed first
    }
    // the four records of the request: child c = the c-th passing base (rows in order), the others a harmless one-letter motif
    const uint32_t row_bit[4] = {NM_BASE_A, NM_BASE_T, NM_BASE_G, NM_BASE_C};
    uint32_t left = bases;
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t at = cc.pos_of[4 * r + c];
        uint8_t *m = cc.masks + cc.rec[at].mask_off;
        if (c < nv) {
            const int k = __ffs(left) - 1;
            left &= left - 1;
            const uint32_t set = lane < W ? ((int)lane == col ? row_bit[k] : my_set) : 15u;
            const unsigned long long spec_cols = __ballot(lane < W && set != 15u);     // never empty: the modified position is specified
            const int lo = __ffsll((long long)spec_cols) - 1, hi = 63 - __clzll((long long)spec_cols);
            if ((int)lane >= lo && (int)lane <= hi) m[lane - lo] = (uint8_t)set;
            if (lane == 0) { cc.rec[at].len = (uint8_t)(hi - lo + 1); cc.rec[at].modpos = (uint8_t)(P.pad - lo); }
        } else if (lane == 0) {
            m[0] = (uint8_t)sets[P.pad];
            cc.rec[at].len = 1;
            cc.rec[at].modpos = 0;
        }
    }
    // ---- the children's programs (what compile_common_kernel does for a host-written batch): built in LDS, one lane per child; the
    // group's common constraints are factored out by the WAVE — lane i owns dword i of the programs (common_one of nmscan_device.h
    // walks them on one thread: 12 us of dependent LDS round trips per request) —, then the programs go to the table.  The four
    // records sit next to each other (the batch is sorted by (slot, bin) and a group holds exactly this request's children).
    __threadfence();                                                  // the records and masks written above are read back below
    const uint32_t first = cc.pos_of[4 * r];
    uint32_t *wp = lds_prog[wave];
    if (cc.pdw <= 64) {
        const uint32_t pdw = cc.pdw, sdw = pdw / 2;
        if (lane < 4) nmdetail::compile_one(first + lane, cc.rec, cc.masks, wp, cc.wide, cc.np, cc.fold_modpos, lane);
        __threadfence_block();
        uint4 rg = cc.range[entry];
        rg.y = nv;
        rg.z = 0xFFFFFFFFu;
        rg.w = 0;
        if (cc.common && nv >= 2) {
            // common[i] = AND over the children of their dword i
            uint32_t common = 0xFFFFFFFFu;
            if (lane < pdw) for (uint32_t k = 0; k < nv; ++k) common &= wp[k * pdw + lane];
            else common = 0;
            if (__ballot(common != 0)) {
                if (lane < pdw) cc.programs[((size_t)cc.n_prog + entry) * pdw + lane] = common;
                // every child keeps what is not common; siblings keep at most ONE constraint per strand: then a child's program
                // shrinks to two descriptors (dword index << 5 | bit; index = dwords per strand when nothing is left)
                bool single = true;
                uint32_t rest[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < nv; ++k) {
                    rest[k] = lane < pdw ? wp[k * pdw + lane] & ~common : 0u;
                    const unsigned long long nz = __ballot(rest[k] != 0);
                    const unsigned long long lo_mask = sdw >= 64 ? ~0ull : ((1ull << sdw) - 1);
                    const unsigned long long s0 = nz & lo_mask, s1 = (nz >> sdw) & lo_mask;
                    // more than one dword of a strand left, or more than one bit in the one that is
                    if (__popcll(s0) > 1 || __popcll(s1) > 1) single = false;
                    if (__ballot(rest[k] != 0 && __popc(rest[k]) > 1)) single = false;
                }
                for (uint32_t k = 0; k < nv; ++k) {
                    if (single) {
                        const unsigned long long nz = __ballot(rest[k] != 0);
                        const unsigned long long lo_mask = sdw >= 64 ? ~0ull : ((1ull << sdw) - 1);
                        uint32_t desc[2];
                        for (uint32_t st = 0; st < 2; ++st) {
                            const unsigned long long sm = (nz >> (st * sdw)) & lo_mask;
                            desc[st] = sdw << 5;
                            if (sm) {
                                const int i = __ffsll((long long)sm) - 1;                       // dword within the strand
                                const uint32_t word = __shfl(rest[k], (int)(st * sdw) + i);
                                desc[st] = ((uint32_t)i << 5) | (uint32_t)(__ffs(word) - 1);
                            }
                        }
                        if (lane < pdw) wp[k * pdw + lane] = lane == 0 ? desc[0] : lane == 1 ? desc[1] : rest[k];
                    } else if (lane < pdw) {
                        wp[k * pdw + lane] = rest[k];
                    }
                }
                rg.w = single ? 1u : 0u;
                rg.z = cc.n_prog + entry;
            }
        }
        if (lane == 0) cc.range[entry] = rg;
        __threadfence_block();
        for (uint32_t i = lane; i < 4 * pdw; i += 64) cc.programs[(size_t)first * pdw + i] = wp[i];
    } else {
        if (lane < 4) nmdetail::compile_one(first + lane, cc.rec, cc.masks, cc.programs, cc.wide, cc.np, cc.fold_modpos);
        __threadfence();
        if (lane == 0 && cc.common) nmdetail::common_one(entry, cc.range, cc.programs, cc.programs + (size_t)cc.n_prog * cc.pdw, cc.pdw, cc.n_prog, 8u);
    }
}

// The window batch; spec != NULL: with the speculative children of its PSSM requests scored behind it (see above) — then the whole
// batch rides in the staging pair of that scoring batch: ONE copy in, one clear, window kernel, children kernel, scoring kernel, ONE
// copy out (a lock-step round of the search is a chain of commands, ~10 us each all told: their number is its cost)
static int win_batch_begin_impl(nm_ctx *c, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                                uint32_t ws, const nmdetail::WinSpec *spec, int flight = 0) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (flight < 0 || flight >= NM_FLIGHTS) return fail(NM_EINVAL, "bad flight");
    if (c->win_wait[flight].open) return fail(NM_ESTATE, "nm_win_batch_w_begin: the previous batch has not been collected (nm_win_batch_w_end)");
    if (n_req == 0) {
        nmdetail::wait_set(c, c->win_wait[flight], nm_ctx::Waiting{nullptr, 0, nullptr, true});
        return NM_OK;
    }
    if (!req_task || !req_kind || !req_sets) return fail(NM_EINVAL, "NULL argument");
    if (ws == 0 || ws > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "width stride %u outside 1..%d", ws, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    uint32_t max_nw = 1, max_w = 1;
    for (uint32_t r = 0; r < n_req; ++r) {
        if (req_task[r] >= c->win_tasks.size()) return fail(NM_EINVAL, "request %u: window task %u does not exist", r, req_task[r]);
        if (req_kind[r] > 1) return fail(NM_EINVAL, "request %u: kind must be 0 (pssm) or 1 (remove)", r);
        max_nw = std::max(max_nw, c->win_tasks[req_task[r]].nw);
        max_w = std::max(max_w, c->win_tasks[req_task[r]].width);
        if (c->win_tasks[req_task[r]].width > ws) return fail(NM_EINVAL, "request %u: window task of width %u, width stride %u", r, c->win_tasks[req_task[r]].width, ws);
    }
    if (c->win_tasks_dirty) {
        // the set-up work on the ctx stream is complete, no earlier batch is in flight: then the task table is replaced
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
        HIP_TRY(nmdetail::sync_flight_streams(c));
        if (c->d_win_tasks_cap < c->win_tasks.size()) {
            if (c->d_win_tasks) (void)nmdetail::dev_free(c->d_win_tasks);
            c->d_win_tasks = nullptr;
            c->d_win_tasks_cap = 0;
            HIP_TRY(nmdetail::dev_malloc(&c->d_win_tasks, c->win_tasks.size() * 2 * sizeof(WinTask)));
            c->d_win_tasks_cap = c->win_tasks.size() * 2;
        }
        HIP_TRY(hipMemcpy(c->d_win_tasks, c->win_tasks.data(), c->win_tasks.size() * sizeof(WinTask), hipMemcpyHostToDevice));
        c->win_tasks_dirty = false;
    }
    const uint32_t stride = 2 + 4 * ws;
    // input tables: task, kind, sets (+ spec: the search task and the (slot, bin) range entry of every request); outputs: the counts
    // (+ spec: {column, bases} per request)
    const size_t o_kind = (size_t)n_req * 4, o_sets = (o_kind + n_req + 15) & ~(size_t)15;
    const size_t o_stask = (o_sets + (size_t)n_req * ws + 15) & ~(size_t)15;
    const size_t o_entry = o_stask + (spec ? (size_t)n_req * 4 : 0);
    const size_t in_bytes = (o_entry + (spec ? (size_t)n_req * 4 : 0) + 15) & ~(size_t)15;
    const size_t out_bytes = (size_t)n_req * stride * 4, info_bytes = spec ? (size_t)n_req * 8 : 0;
    // slices of a task's words per request (blockIdx.z): one workgroup per 256 words when the batch is small, fewer and
    // fatter ones when a thousand requests already fill the device several times over (a workgroup's fixed cost — the match
    // mask set-up, the reduction of its 34 counters, its atomics — is most of what it does)
    const uint32_t n_col_groups = (max_w + WIN_COL_GROUP - 1) / WIN_COL_GROUP;
    uint32_t gz = std::max<uint32_t>(1, std::min<uint32_t>(64, (max_nw + 255) / 256));         // >= 1: tasks of an empty shard have no windows
    if (getenv("NM_WIN_THIN") == nullptr) {
        const uint64_t want = std::max<uint64_t>(1, (uint64_t)c->n_cus * 8 / std::max<uint64_t>(1, (uint64_t)n_req * n_col_groups));
        gz = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(gz, want));
    }
    const hipStream_t st = batch_stream(c, flight);
    auto fill_tables = [&](uint8_t *h) {
        memcpy(h, req_task, (size_t)n_req * 4);
        memcpy(h + o_kind, req_kind, n_req);
        memcpy(h + o_sets, req_sets, (size_t)n_req * ws);
        if (!spec) return;
        // range entry of a request's (slot, bin) group in the scoring batch: active index of the slot * n_bins + bin, the active
        // slots numbered in ascending order (score_impl does the same)
        bool used[NM_MAX_MOD_SLOTS] = {};
        for (uint32_t r = 0; r < n_req; ++r) used[spec->req_slot[r]] = true;
        int act[NM_MAX_MOD_SLOTS], na = 0;
        for (int s = 0; s < NM_MAX_MOD_SLOTS; ++s) act[s] = used[s] ? na++ : -1;
        uint32_t *h_entry = reinterpret_cast<uint32_t *>(h + o_entry);
        for (uint32_t r = 0; r < n_req; ++r) h_entry[r] = (uint32_t)act[spec->req_slot[r]] * c->n_bins + spec->req_bin[r];
        memcpy(h + o_stask, spec->req_search_task, (size_t)n_req * 4);
    };
    if (spec) {
        // four candidate slots per request, (bin, slot) known here, motifs written by spec_children_kernel from the window counts
        std::vector<uint32_t> cbin((size_t)n_req * 4);
        std::vector<uint8_t> cslot((size_t)n_req * 4);
        for (uint32_t r = 0; r < n_req; ++r)
            for (int k = 0; k < 4; ++k) { cbin[4 * r + k] = spec->req_bin[r]; cslot[4 * r + k] = spec->req_slot[r]; }
        const void *h_out = nullptr;
        nmdetail::SpecSource src;
        src.mask_stride = ws;
        src.extra_in = in_bytes;
        src.extra_out = out_bytes + info_bytes;
        src.h_extra_out = &h_out;
        src.fill_host = fill_tables;
        const SpecParams P{n_req, ws, spec->width, spec->pad, spec->min_kl, spec->freq_threshold};
        src.launch = [&](hipStream_t s, const uint8_t *d_in, uint8_t *d_out, const nmdetail::SpecCompile &cc) -> int {
            hipLaunchKernelGGL(win_request_kernel, dim3(n_req, n_col_groups, gz), dim3(256), 0, s, c->d_win_tasks, n_req, reinterpret_cast<const uint32_t *>(d_in),
                               d_in + o_kind, d_in + o_sets, c->d_win_planes, c->d_win_alive, reinterpret_cast<int *>(d_out), ws);
            hipLaunchKernelGGL(spec_children_kernel, dim3((n_req + 3) / 4), dim3(256), 0, s, P, d_in + o_kind, d_in + o_sets,
                               reinterpret_cast<const uint32_t *>(d_in + o_stask), reinterpret_cast<const uint32_t *>(d_in + o_entry),
                               reinterpret_cast<const int *>(d_out), c->d_spec_bg, reinterpret_cast<int *>(d_out + out_bytes), cc, ws);
            HIP_TRY(hipGetLastError());
            return NM_OK;
        };
        const int rc = nmdetail::score_batch_spec_begin(c, flight, n_req * 4, cbin.data(), cslot.data(), src, st);
        if (rc) return rc;
        nmdetail::wait_set(c, c->win_wait[flight], nm_ctx::Waiting{h_out, out_bytes, c->spec_wait[flight].stage, true});
        return NM_OK;
    }
    const size_t o_out = in_bytes;
    int rc = ensure_stage(c, o_out + out_bytes, 2 + flight);   // its own pair: a scoring batch enqueued behind this one never waits for it on the host
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
    fill_tables(hs);
    // tables in, counts cleared, counts out: kernels on the batch's stream, no copy engine in the chain (nmscan_internal.h: stage_in)
    const bool by_kernels = o_out + out_bytes <= nmdetail::STAGE_KERNEL_MAX && getenv("NM_STAGE_COPIES") == nullptr;
    if (by_kernels) {
        rc = nmdetail::stage_in(st, hs, ds, o_out, ds + o_out, out_bytes);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(ds + o_out, 0, out_bytes, st));
    }
    busy_begin_on(c, st);
    hipLaunchKernelGGL(win_request_kernel, dim3(n_req, n_col_groups, gz), dim3(256), 0, st, c->d_win_tasks, n_req,
                       reinterpret_cast<const uint32_t *>(ds), ds + o_kind, ds + o_sets, c->d_win_planes, c->d_win_alive,
                       reinterpret_cast<int *>(ds + o_out), ws);
    busy_end_on(c, st);
    HIP_TRY(hipGetLastError());
    if (by_kernels) {
        rc = nmdetail::stage_out(st, ds + o_out, hs + o_out, out_bytes);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(hs + o_out, ds + o_out, out_bytes, hipMemcpyDeviceToHost, st));
    }
    rc = release_stage(c, st);
    if (rc) return rc;
    nmdetail::wait_set(c, c->win_wait[flight], nm_ctx::Waiting{hs + o_out, out_bytes, c->cur_stage, true});
    return NM_OK;
}

int nm_win_batch_w_begin(nm_ctx *c, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                         uint32_t ws) {
    return win_batch_begin_impl(c, n_req, req_task, req_kind, req_sets, ws, nullptr);
}

}  // extern "C"

int nmdetail::win_batch_spec_begin(nm_ctx *c, int flight, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                                   uint32_t ws, const WinSpec *spec) {
    if (!spec) return win_batch_begin_impl(c, n_req, req_task, req_kind, req_sets, ws, nullptr, flight);
    if (n_req && (!spec->req_bin || !spec->req_slot || !spec->req_search_task)) return fail(NM_EINVAL, "NULL argument");
    if (!c || !c->d_spec_bg || spec->width != c->spec_width) return fail(NM_ESTATE, "speculative window batch without nmdetail::spec_setup");
    for (uint32_t r = 0; r < n_req; ++r)
        if (spec->req_search_task[r] >= c->spec_tasks || spec->req_slot[r] >= NM_MAX_MOD_SLOTS || spec->req_bin[r] >= c->n_bins)
            return fail(NM_EINVAL, "request %u: search task / slot / bin out of range", r);
    return win_batch_begin_impl(c, n_req, req_task, req_kind, req_sets, ws, spec, flight);
}

// the window replies (like nm_win_batch_w_end) + per request {column or -1, base rows A T G C as bits} and the counts of its children
// in the order of the set bits (n_mod, n_nomod; four slots per request)
int nmdetail::win_batch_spec_end(nm_ctx *c, int flight, uint32_t n_req, int32_t *out, int32_t *spec_info, int64_t *spec_counts) {
    if (!c || flight < 0 || flight >= NM_FLIGHTS) return fail(NM_EINVAL, "bad flight");
    if (!c->win_wait[flight].open) return fail(NM_ESTATE, "win_batch_spec_end without win_batch_spec_begin");
    const nm_ctx::Waiting w = c->win_wait[flight], s = c->spec_wait[flight];
    const bool had_spec = s.open;
    nmdetail::WaitCloser closer{c, &c->win_wait[flight], &c->spec_wait[flight]};      // (the pair stays held until the replies are copied out)
    if (w.bytes == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(w.stage->busy));                    // recorded behind the whole chain
    if (out) memcpy(out, w.h, w.bytes);
    if (spec_info && had_spec) memcpy(spec_info, static_cast<const uint8_t *>(w.h) + w.bytes, (size_t)n_req * 8);
    else if (spec_info) for (uint32_t r = 0; r < n_req; ++r) { spec_info[2 * r] = -1; spec_info[2 * r + 1] = 0; }
    if (spec_counts && had_spec && s.h && s.bytes == (size_t)n_req * 4 * 2 * sizeof(int64_t)) memcpy(spec_counts, s.h, s.bytes);
    else if (spec_counts) memset(spec_counts, 0, (size_t)n_req * 4 * 2 * sizeof(int64_t));
    return NM_OK;
}

int nmdetail::spec_setup(nm_ctx *c, uint32_t n_tasks, uint32_t width, const double *bg_pssm) {
    if (!c || (n_tasks && !bg_pssm)) return fail(NM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
    HIP_TRY(nmdetail::sync_flight_streams(c));
    if (c->d_spec_bg) (void)nmdetail::dev_free(c->d_spec_bg);
    c->d_spec_bg = nullptr;
    c->spec_tasks = c->spec_width = 0;
    if (n_tasks == 0) return NM_OK;
    HIP_TRY(nmdetail::dev_malloc(&c->d_spec_bg, (size_t)n_tasks * 4 * width * sizeof(double)));
    HIP_TRY(hipMemcpy(c->d_spec_bg, bg_pssm, (size_t)n_tasks * 4 * width * sizeof(double), hipMemcpyHostToDevice));
    c->spec_tasks = n_tasks;
    c->spec_width = width;
    return NM_OK;
}

extern "C" {

int nm_win_batch_w_end(nm_ctx *c, int32_t *out) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->win_wait[0].open) return fail(NM_ESTATE, "nm_win_batch_w_end without nm_win_batch_w_begin");
    const nm_ctx::Waiting w = c->win_wait[0];
    c->win_wait[0].open = false;
    if (w.bytes == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(w.stage->busy));
    if (out) memcpy(out, w.h, w.bytes);                         // NULL: the batch is dropped
    return NM_OK;
}

int nm_win_batch_w(nm_ctx *c, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                   uint32_t ws, int32_t *out) {
    if (n_req && !out) return fail(NM_EINVAL, "NULL argument");
    const int rc = nm_win_batch_w_begin(c, n_req, req_task, req_kind, req_sets, ws);
    return rc ? rc : nm_win_batch_w_end(c, out);
}

int nm_win_batch(nm_ctx *c, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                 int32_t *out) {
    return nm_win_batch_w(c, n_req, req_task, req_kind, req_sets, NM_WIN_MAX_WIDTH, out);
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

static int ensure_rank(nm_ctx *c, int b) {
    if (c->d_rank[b]) return NM_OK;
    HIP_TRY(nmdetail::dev_malloc(&c->d_rank[b], (size_t)c->n_chunks * RANK_PER_CHUNK * 4));
    HIP_TRY(nmdetail::dev_malloc(&c->d_base_total[b], (size_t)c->n_contigs * 8));
    if (c->n_contigs) hipLaunchKernelGGL(rank_build_kernel, dim3(c->n_contigs), dim3(64), 0, c->stream, seq_planes(c),
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
    HIP_TRY(nmdetail::dev_malloc(&d_out, (size_t)c->n_contigs * 8));
    if (c->n_contigs) hipLaunchKernelGGL(base_count_kernel, dim3((c->n_contigs + 255) / 256), dim3(256), 0, c->stream, seq_planes(c),
                       c->d_contig_chunk, c->d_contig_len, c->n_contigs, b, pad, c->d_base_total[b], d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)c->n_contigs * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)nmdetail::dev_free(d_out);
    return NM_OK;
}

// The samples of nm_bg_counts_runs arrive as RUNS (run r: run_count[r] consecutive samples on contig run_contig[r]); the
// per-sample contig column the counting kernel reads is written on the device.
struct BgRun { uint32_t begin_lo, begin_hi, count, contig; };

__global__ __launch_bounds__(256) void bg_expand_runs_kernel(const BgRun *__restrict__ runs, uint32_t *__restrict__ sample_contig) {
    const BgRun r = runs[blockIdx.x];
    uint32_t *dst = sample_contig + (((uint64_t)r.begin_hi << 32) | r.begin_lo);
    for (uint32_t i = threadIdx.x; i < r.count; i += blockDim.x) dst[i] = r.contig;
}

static int bg_counts_impl(nm_ctx *c, uint8_t base, uint32_t pad, uint64_t n_samples, const uint32_t *sample_contig,
                          const std::vector<BgRun> *runs, const uint32_t *sample_rank, uint32_t n_tasks,
                          const uint64_t *task_begin, int64_t *out) {
    const int b = base_index(base);
    if (b < 0) return fail(NM_EINVAL, "base must be one of A C G T");
    const uint32_t W = 2 * pad + 1;
    if (W > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", W, WIN_MAX_W);
    if (n_tasks == 0) return NM_OK;
    if (n_samples && !sample_rank) return fail(NM_EINVAL, "NULL sample column");
    if (task_begin[0] != 0 || task_begin[n_tasks] != n_samples) return fail(NM_EINVAL, "task_begin must run from 0 to n_samples");
    constexpr uint32_t SPB = 2048;                       // samples per workgroup
    std::vector<BgBlock> blocks;
    for (uint32_t t = 0; t < n_tasks; ++t) {
        if (task_begin[t + 1] < task_begin[t]) return fail(NM_EINVAL, "task_begin must be non-decreasing");
        for (uint64_t s0 = task_begin[t]; s0 < task_begin[t + 1]; s0 += SPB)
            blocks.push_back(BgBlock{t, (uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)std::min<uint64_t>(SPB, task_begin[t + 1] - s0)});
    }
    if (sample_contig)
        for (uint64_t i = 0; i < n_samples; ++i)
            if (sample_contig[i] >= c->n_contigs) return fail(NM_EINVAL, "sample %llu: contig %u >= %u", (unsigned long long)i, sample_contig[i], c->n_contigs);
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rank(c, b);
    if (rc) return rc;
    const size_t n_runs = runs ? runs->size() : 0;
    const size_t o_rank = (size_t)n_samples * 4, o_blk = (o_rank + (size_t)n_samples * 4 + 15) & ~(size_t)15;
    const size_t o_runs = (o_blk + blocks.size() * sizeof(BgBlock) + 15) & ~(size_t)15;
    const size_t o_out = (o_runs + n_runs * sizeof(BgRun) + 15) & ~(size_t)15;
    const uint32_t ws = (W + 7u) & ~7u;                       // columns per output row
    const size_t out_bytes = (size_t)n_tasks * 4 * ws * 8;
    rc = ensure_stage(c, o_out + out_bytes);
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
    if (n_samples) {
        if (sample_contig) memcpy(hs, sample_contig, (size_t)n_samples * 4);
        memcpy(hs + o_rank, sample_rank, (size_t)n_samples * 4);
    }
    if (!blocks.empty()) memcpy(hs + o_blk, blocks.data(), blocks.size() * sizeof(BgBlock));
    if (n_runs) memcpy(hs + o_runs, runs->data(), n_runs * sizeof(BgRun));
    const size_t up0 = sample_contig ? 0 : o_rank;       // runs: the contig column never crosses the bus
    HIP_TRY(hipMemcpyAsync(ds + up0, hs + up0, o_out - up0, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(ds + o_out, 0, out_bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
    if (n_runs) {
        hipLaunchKernelGGL(bg_expand_runs_kernel, dim3((unsigned)n_runs), dim3(256), 0, c->stream,
                           reinterpret_cast<const BgRun *>(ds + o_runs), reinterpret_cast<uint32_t *>(ds));
        HIP_TRY(hipGetLastError());
    }
    if (!blocks.empty()) {
        hipLaunchKernelGGL(bg_counts_kernel, dim3((unsigned)blocks.size()), dim3(256), 0, c->stream, seq_planes(c),
                           c->d_rank[b], c->d_contig_chunk, c->d_contig_len, reinterpret_cast<const BgBlock *>(ds + o_blk),
                           reinterpret_cast<const uint32_t *>(ds), reinterpret_cast<const uint32_t *>(ds + o_rank), b, pad,
                           reinterpret_cast<unsigned long long *>(ds + o_out), ws, c->d_err);
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
                out[((size_t)t * 4 + r) * W + col] = (int64_t)ho[((size_t)t * 4 + r) * ws + col];
    return NM_OK;
}

int nm_bg_counts(nm_ctx *c, uint8_t base, uint32_t pad, uint64_t n_samples, const uint32_t *sample_contig,
                 const uint32_t *sample_rank, uint32_t n_tasks, const uint64_t *task_begin, int64_t *out) {
    if (!c || !task_begin || !out) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (n_samples && !sample_contig) return fail(NM_EINVAL, "NULL sample column");
    return bg_counts_impl(c, base, pad, n_samples, n_samples ? sample_contig : nullptr, nullptr, sample_rank, n_tasks, task_begin, out);
}

int nm_bg_counts_runs(nm_ctx *c, uint8_t base, uint32_t pad, uint32_t n_runs, const uint32_t *run_contig, const uint32_t *run_count,
                      const uint32_t *sample_rank, uint32_t n_tasks, const uint32_t *task_run_begin, int64_t *out) {
    if (!c || !task_run_begin || !out || (n_runs && (!run_contig || !run_count))) return fail(NM_EINVAL, "NULL argument");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (task_run_begin[0] != 0 || task_run_begin[n_tasks] != n_runs) return fail(NM_EINVAL, "task_run_begin must run from 0 to n_runs");
    std::vector<BgRun> runs(n_runs);
    std::vector<uint64_t> task_begin(n_tasks + 1, 0);
    uint64_t at = 0;
    uint32_t t = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {
        if (run_contig[r] >= c->n_contigs) return fail(NM_EINVAL, "run %u: contig %u >= %u", r, run_contig[r], c->n_contigs);
        while (t < n_tasks && task_run_begin[t] <= r) {
            if (t && task_run_begin[t] < task_run_begin[t - 1]) return fail(NM_EINVAL, "task_run_begin must be non-decreasing");
            task_begin[t++] = at;
        }
        runs[r] = BgRun{(uint32_t)at, (uint32_t)(at >> 32), run_count[r], run_contig[r]};
        at += run_count[r];
    }
    while (t <= n_tasks) task_begin[t++] = at;
    return bg_counts_impl(c, base, pad, at, nullptr, &runs, sample_rank, n_tasks, task_begin.data(), out);
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
        if (n_rows) hipLaunchKernelGGL(win_gather_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, c->stream, t, seq_planes(c),
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
            HIP_TRY(nmdetail::dev_malloc(&ms.rank[k], (size_t)c->n_chunks * RANK_PER_CHUNK * 4));
            HIP_TRY(nmdetail::dev_malloc(&ms.rank_total[k], (size_t)c->n_contigs * 8));
            if (c->n_contigs) hipLaunchKernelGGL(rank_build_kernel, dim3(c->n_contigs), dim3(64), 0, c->stream, seq_planes(c),
                               static_cast<const uint32_t *>(ms.planes[k == 0 ? 2 : 4]), c->d_contig_chunk, c->d_contig_len, -1,
                               ms.rank[k], ms.rank_total[k]);
            HIP_TRY(hipGetLastError());
        }
        ms.meth_pad = 0xFFFFFFFFu;
    }
    if (ms.meth_pad == pad && ms.meth_counts.size() == (size_t)c->n_contigs * 4) return NM_OK;
    uint64_t *d_out = nullptr;
    HIP_TRY(nmdetail::dev_malloc(&d_out, (size_t)c->n_contigs * 4 * 8));
    if (c->n_contigs) hipLaunchKernelGGL(meth_count_kernel, dim3((c->n_contigs + 255) / 256), dim3(256), 0, c->stream, ms.planes[2], ms.planes[4],
                       c->d_contig_chunk, c->d_contig_len, c->n_contigs, pad, ms.rank_total[0], ms.rank_total[1], d_out);
    HIP_TRY(hipGetLastError());
    ms.meth_counts.assign((size_t)c->n_contigs * 4, 0);
    HIP_TRY(hipMemcpyAsync(ms.meth_counts.data(), d_out, (size_t)c->n_contigs * 4 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)nmdetail::dev_free(d_out);
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
        if (t.n) hipLaunchKernelGGL(win_gather_contigs_kernel, dim3((t.n + 255) / 256), dim3(256), 0, c->stream, t, seq_planes(c),
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


// ------------------------------------------------------------------------------------------------------
// random.sample(range(n), k) of every generator stream ON THE DEVICE.  When all streams start from the same state (the
// reference seeds every (bin, mod type) task afresh, find_motifs_bin.py:152-171) they all read ONE sequence of MT19937
// outputs, which the host generates once; what differs per stream is how the sequence is CONSUMED, and that is a scan:
// a draw r = raw >> (32 - bits(n)) is rejected when r >= n (Random._randbelow_with_getrandbits), skipped when r was
// selected before (the `while j in selected` loop of random.sample's set branch, random.py:449-466), selected otherwise,
// until k are selected.  One wave per stream walks its calls in order, 64 raw draws at a time: range test per lane,
// membership of earlier tiles in a bitmap (one word array per stream, bits cleared again at the end of the call), first
// occurrence inside the tile by a leader loop over the distinct values, cut at the k-th selection.  The pool branch
// (n <= setsize: tiny contigs) is replayed by lane 0.
// ------------------------------------------------------------------------------------------------------
struct DrawCall { uint32_t n, k, out_lo, out_hi; };      // out: first rank in `ranks` (kept samples) or, bit 63 set, in the discard area

// A stream's scratch and outputs are touched by ONE wave only: what its lanes hand each other through global memory has to
// be ordered, not made visible to the other dies.  Atomics and write-through stores are performed at the L2 of the wave's
// own XCD; "all my earlier memory operations have been performed" (s_waitcnt) plus loads that bypass the L1 is all it takes
// — an agent-scope fence writes back and invalidates caches and cost 10+ us per 64 draws.
__device__ __forceinline__ uint32_t load_coherent(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_mem_sync() { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(64) void bg_draw_kernel(const uint32_t *__restrict__ raw, uint32_t raw_len, const DrawCall *__restrict__ calls,
                                                     const uint32_t *__restrict__ group_call_off, const uint64_t *__restrict__ group_scratch_off,
                                                     uint32_t *scratch, uint32_t *ranks, uint32_t *discard, uint32_t *consumed /*[n_groups]*/,
                                                     uint32_t lds_words) {
    // The membership bitmap (or the pool) of the group's calls lives in LDS when it fits the launch's dynamic allocation — the words of
    // a 300 kbp contig's valid starts are 10 KB —: every tile of 64 draws is a dependent load, a round of atomics and a fence on it, and
    // through a flat pointer to LDS that is a hundred clocks instead of a trip to L2.  A group that needs more keeps the global scratch.
    extern __shared__ uint32_t lds_bm[];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint64_t my_words = group_scratch_off[g + 1] - group_scratch_off[g];
    const bool in_lds = my_words <= (uint64_t)lds_words;
    uint32_t *bm = in_lds ? static_cast<uint32_t *>(lds_bm) : scratch + group_scratch_off[g];
    if (in_lds) {
        for (uint32_t i = lane; i < (uint32_t)my_words; i += 64) lds_bm[i] = 0;
        wave_mem_sync();
    }
    uint32_t p = 0;                                       // raw draws consumed so far (wave-uniform)
    bool overflow = false;
    for (uint32_t ci = group_call_off[g]; ci < group_call_off[g + 1] && !overflow; ++ci) {
        const DrawCall cl = calls[ci];
        const uint64_t out_at = ((uint64_t)(cl.out_hi & 0x7FFFFFFFu) << 32) | cl.out_lo;
        uint32_t *out = ((cl.out_hi >> 31) ? discard : ranks) + out_at;
        const uint32_t n = cl.n, k = cl.k;
        if (k == 0) continue;
        // random.py:449-453: setsize = 21, + 4 ** ceil(log(3 k, 4)) for k > 5 — the smallest power of four >= 3 k
        uint64_t setsize = 21;
        if (k > 5) {
            uint64_t p4 = 1;
            while (p4 < 3ull * k) p4 *= 4;
            setsize += p4;
        }
        if ((uint64_t)n <= setsize) {
            // pool branch: j = randbelow(n - i); result[i] = pool[j]; pool[j] = pool[n - i - 1]
            for (uint32_t i = lane; i < n; i += 64) bm[i] = i;
            wave_mem_sync();
            if (lane == 0) {
                for (uint32_t i = 0; i < k && !overflow; ++i) {
                    const uint32_t m = n - i, sh = (uint32_t)__clz((int)m);
                    uint32_t r = 0xFFFFFFFFu;
                    while (r >= m) {
                        if (p >= raw_len) { overflow = true; break; }
                        r = raw[p++] >> sh;
                    }
                    if (overflow) break;
                    out[i] = load_coherent(bm + r);
                    __hip_atomic_store(bm + r, load_coherent(bm + m - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
            overflow = __builtin_amdgcn_readfirstlane((int)overflow) != 0;
            wave_mem_sync();
            for (uint32_t i = lane; i < n; i += 64) bm[i] = 0;       // the scratch goes back to all zero
            wave_mem_sync();
            continue;
        }
        const uint32_t sh = (uint32_t)__clz((int)n);         // 32 - n.bit_length()
        uint32_t selected = 0;
        while (selected < k) {
            if (p >= raw_len) { overflow = true; break; }
            const uint32_t idx = p + lane;
            const bool in_range = idx < raw_len;
            const uint32_t r = in_range ? raw[idx] >> sh : 0xFFFFFFFFu;
            const bool valid = in_range && r < n;
            const uint32_t word = valid ? r >> 5 : 0u, bit = 1u << (r & 31u);
            const bool cand = valid && !(load_coherent(bm + word) & bit);
            // Every candidate sets its bit at once and looks at what was there: a candidate that finds the bit SET shares its value with
            // another lane of this tile (nobody had it before the tile) — one tile in twenty.  Only then the lanes of that value are
            // sorted out, the lowest keeps it (random.sample takes the first occurrence).  (The loop used to take one turn per DISTINCT
            // value of every tile, ~50 turns: most of the kernel's 3 ms for the 2e7 draws of the 1 Gbp plan.)
            bool keep = cand;
            uint32_t before = 0;
            if (cand) before = atomicOr(bm + word, bit);
            unsigned long long lost = __ballot(cand && (before & bit));
            while (lost) {                                   // wave-uniform: one turn per value that occurs twice in the tile
                const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)lost) - 1);
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);
                const unsigned long long same = __ballot(cand && r == v);
                const int first = __ffsll((long long)same) - 1;
                if (cand && r == v && (int)lane != first) keep = false;
                lost &= ~same;
            }
            const bool first_of_value = keep;                // (its bit is set; a later lane of the same value set it too, at most)
            unsigned long long sel = __ballot(keep);
            uint32_t advance = 64;
            if (selected + (uint32_t)__popcll(sel) >= k) {   // the call ends inside this tile: at its (k - selected)-th selection
                unsigned long long m = sel;
                for (uint32_t skip = k - selected - 1; skip; --skip) m &= m - 1;
                const int last = __ffsll((long long)m) - 1;
                if (last < 63) sel &= (2ull << last) - 1ull;
                keep = keep && (int)lane <= last;
                advance = (uint32_t)last + 1u;
            }
            if (keep) out[selected + (uint32_t)__popcll(sel & ((1ull << lane) - 1ull))] = r;
            else if (first_of_value) atomicAnd(bm + word, ~bit);      // drawn after the call's last selection: not taken, the bit goes back
            selected += (uint32_t)__popcll(sel);
            p += advance;
            wave_mem_sync();                                 // this tile's bits before the next tile's membership loads
        }
        // clear the bits of this call: the bitmap is all zero between calls
        wave_mem_sync();
        for (uint32_t i = lane; i < selected; i += 64) {
            const uint32_t v = load_coherent(out + i);
            atomicAnd(bm + (v >> 5), ~(1u << (v & 31u)));
        }
        wave_mem_sync();
    }
    if (lane == 0) consumed[g] = overflow ? 0xFFFFFFFFu : p;
}

// ------------------------------------------------------------------------------------------------------
// nm_plan_windows: window extraction of ALL (bin, mod type) tasks in one call (find_motifs_bin.py:625-686 per task): the
// reference walks the task's contigs in order; per contig it draws the background sample (random.sample of the valid
// starts, seq.py:202-225) and then gathers the methylation windows, and gives the task up (None) at the first contig
// without a window — the contigs up to and including that one have consumed random numbers.  Here: the plan on the host
// (counts come from the rank tables), ONE gather launch for the windows of every task, the draws of all generator
// streams on host threads while that kernel runs, written straight into the pinned staging buffer, one counting launch
// per canonical base.
// ------------------------------------------------------------------------------------------------------
int nm_plan_windows(nm_ctx *c, uint32_t n_tasks, const uint32_t *task_slot, const uint8_t *task_base, const uint32_t *task_group,
                    const uint32_t *task_contig_begin, const uint32_t *contig_id, uint32_t pad, double freq, uint32_t n_groups,
                    const uint32_t *group_init_state, int shared_init, uint8_t *task_status, uint32_t *task_window,
                    uint64_t *task_n_windows, uint64_t *task_n_bg, int64_t *bg_counts, uint32_t final_state[625]) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (n_tasks == 0) return NM_OK;
    if (!task_slot || !task_base || !task_group || !task_contig_begin || !task_status || !task_window || !task_n_windows || !task_n_bg ||
        !bg_counts || !group_init_state || !final_state || (task_contig_begin[n_tasks] && !contig_id))
        return fail(NM_EINVAL, "NULL argument");
    const uint32_t W = 2 * pad + 1;
    if (W > (uint32_t)WIN_MAX_W) return fail(NM_ERANGE, "window width %u outside 1..%d", W, WIN_MAX_W);
    HIP_TRY(hipSetDevice(c->device));
    // NM_PLAN_TIMING: where the wall time of this call goes (stderr, one line)
    const bool timing = getenv("NM_PLAN_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now(), t_counts = 0, t_plan = 0, t_stage = 0, t_draws = 0, t_tail = 0;
    auto lap = [&](double &acc) { const double t = now(); acc += t - t_mark; t_mark = t; };
    // ---- counts the plan needs: valid sample starts per contig and base, confident rows per contig and slot
    std::vector<uint64_t> n_valid[4];
    for (uint32_t t = 0; t < n_tasks; ++t) {
        const int b = base_index(task_base[t]);
        if (b < 0) return fail(NM_EINVAL, "task %u: base must be one of A C G T", t);
        if (task_slot[t] >= NM_MAX_MOD_SLOTS || !c->slots[task_slot[t]].present) return fail(NM_ESTATE, "task %u: mod slot %u holds no pileup", t, task_slot[t]);
        if (task_group[t] >= n_groups || (t && task_group[t] < task_group[t - 1])) return fail(NM_EINVAL, "task_group must be non-decreasing and < n_groups");
        if (n_valid[b].empty()) {
            n_valid[b].assign(c->n_contigs, 0);
            const int rc = nm_contig_base_counts(c, task_base[t], pad, n_valid[b].data());
            if (rc) return rc;
        }
        const int rc = ensure_slot_counts(c, task_slot[t], pad);
        if (rc) return rc;
    }
    lap(t_counts);
    // ---- the plan
    struct Call { uint64_t n, k, out; };                   // random.sample(range(n), k); out: first rank in the staging buffer or ~0 (discarded)
    std::vector<std::vector<Call>> group_calls(n_groups);
    std::vector<WinSegment> segs;
    std::vector<WinBlock> blocks;
    std::vector<WinTask> new_tasks;
    std::vector<BgRun> runs;
    std::vector<BgBlock> bg_blocks[4];
    uint64_t n_samples = 0, planes_used = c->win_planes_used, alive_used = c->win_alive_used;
    constexpr uint32_t SPB = 2048;
    for (uint32_t t = 0; t < n_tasks; ++t) {
        const uint32_t a = task_contig_begin[t], e = task_contig_begin[t + 1];
        if (e < a) return fail(NM_EINVAL, "task_contig_begin must be non-decreasing");
        const int b = base_index(task_base[t]);
        const ModSlot &ms = c->slots[task_slot[t]];
        task_status[t] = 1;
        task_window[t] = 0xFFFFFFFFu;
        task_n_windows[t] = task_n_bg[t] = 0;
        uint32_t stop = e;
        for (uint32_t k = a; k < e; ++k) {
            const uint32_t ci = contig_id[k];
            if (ci >= c->n_contigs) return fail(NM_EINVAL, "task %u: contig %u >= %u", t, ci, c->n_contigs);
            if (ms.meth_counts[4 * (size_t)ci] + ms.meth_counts[4 * (size_t)ci + 1] == 0) { stop = k; break; }
        }
        const uint32_t drawn_end = std::min(stop + 1, e);
        uint64_t total = 0, n_bg = 0;
        const size_t seg0 = segs.size(), run0 = runs.size();
        const uint64_t samples0 = n_samples;
        for (uint32_t k = a; k < drawn_end; ++k) {
            const uint32_t ci = contig_id[k];
            const uint64_t len = c->contig_len[ci];
            const uint64_t want = std::max<uint64_t>((uint64_t)std::ceil((double)len * freq), 50);       // find_motifs_bin.py:633
            if (len < W || want > len - W + 1) return fail(NM_EINVAL, "Too many samples requested for unique subsequences");
            if (n_valid[b][ci] < want)
                return fail(NM_EINVAL, "Not enough subsequences with '%c' in the middle (found %llu, need %llu)", task_base[t],
                            (unsigned long long)n_valid[b][ci], (unsigned long long)want);
            const bool keep = stop == e;
            group_calls[task_group[t]].push_back(Call{n_valid[b][ci], want, keep ? n_samples : ~0ull});
            if (!keep) continue;
            runs.push_back(BgRun{(uint32_t)n_samples, (uint32_t)(n_samples >> 32), (uint32_t)want, ci});
            n_samples += want;
            n_bg += want;
            for (uint32_t strand = 0; strand < 2; ++strand) {       // plus rows, then minus rows (find_motifs_bin.py:640-659)
                const uint64_t n = ms.meth_counts[4 * (size_t)ci + strand];
                if (!n) continue;
                segs.push_back(WinSegment{(uint32_t)total, ci, strand, (uint32_t)ms.meth_counts[4 * (size_t)ci + 2 + strand]});
                total += n;
            }
        }
        if (stop != e || total == 0 || n_bg == 0) {                  // None: no methylation windows (find_motifs_bin.py:662-664)
            segs.resize(seg0);
            runs.resize(run0);
            n_samples = samples0;
            for (Call &cl : group_calls[task_group[t]])
                if (cl.out != ~0ull && cl.out >= samples0) cl.out = ~0ull;
            continue;
        }
        if (total >= 0xFFFFFFFFull) return fail(NM_ERANGE, "more than 4G windows in one task");
        WinTask wt{};
        wt.n = (uint32_t)total;
        wt.nw = (wt.n + 31) / 32;
        wt.width = W;
        wt.plane_off = planes_used;
        wt.alive_off = alive_used;
        planes_used += (uint64_t)W * 5 * wt.nw;
        alive_used += wt.nw;
        const uint32_t tid = (uint32_t)(c->win_tasks.size() + new_tasks.size());
        new_tasks.push_back(wt);
        for (uint32_t first = 0; first < wt.n; first += 256)
            blocks.push_back(WinBlock{tid, first, (uint32_t)seg0, (uint32_t)(segs.size() - seg0), task_slot[t]});
        for (uint64_t s0 = samples0; s0 < n_samples; s0 += SPB)
            bg_blocks[b].push_back(BgBlock{t, (uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)std::min<uint64_t>(SPB, n_samples - s0)});
        task_status[t] = 0;
        task_window[t] = tid;
        task_n_windows[t] = total;
        task_n_bg[t] = n_bg;
    }
    lap(t_plan);
    // device draws: flat call table, a bound on the raw draws any stream consumes, scratch words per stream
    bool device_draws = shared_init && n_groups >= 32 && n_samples && getenv("NM_HOST_DRAWS") == nullptr;
    std::vector<DrawCall> flat;
    std::vector<uint32_t> call_off(n_groups + 1, 0);
    std::vector<uint64_t> scratch_off(n_groups + 1, 0);
    uint64_t raw_len = 0, n_discard = 0;
    if (device_draws) {
        for (uint32_t g = 0; g < n_groups && device_draws; ++g) {
            double expect = 0;
            uint64_t words = 1;
            for (const Call &cl : group_calls[g]) {
                uint64_t setsize = 21;
                if (cl.k > 5) {
                    uint64_t p4 = 1;
                    while (p4 < 3 * cl.k) p4 *= 4;
                    setsize += p4;
                }
                const bool pool = cl.n <= setsize;
                if (cl.n >= 0xFFFFFFFFull || (pool && cl.k > 4096)) { device_draws = false; break; }   // (a long pool replay is serial: host)
                const int bits = 64 - __builtin_clzll(cl.n | 1);
                // a draw is in range with probability n / 2^bits (> 1/2) and new with probability >= 1 - k / n
                expect += (double)cl.k * ((double)(1ull << bits) / (double)cl.n) / std::max(0.05, 1.0 - (double)cl.k / (double)cl.n) + 64.0;
                words = std::max<uint64_t>(words, pool ? cl.n : (cl.n + 31) / 32);
                const uint64_t out = cl.out == ~0ull ? ((1ull << 63) | n_discard) : cl.out;
                if (cl.out == ~0ull) n_discard += cl.k;
                flat.push_back(DrawCall{(uint32_t)cl.n, (uint32_t)cl.k, (uint32_t)out, (uint32_t)(out >> 32)});
            }
            call_off[g + 1] = (uint32_t)flat.size();
            scratch_off[g + 1] = scratch_off[g] + words;
            raw_len = std::max<uint64_t>(raw_len, (uint64_t)(expect * 1.3) + 4096);
        }
        if (raw_len > (8ull << 20) || scratch_off[n_groups] > (1ull << 30)) device_draws = false;   // one very long stream: nothing to run in parallel
    }
    // ---- window pools and the task table
    int rc = win_grow(c, &c->d_win_planes, &c->win_planes_cap, c->win_planes_used, planes_used - c->win_planes_used);
    if (rc) return rc;
    rc = win_grow(c, &c->d_win_alive, &c->win_alive_cap, c->win_alive_used, alive_used - c->win_alive_used);
    if (rc) return rc;
    c->win_tasks.insert(c->win_tasks.end(), new_tasks.begin(), new_tasks.end());
    c->win_planes_used = planes_used;
    c->win_alive_used = alive_used;
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));           // window batches of an earlier search (batch_stream)
    HIP_TRY(nmdetail::sync_flight_streams(c));
    if (c->d_win_tasks_cap < c->win_tasks.size()) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_win_tasks) (void)nmdetail::dev_free(c->d_win_tasks);
        c->d_win_tasks = nullptr;
        c->d_win_tasks_cap = 0;
        HIP_TRY(nmdetail::dev_malloc(&c->d_win_tasks, c->win_tasks.size() * 2 * sizeof(WinTask)));
        c->d_win_tasks_cap = c->win_tasks.size() * 2;
    }
    HIP_TRY(hipMemcpyAsync(c->d_win_tasks, c->win_tasks.data(), c->win_tasks.size() * sizeof(WinTask), hipMemcpyHostToDevice, c->stream));
    c->win_tasks_dirty = false;
    // ---- staging: [ranks | sample contig column (device only)] | segments | window blocks | runs | bg blocks (4 lists) | counts.
    // With device draws the two sample columns never exist on the host: they get a device buffer of their own and the pinned
    // staging pair stays small (pinning 160 MB for a 1 Gbp plan was 10 ms of this call).
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    struct DeviceBuffer {
        uint8_t *p = nullptr;
        ~DeviceBuffer() {                                    // (error paths leave kernels in flight: wait before the memory goes back)
            if (p) { (void)hipDeviceSynchronize(); (void)nmdetail::dev_free(p); }
        }
    } d_samples;
    const size_t col_bytes = up16((size_t)n_samples * 4);
    if (device_draws) HIP_TRY(nmdetail::dev_malloc(&d_samples.p, 2 * col_bytes + 16));
    const size_t o_rank = 0, o_contig = device_draws ? 0 : col_bytes, o_seg = device_draws ? 0 : up16(o_contig + (size_t)n_samples * 4);
    const size_t o_blk = up16(o_seg + segs.size() * sizeof(WinSegment)), o_runs = up16(o_blk + blocks.size() * sizeof(WinBlock));
    size_t o_bg[4], at = up16(o_runs + runs.size() * sizeof(BgRun));
    for (int b = 0; b < 4; ++b) { o_bg[b] = at; at = up16(at + bg_blocks[b].size() * sizeof(BgBlock)); }
    const uint32_t ws = (W + 7u) & ~7u;                       // columns per row of the background counts
    const size_t o_out = at, out_bytes = (size_t)n_tasks * 4 * ws * 8;
    rc = ensure_stage(c, o_out + out_bytes);
    if (rc) return rc;
    uint8_t *hs = static_cast<uint8_t *>(c->h_stage), *ds = static_cast<uint8_t *>(c->d_stage);
    uint32_t *const d_rank_col = reinterpret_cast<uint32_t *>(device_draws ? d_samples.p : ds + o_rank);
    uint32_t *const d_contig_col = reinterpret_cast<uint32_t *>(device_draws ? d_samples.p + col_bytes : ds + o_contig);
    if (!segs.empty()) memcpy(hs + o_seg, segs.data(), segs.size() * sizeof(WinSegment));
    if (!blocks.empty()) memcpy(hs + o_blk, blocks.data(), blocks.size() * sizeof(WinBlock));
    if (!runs.empty()) memcpy(hs + o_runs, runs.data(), runs.size() * sizeof(BgRun));
    for (int b = 0; b < 4; ++b)
        if (!bg_blocks[b].empty()) memcpy(hs + o_bg[b], bg_blocks[b].data(), bg_blocks[b].size() * sizeof(BgBlock));
    HIP_TRY(hipMemcpyAsync(ds + o_seg, hs + o_seg, o_out - o_seg, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(ds + o_out, 0, out_bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
    if (!blocks.empty()) {
        AllSlotRows rows{};
        for (int sl = 0; sl < NM_MAX_MOD_SLOTS; ++sl) {
            const ModSlot &ms = c->slots[sl];
            rows.s[sl] = SlotRows{ms.planes[2], ms.planes[4], ms.rank[0], ms.rank[1]};
        }
        nmdetail::busy_begin(c);
        hipLaunchKernelGGL(win_gather_all_kernel, dim3((unsigned)blocks.size()), dim3(256), 0, c->stream, c->d_win_tasks,
                           reinterpret_cast<const WinBlock *>(ds + o_blk), seq_planes(c), rows, c->d_contig_chunk, c->d_contig_len,
                           reinterpret_cast<const WinSegment *>(ds + o_seg), pad, c->d_win_planes, c->d_win_alive, c->d_err);
        nmdetail::busy_end(c);
        HIP_TRY(hipGetLastError());
    }
    lap(t_stage);
    // ---- the draws.  Streams that all start from one state (the plain-pileup task order) are consumed on the device
    // (bg_draw_kernel); otherwise, or when a stream is too long for that to pay, on host threads while the gather kernel runs.
    auto host_draws = [&]() -> int {
        std::vector<uint32_t> unpinned;                      // (only when the device draws were tried first: no pinned column then)
        if (device_draws) unpinned.resize(n_samples + 1);
        uint32_t *ranks = device_draws ? unpinned.data() : reinterpret_cast<uint32_t *>(hs + o_rank);
        std::vector<int> rcs(n_groups, NM_OK);
        std::vector<std::string> errs(n_groups);
        const unsigned threads = std::max(1u, std::min<unsigned>({16u, std::thread::hardware_concurrency(), n_groups}));
        std::vector<std::vector<uint32_t>> last(threads);
        auto work = [&](unsigned th) {
            std::vector<uint32_t> st(625), scratch;
            for (uint32_t g = th; g < n_groups; g += threads) {
                memcpy(st.data(), group_init_state + (shared_init ? 0 : (size_t)g * 625), 625 * 4);
                for (const Call &cl : group_calls[g]) {
                    uint32_t *dst = cl.out == ~0ull ? (scratch.resize(cl.k), scratch.data()) : ranks + cl.out;
                    rcs[g] = nm_py_random_sample(st.data(), cl.n, cl.k, dst);
                    if (rcs[g] != NM_OK) { errs[g] = nm_last_error(); break; }
                }
                if (g == n_groups - 1) last[th] = st;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned th = 1; th < threads; ++th) pool.emplace_back(work, th);
        work(0);
        for (auto &th : pool) th.join();
        for (uint32_t g = 0; g < n_groups; ++g)
            if (rcs[g] != NM_OK) {
                (void)hipStreamSynchronize(c->stream);
                return fail(rcs[g], "%s", errs[g].c_str());
            }
        for (unsigned th = 0; th < threads; ++th)
            if (!last[th].empty()) memcpy(final_state, last[th].data(), 625 * 4);
        if (n_samples) HIP_TRY(hipMemcpyAsync(d_rank_col, ranks, (size_t)n_samples * 4, hipMemcpyHostToDevice, c->stream));
        if (device_draws) HIP_TRY(hipStreamSynchronize(c->stream));
        return NM_OK;
    };
    bool drew_on_device = false;
    if (device_draws) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t q_calls = 0, q_coff = up(q_calls + flat.size() * sizeof(DrawCall)), q_soff = up(q_coff + call_off.size() * 4);
        const size_t q_raw = up(q_soff + scratch_off.size() * 8), q_cons = up(q_raw + raw_len * 4), q_scr = up(q_cons + (size_t)n_groups * 4);
        const size_t q_disc = up(q_scr + scratch_off[n_groups] * 4), q_end = up(q_disc + (n_discard + 1) * 4);
        uint8_t *d_draw = nullptr;
        HIP_TRY(nmdetail::dev_malloc(&d_draw, q_end));
        std::vector<uint8_t> hbuf(q_cons, 0);
        memcpy(hbuf.data() + q_calls, flat.data(), flat.size() * sizeof(DrawCall));
        memcpy(hbuf.data() + q_coff, call_off.data(), call_off.size() * 4);
        memcpy(hbuf.data() + q_soff, scratch_off.data(), scratch_off.size() * 8);
        uint32_t st[625];
        memcpy(st, group_init_state, sizeof st);
        nm_mt_outputs(st, raw_len, reinterpret_cast<uint32_t *>(hbuf.data() + q_raw));
        std::vector<uint32_t> consumed(n_groups, 0);
        hipError_t e = hipMemcpyAsync(d_draw, hbuf.data(), q_cons, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_draw + q_scr, 0, scratch_off[n_groups] * 4, c->stream);
        if (e == hipSuccess) {
            nmdetail::busy_begin(c);
            uint64_t max_words = 0;
            for (uint32_t g = 0; g < n_groups; ++g) max_words = std::max<uint64_t>(max_words, scratch_off[g + 1] - scratch_off[g]);
            static const bool no_lds = getenv("NM_DRAW_GLOBAL_BITMAP") != nullptr;       // (A/B: every bitmap in global memory)
            const uint32_t lds_words = no_lds ? 0u : (uint32_t)std::min<uint64_t>(max_words, 8192);       // at most 32 KB per workgroup: five per CU
            hipLaunchKernelGGL(bg_draw_kernel, dim3(n_groups), dim3(64), (size_t)lds_words * 4, c->stream, reinterpret_cast<const uint32_t *>(d_draw + q_raw),
                               (uint32_t)raw_len, reinterpret_cast<const DrawCall *>(d_draw + q_calls),
                               reinterpret_cast<const uint32_t *>(d_draw + q_coff), reinterpret_cast<const uint64_t *>(d_draw + q_soff),
                               reinterpret_cast<uint32_t *>(d_draw + q_scr), d_rank_col,
                               reinterpret_cast<uint32_t *>(d_draw + q_disc), reinterpret_cast<uint32_t *>(d_draw + q_cons), lds_words);
            nmdetail::busy_end(c);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(consumed.data(), d_draw + q_cons, (size_t)n_groups * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)nmdetail::dev_free(d_draw);
        if (e != hipSuccess) return fail(NM_EHIP, "device draws: %s", hipGetErrorString(e));
        drew_on_device = true;
        for (uint32_t g = 0; g < n_groups; ++g)
            if (consumed[g] == 0xFFFFFFFFu) drew_on_device = false;       // a stream outran the generated sequence (1.3 x its expectation): host replay
        if (drew_on_device) {
            memcpy(st, group_init_state, sizeof st);
            nm_mt_outputs(st, consumed[n_groups - 1], nullptr);
            memcpy(final_state, st, sizeof st);
        }
    }
    if (!drew_on_device) {
        rc = host_draws();
        if (rc) return rc;
    }
    lap(t_draws);
    if (n_samples) {
        nmdetail::busy_begin(c);
        hipLaunchKernelGGL(bg_expand_runs_kernel, dim3((unsigned)runs.size()), dim3(256), 0, c->stream,
                           reinterpret_cast<const BgRun *>(ds + o_runs), d_contig_col);
        HIP_TRY(hipGetLastError());
        for (int b = 0; b < 4; ++b) {
            if (bg_blocks[b].empty()) continue;
            hipLaunchKernelGGL(bg_counts_kernel, dim3((unsigned)bg_blocks[b].size()), dim3(256), 0, c->stream, seq_planes(c), c->d_rank[b],
                               c->d_contig_chunk, c->d_contig_len, reinterpret_cast<const BgBlock *>(ds + o_bg[b]), d_contig_col, d_rank_col, b, pad,
                               reinterpret_cast<unsigned long long *>(ds + o_out), ws, c->d_err);
            HIP_TRY(hipGetLastError());
        }
        nmdetail::busy_end(c);
    }
    unsigned int err = 0;
    HIP_TRY(hipMemcpyAsync(hs + o_out, ds + o_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    rc = release_stage(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err & 8u) return fail(NM_ESTATE, "a window's row rank is beyond the rows of its contig (the slot's planes changed?)");
    if (err) return fail(NM_EINVAL, "a sample rank is not below the contig's number of valid starts");
    const uint64_t *ho = reinterpret_cast<const uint64_t *>(hs + o_out);
    for (uint32_t t = 0; t < n_tasks; ++t)
        for (uint32_t r = 0; r < 4; ++r)
            for (uint32_t col = 0; col < W; ++col)
                bg_counts[((size_t)t * 4 + r) * W + col] = (int64_t)ho[((size_t)t * 4 + r) * ws + col];
    lap(t_tail);
    if (timing)
        fprintf(stderr, "[nm_plan_windows] %u tasks, %llu samples (%s draws): base / row counts %.1f ms, plan %.1f ms, pools + staging + gather launch "
                        "%.1f ms, draws %.1f ms, background counts + wait %.1f ms\n", n_tasks, (unsigned long long)n_samples,
                drew_on_device ? "device" : "host", t_counts * 1e3, t_plan * 1e3, t_stage * 1e3, t_draws * 1e3, t_tail * 1e3);
    return NM_OK;
}

// Where on the device's clock the phases of nm_timing_reset(ctx, 1 | 2) lie: begin / end of each in ms after the first phase that
// `epoch` recorded (the ctx itself, or another ctx of the same device — several engines that work side by side are laid on one time
// line that way; phases on different streams may overlap, their union is the time the device was busy).
int nm_timing_intervals(nm_ctx *c, nm_ctx *epoch, uint64_t capacity, double *begin_ms, double *end_ms, uint64_t *n) {
    if (!c || !epoch || !n || (capacity && (!begin_ms || !end_ms))) return fail(NM_EINVAL, "NULL argument");
    *n = c->ev_used;
    if (!capacity) return NM_OK;
    if (capacity < c->ev_used) return fail(NM_EINVAL, "nm_timing_intervals: %zu phases recorded, room for %llu", c->ev_used, (unsigned long long)capacity);
    if (!epoch->ev_used) return fail(NM_ESTATE, "nm_timing_intervals: the epoch ctx has recorded no phase");
    hipEvent_t zero = epoch->ev_pool[0].first;
    HIP_TRY(hipEventSynchronize(zero));
    for (size_t i = 0; i < c->ev_used; ++i) {
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(c->ev_pool[i].second));
        HIP_TRY(hipEventElapsedTime(&a, zero, c->ev_pool[i].first));
        HIP_TRY(hipEventElapsedTime(&b, zero, c->ev_pool[i].second));
        begin_ms[i] = a;
        end_ms[i] = b;
    }
    return NM_OK;
}

}  // extern "C"
