// libnmscan — per-contig READ methylation of motifs: the table binnary starts from (reference: nanomotif/main.py:142-193,
// where it is produced by `epymetheus.methylation_pattern`, the Rust crate epimetheus-py 0.7.5 — a third-party
// dependency whose source is not part of the reference tree; its published behaviour is restated in
// oracle/contig_methylation.py, parity UNPINNED).  C ABI: nm_readstats_upload / nm_contig_methylation, include/nmscan.h.
//
// Per (contig, motif): the motif's sites on both strands (same scan as scoring: utils.py:44-67 semantics) that carry a
// pileup record passing the read filters; per site the read fraction n_modified / n_valid_cov; the row reports the number
// of such sites, their mean coverage, the MEDIAN of the fractions and their coverage-weighted mean.
//
// Layout: per mod code two PRESENCE planes over the padded coordinate space (bit = a passing record on '+' / '-'), a
// popcount-prefix rank table per 512 bp (per contig), and the records' (n_valid_cov, n_modified) in plane order — the
// value of a site is found by rank, nothing is stored per base pair.  A query scans the assembly once per pass with the
// scoring kernels' tile and constraint evaluation: pass 1 counts the sites of every (motif, contig) and sums their
// coverage, a prefix sum places the segments, pass 2 writes every site's fraction (IEEE bits: non-negative doubles order
// like their bit patterns) into its segment, a segmented radix sort (rocPRIM) orders each segment, and one thread per
// segment reads the middle.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "nmscan_device.h"

using namespace nmdetail;

namespace {

// (1) rows that pass the read filters set their bit in the presence plane of their strand
__global__ void rs_mark_kernel(uint64_t n_rows, const uint32_t *__restrict__ contig_id, const uint32_t *__restrict__ position,
                               const uint8_t *__restrict__ strand, const int32_t *__restrict__ n_valid,
                               const int32_t *__restrict__ n_diff, int32_t min_cov, double min_frac,
                               const uint32_t *__restrict__ contig_chunk, const uint64_t *__restrict__ contig_len,
                               uint32_t n_contigs, uint32_t *Pp, uint32_t *Pm, uint8_t *__restrict__ pass, unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    pass[i] = 0;
    const uint32_t cid = contig_id[i];
    if (cid == 0xFFFFFFFFu) return;                                   // contig not resident on this device
    const uint32_t pos = position[i];
    if (cid >= n_contigs || pos >= contig_len[cid]) { atomicOr(err, 1u); return; }
    const int32_t nv = n_valid[i];
    if (nv < min_cov || nv <= 0) return;                               // n_valid_cov >= min_valid_read_coverage
    if (n_diff) {
        const int32_t nd = n_diff[i];
        if (nd < 0) { atomicOr(err, 16u); return; }
        if ((double)nv / ((double)nv + (double)nd) < min_frac) return; // n_valid_cov / (n_valid_cov + n_diff) >= 0.8
    }
    const uint8_t st = strand[i];
    if (st != '+' && st != '-') { atomicOr(err, 2u); return; }
    const uint64_t g = (uint64_t)contig_chunk[cid] * CHUNK_BP + pos;
    const uint32_t bit = 1u << (g & 31);
    const uint32_t old = atomicOr((st == '+' ? Pp : Pm) + (g >> 5), bit);
    if (old & bit) { atomicOr(err, 4u); return; }                      // duplicate (contig, position, strand)
    pass[i] = 1;
}

// (2) one wave per contig: rank[block] = set bits of the contig before the 512-bp block; total per contig
__global__ __launch_bounds__(64) void rs_rank_kernel(const uint32_t *__restrict__ plane, const uint32_t *__restrict__ contig_chunk,
                                                     const uint64_t *__restrict__ contig_len, uint32_t *__restrict__ rank,
                                                     uint64_t *__restrict__ total) {
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    const uint32_t c0 = contig_chunk[ci];
    const uint32_t nblk = (uint32_t)((contig_len[ci] + GAP_BP + CHUNK_BP - 1) / CHUNK_BP) * RANK_PER_CHUNK;
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < nblk; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t cnt = 0;
        if (j < nblk) {
            const size_t w = (size_t)c0 * CHUNK_WORDS + (size_t)j * RANK_BLOCK_WORDS;
            for (int k = 0; k < RANK_BLOCK_WORDS; ++k) cnt += __popc(plane[w + k]);
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

// (3) every passing row stores its counts at its rank: values end up in plane (= position) order per strand
__global__ void rs_fill_kernel(uint64_t n_rows, const uint32_t *__restrict__ contig_id, const uint32_t *__restrict__ position,
                               const uint8_t *__restrict__ strand, const int32_t *__restrict__ n_valid,
                               const int32_t *__restrict__ n_mod, const uint8_t *__restrict__ pass,
                               const uint32_t *__restrict__ contig_chunk, const uint32_t *__restrict__ Pp,
                               const uint32_t *__restrict__ Pm, const uint32_t *__restrict__ rank_p,
                               const uint32_t *__restrict__ rank_m, const uint64_t *__restrict__ base_p,
                               const uint64_t *__restrict__ base_m, uint2 *__restrict__ val_p, uint2 *__restrict__ val_m,
                               unsigned int *err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows || !pass[i]) return;
    const uint32_t cid = contig_id[i];
    const bool plus = strand[i] == '+';
    const uint64_t g = (uint64_t)contig_chunk[cid] * CHUNK_BP + position[i];
    const uint32_t *P = plus ? Pp : Pm;
    const size_t w = (size_t)(g >> 5), w0 = w & ~(size_t)(RANK_BLOCK_WORDS - 1);
    uint64_t idx = (plus ? base_p : base_m)[cid] + (plus ? rank_p : rank_m)[w / RANK_BLOCK_WORDS];
    for (size_t k = w0; k < w; ++k) idx += __popc(P[k]);
    idx += __popc(P[w] & ((1u << (g & 31)) - 1u));
    const int32_t nm = n_mod[i];
    if (nm < 0 || nm > n_valid[i]) atomicOr(err, 32u);                 // n_modified outside [0, n_valid_cov]
    (plus ? val_p : val_m)[idx] = make_uint2((uint32_t)n_valid[i], (uint32_t)nm);
}

struct CmArgs {
    Planes seq;
    const uint32_t *Pp, *Pm;            // presence planes of the batch's mod code
    const uint32_t *rank_p, *rank_m;    // per 512-bp block, relative to the contig
    const uint64_t *base_p, *base_m;    // per contig: first value index
    const uint2 *val_p, *val_m;         // (n_valid_cov, n_modified) in plane order
    const uint32_t *chunk_contig;       // per chunk: contig id or ~0 for pad chunks
    uint32_t n_chunks, n_contigs, n_motifs;
    const uint32_t *programs;           // [n_motifs][2 * PDW]
    unsigned int *seg_cnt;              // [n_motifs * n_contigs] sites
    unsigned long long *seg_valid, *seg_mod;   // sums of n_valid_cov / n_modified over the sites
    const unsigned int *seg_off;        // EMIT: first key of every segment
    unsigned int *seg_fill;             // EMIT: keys written so far
    unsigned long long *keys;           // EMIT: IEEE bits of n_modified / n_valid_cov per site
};

// One wave per chunk; the tile is built once and serves every motif of the batch.  EMIT = false: pass 1 (counts and
// sums per (motif, contig)); EMIT = true: pass 2 (the fractions into their segments).
template <class K, bool EMIT>
__global__ __launch_bounds__(256) void cm_scan_kernel(CmArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunk = blockIdx.x * 4 + wave;
    if (chunk >= a.n_chunks) return;
    const uint32_t contig = ((cu32p)a.chunk_contig)[chunk];
    if (contig == 0xFFFFFFFFu) return;
    const StatePlanes stp[1] = {StatePlanes{nullptr, nullptr, a.Pp, a.Pp, a.Pm, a.Pm}};
    RawChunk<K> raw;
    raw.load(a.seq, stp, chunk, lane);
    Tile<K> tile;
    tile.expand(raw);
    // value index of the first record of this lane's words, per strand: contig base + block rank + the records of the
    // lanes before it in its 16-word block (a lane owns 4 words)
    uint64_t first[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t (&pw)[T_WORDS] = raw.s[0][s * 2];
        const uint32_t mine = __popc(pw[0]) + __popc(pw[1]) + __popc(pw[2]) + __popc(pw[3]);
        const uint32_t a1 = __shfl_up(mine, 1), a2 = __shfl_up(mine, 2), a3 = __shfl_up(mine, 3);
        const int q = lane & 3;
        const uint32_t before = (q >= 1 ? a1 : 0u) + (q >= 2 ? a2 : 0u) + (q >= 3 ? a3 : 0u);
        const uint32_t blk = chunk * RANK_PER_CHUNK + (uint32_t)(lane >> 2);
        first[s] = (s == 0 ? a.base_p : a.base_m)[contig] + (s == 0 ? a.rank_p : a.rank_m)[blk] + before;
    }
    for (uint32_t m = 0; m < a.n_motifs; ++m) {
        cu32p prog = (cu32p)(a.programs + (size_t)m * (2 * K::PDW));
        uint32_t sites[2][T_WORDS];
        uint32_t n_lane = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t acc[T_WORDS];
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) acc[t] = 0xFFFFFFFFu;
            eval_strand<K>(prog + s * K::PDW, tile, acc);
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) {
                sites[s][t] = acc[t] & raw.s[0][s * 2][t];
                n_lane += __popc(sites[s][t]);
            }
        }
        if (__ballot(n_lane != 0) == 0) continue;                      // wave-uniform: no site of this motif in the chunk
        const size_t seg = (size_t)m * a.n_contigs + contig;
        uint64_t sum_valid = 0, sum_mod = 0;
        unsigned long long *dst = nullptr;
        if (EMIT) {
            uint32_t x = n_lane;                                       // exclusive prefix of the lanes' site counts
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            const uint32_t total = __shfl(x, 63);
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(a.seg_fill + seg, total);
            at = __shfl(at, 0);
            dst = a.keys + a.seg_off[seg] + at + (x - n_lane);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint2 *val = s == 0 ? a.val_p : a.val_m;
            uint64_t idx = first[s];
#pragma unroll
            for (int t = 0; t < T_WORDS; ++t) {
                const uint32_t pw = raw.s[0][s * 2][t];
                uint32_t x = sites[s][t];
                while (x) {
                    const uint32_t b = (uint32_t)__builtin_ctz(x);
                    x &= x - 1;
                    const uint2 v = val[idx + __popc(pw & ((1u << b) - 1u))];
                    if (EMIT) *dst++ = (unsigned long long)__double_as_longlong((double)v.y / (double)v.x);
                    else { sum_valid += v.x; sum_mod += v.y; }
                }
                idx += __popc(pw);
            }
        }
        if (!EMIT) {
            uint32_t n = n_lane;
            for (int o = 32; o; o >>= 1) {
                n += __shfl_xor(n, o);
                sum_valid += __shfl_xor(sum_valid, o);
                sum_mod += __shfl_xor(sum_mod, o);
            }
            if (lane == 0) {
                atomicAdd(a.seg_cnt + seg, n);
                atomicAdd(a.seg_valid + seg, (unsigned long long)sum_valid);
                atomicAdd(a.seg_mod + seg, (unsigned long long)sum_mod);
            }
        }
    }
}

// one thread per (motif, contig) segment: the middle of its sorted fractions, the mean coverage, the weighted mean
__global__ void cm_finalize_kernel(uint32_t n_seg, const unsigned int *__restrict__ seg_cnt, const unsigned int *__restrict__ seg_off,
                                   const unsigned long long *__restrict__ seg_valid, const unsigned long long *__restrict__ seg_mod,
                                   const unsigned long long *__restrict__ sorted, uint32_t *__restrict__ out_n,
                                   double *__restrict__ out_cov, double *__restrict__ out_median, double *__restrict__ out_wmean) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t n = seg_cnt[s];
    out_n[s] = n;
    if (n == 0) { out_cov[s] = 0; out_median[s] = 0; out_wmean[s] = 0; return; }
    const unsigned long long *k = sorted + seg_off[s];
    const double hi = __longlong_as_double((long long)k[n / 2]);
    out_median[s] = (n & 1) ? hi : (__longlong_as_double((long long)k[n / 2 - 1]) + hi) / 2.0;
    out_cov[s] = (double)seg_valid[s] / (double)n;
    out_wmean[s] = (double)seg_mod[s] / (double)seg_valid[s];
}

// 64-bit sum of the per-segment site counts (the prefix sum that places the segments is 32 bits wide)
__global__ __launch_bounds__(256) void cm_total_kernel(const unsigned int *__restrict__ cnt, uint32_t n, unsigned long long *total) {
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += cnt[i];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(total, acc);
}

template <class K>
void launch_scan(const CmArgs &a, bool emit, hipStream_t s) {
    const dim3 grid((a.n_chunks + 3) / 4), blk(256);
    if (emit) hipLaunchKernelGGL((cm_scan_kernel<K, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((cm_scan_kernel<K, false>), grid, blk, 0, s, a);
}

}  // namespace

namespace nmdetail {

void free_readstats(nm_ctx *c) {
    for (auto &rs : c->readstats) {
        void *ptrs[] = {rs.planes, rs.rank[0], rs.rank[1], rs.base[0], rs.base[1], rs.val[0], rs.val[1]};
        for (void *p : ptrs)
            if (p) (void)dev_free(p);
        rs = ReadStats{};
    }
    if (c->d_chunk_contig) (void)dev_free(c->d_chunk_contig);
    c->d_chunk_contig = nullptr;
}

}  // namespace nmdetail

extern "C" {

int nm_readstats_upload(nm_ctx *c, uint32_t slot, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                        const uint8_t *strand, const int32_t *n_valid_cov, const int32_t *n_modified, const int32_t *n_diff,
                        int32_t min_valid_read_coverage, double min_valid_cov_to_diff_fraction, int rows_on_device, uint64_t *n_kept) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (slot >= NM_MAX_MOD_SLOTS) return fail(NM_EINVAL, "slot %u >= %d", slot, NM_MAX_MOD_SLOTS);
    if (n_rows && (!contig_id || !position || !strand || !n_valid_cov || !n_modified)) return fail(NM_EINVAL, "NULL column");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ReadStats &rs = c->readstats[slot];
    {
        void *old[] = {rs.planes, rs.rank[0], rs.rank[1], rs.base[0], rs.base[1], rs.val[0], rs.val[1]};
        for (void *p : old)
            if (p) (void)dev_free(p);
        rs = ReadStats{};
    }
    std::vector<void *> tmp;
    struct Guard {
        nm_ctx *c;
        std::vector<void *> &tmp;
        ReadStats &rs;
        bool keep = false;
        ~Guard() {
            (void)hipStreamSynchronize(c->stream);
            for (void *p : tmp) (void)dev_free(p);
            if (!keep) {
                void *old[] = {rs.planes, rs.rank[0], rs.rank[1], rs.base[0], rs.base[1], rs.val[0], rs.val[1]};
                for (void *p : old)
                    if (p) (void)dev_free(p);
                rs = ReadStats{};
            }
        }
    } guard{c, tmp, rs};
    const size_t words = plane_words(c);
    if (!c->d_chunk_contig) {                                          // per chunk: its contig (pad chunks: ~0)
        std::vector<uint32_t> cc(c->n_chunks, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < c->n_contigs; ++i)
            for (uint32_t k = 0; k < c->contig_nchunks[i]; ++k) cc[c->contig_chunk[i] + k] = i;
        HIP_TRY(dev_malloc(&c->d_chunk_contig, (size_t)c->n_chunks * 4));
        HIP_TRY(hipMemcpy(c->d_chunk_contig, cc.data(), (size_t)c->n_chunks * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(dev_malloc(&rs.planes, words * 4 * 2));
    HIP_TRY(hipMemsetAsync(rs.planes, 0, words * 4 * 2, c->stream));
    uint32_t *Pp = rs.planes, *Pm = rs.planes + words;
    // the raw columns on the device
    const void *src[6] = {contig_id, position, strand, n_valid_cov, n_modified, n_diff};
    const size_t esz[6] = {4, 4, 1, 4, 4, 4};
    const void *col[6];
    for (int k = 0; k < 6; ++k) {
        col[k] = src[k];
        if (rows_on_device || !src[k] || n_rows == 0) continue;
        void *d = nullptr;
        HIP_TRY(dev_malloc(&d, n_rows * esz[k]));
        tmp.push_back(d);
        HIP_TRY(hipMemcpyAsync(d, src[k], n_rows * esz[k], hipMemcpyHostToDevice, c->stream));
        col[k] = d;
    }
    uint8_t *d_pass = nullptr;
    HIP_TRY(dev_malloc(&d_pass, std::max<uint64_t>(n_rows, 1)));
    tmp.push_back(d_pass);
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(unsigned int), c->stream));
    const dim3 blk(256), grid((unsigned)((n_rows + 255) / 256));
    if (n_rows) {
        hipLaunchKernelGGL(rs_mark_kernel, grid, blk, 0, c->stream, n_rows, (const uint32_t *)col[0], (const uint32_t *)col[1],
                           (const uint8_t *)col[2], (const int32_t *)col[3], (const int32_t *)col[5], min_valid_read_coverage,
                           min_valid_cov_to_diff_fraction, c->d_contig_chunk, c->d_contig_len, c->n_contigs, Pp, Pm, d_pass, c->d_err);
        HIP_TRY(hipGetLastError());
    }
    uint64_t *d_total = nullptr;
    HIP_TRY(dev_malloc(&d_total, (size_t)std::max(c->n_contigs, 1u) * 8 * 2));
    tmp.push_back(d_total);
    for (int s = 0; s < 2; ++s) {
        HIP_TRY(dev_malloc(&rs.rank[s], (size_t)c->n_chunks * RANK_PER_CHUNK * 4));
        if (c->n_contigs) hipLaunchKernelGGL(rs_rank_kernel, dim3(c->n_contigs), dim3(64), 0, c->stream, s == 0 ? Pp : Pm, c->d_contig_chunk,
                                             c->d_contig_len, rs.rank[s], d_total + (size_t)s * c->n_contigs);
        HIP_TRY(hipGetLastError());
    }
    std::vector<uint64_t> total((size_t)c->n_contigs * 2, 0);
    unsigned int err = 0;
    if (c->n_contigs) HIP_TRY(hipMemcpyAsync(total.data(), d_total, total.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err & 1u) return fail(NM_EINVAL, "pileup row with contig_id / position outside the uploaded assembly");
    if (err & 2u) return fail(NM_EINVAL, "pileup strand must be '+' or '-'");
    if (err & 4u) return fail(NM_EINVAL, "duplicate (contig, position, strand) rows of one modification code");
    if (err & 16u) return fail(NM_EINVAL, "negative n_diff");
    std::vector<uint64_t> base((size_t)c->n_contigs * 2, 0);
    for (int s = 0; s < 2; ++s) {
        uint64_t at = 0;
        for (uint32_t i = 0; i < c->n_contigs; ++i) {
            base[(size_t)s * c->n_contigs + i] = at;
            at += total[(size_t)s * c->n_contigs + i];
        }
        rs.n_rows[s] = at;
        HIP_TRY(dev_malloc(&rs.base[s], (size_t)std::max(c->n_contigs, 1u) * 8));
        HIP_TRY(dev_malloc(&rs.val[s], std::max<uint64_t>(at, 1) * sizeof(uint2)));
        if (c->n_contigs) HIP_TRY(hipMemcpyAsync(rs.base[s], base.data() + (size_t)s * c->n_contigs, (size_t)c->n_contigs * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (n_rows) {
        hipLaunchKernelGGL(rs_fill_kernel, grid, blk, 0, c->stream, n_rows, (const uint32_t *)col[0], (const uint32_t *)col[1],
                           (const uint8_t *)col[2], (const int32_t *)col[3], (const int32_t *)col[4], d_pass, c->d_contig_chunk,
                           Pp, Pm, rs.rank[0], rs.rank[1], rs.base[0], rs.base[1], rs.val[0], rs.val[1], c->d_err);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err & 32u) return fail(NM_EINVAL, "n_modified outside [0, n_valid_cov]");
    rs.present = true;
    guard.keep = true;
    if (n_kept) *n_kept = rs.n_rows[0] + rs.n_rows[1];
    return NM_OK;
}

int nm_contig_methylation(nm_ctx *c, uint32_t n_motifs, const uint8_t *motif_slot, const uint8_t *motif_len, const uint8_t *motif_modpos,
                          const uint32_t *motif_mask_offset, const uint8_t *motif_masks, uint32_t *out_n_obs, double *out_mean_cov,
                          double *out_median, double *out_weighted_mean) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (!c->dH) return fail(NM_ESTATE, "nm_upload_contigs must come first");
    if (n_motifs == 0) return NM_OK;
    if (!motif_slot || !motif_len || !motif_modpos || !motif_mask_offset || !motif_masks || !out_n_obs || !out_mean_cov || !out_median || !out_weighted_mean)
        return fail(NM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t nc = c->n_contigs;
    if (nc == 0) return NM_OK;
    // programs: every motif on the general path (the modified position's own constraint included); grouped by how far they reach
    std::vector<uint32_t> full((size_t)n_motifs * PROG6_DW);
    std::vector<uint8_t> reach(n_motifs, 0);
    for (uint32_t m = 0; m < n_motifs; ++m) {
        if (motif_slot[m] >= NM_MAX_MOD_SLOTS || !c->readstats[motif_slot[m]].present)
            return fail(NM_ESTATE, "motif %u: read-statistics slot %u holds no pileup (nm_readstats_upload)", m, motif_slot[m]);
        int cls = 0;
        const int rc = compile_program(motif_masks + motif_mask_offset[m], motif_len[m], motif_modpos[m], full.data() + (size_t)m * PROG6_DW, &cls);
        if (rc) return rc;
        reach[m] = (uint8_t)cls;
    }
    // batches: motifs of one mod code, at most MB of them, all of one reach class (32 / 64 / 96 positions either side)
    constexpr uint32_t MB = 32;
    constexpr uint64_t MAX_KEYS = 1ull << 30;                          // 8 GB of keys (+ as much for the sorted copy) per batch
    std::vector<std::vector<uint32_t>> batches;
    for (uint32_t slot = 0; slot < NM_MAX_MOD_SLOTS; ++slot)
        for (int wide = 0; wide < 3; ++wide) {
            std::vector<uint32_t> cur;
            for (uint32_t m = 0; m < n_motifs; ++m)
                if (motif_slot[m] == slot && reach[m] == wide) {
                    cur.push_back(m);
                    if (cur.size() == MB) { batches.push_back(cur); cur.clear(); }
                }
            if (!cur.empty()) batches.push_back(cur);
        }
    std::vector<void *> tmp;
    struct Guard {
        nm_ctx *c;
        std::vector<void *> &tmp;
        ~Guard() {
            (void)hipStreamSynchronize(c->stream);
            for (void *p : tmp) (void)dev_free(p);
        }
    } guard{c, tmp};
    auto alloc = [&](void **p, size_t bytes) -> hipError_t {
        const hipError_t e = device_alloc(p, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) tmp.push_back(*p);
        return e;
    };
    const size_t seg_cap = (size_t)MB * nc;
    unsigned int *d_cnt = nullptr, *d_off = nullptr, *d_fill = nullptr;
    unsigned long long *d_valid = nullptr, *d_mod = nullptr, *d_total = nullptr;
    uint32_t *d_prog = nullptr, *d_on = nullptr;
    double *d_ocov = nullptr, *d_omed = nullptr, *d_owm = nullptr;
    HIP_TRY(alloc((void **)&d_cnt, (seg_cap + 1) * 4));
    HIP_TRY(alloc((void **)&d_off, (seg_cap + 1) * 4));
    HIP_TRY(alloc((void **)&d_fill, seg_cap * 4));
    HIP_TRY(alloc((void **)&d_valid, seg_cap * 8));
    HIP_TRY(alloc((void **)&d_mod, seg_cap * 8));
    HIP_TRY(alloc((void **)&d_total, 8));
    HIP_TRY(alloc((void **)&d_prog, (size_t)MB * PROG6_DW * 4));
    HIP_TRY(alloc((void **)&d_on, seg_cap * 4));
    HIP_TRY(alloc((void **)&d_ocov, seg_cap * 8));
    HIP_TRY(alloc((void **)&d_omed, seg_cap * 8));
    HIP_TRY(alloc((void **)&d_owm, seg_cap * 8));
    size_t scan_bytes = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_cnt, d_off, 0u, seg_cap + 1, rocprim::plus<unsigned int>(), c->stream));
    void *d_scan = nullptr;
    HIP_TRY(alloc(&d_scan, scan_bytes));
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
    void *d_sort_tmp = nullptr;
    uint64_t keys_cap = 0;
    size_t sort_cap = 0;
    for (size_t bi = 0; bi < batches.size(); ++bi) {
        std::vector<uint32_t> batch = batches[bi];
        const uint32_t nb = (uint32_t)batch.size();
        const int G = reach[batch[0]] + 1;                            // word-groups either side of the modified base
        const uint32_t pdw = 16u * G;                                 // dwords per strand on the 8-plane tile
        const size_t n_seg = (size_t)nb * nc;
        std::vector<uint32_t> prog((size_t)nb * 2 * pdw);
        for (uint32_t k = 0; k < nb; ++k) slice_program(full.data() + (size_t)batch[k] * PROG6_DW, G, prog.data() + (size_t)k * 2 * pdw);
        const ReadStats &rs = c->readstats[motif_slot[batch[0]]];
        const size_t words = plane_words(c);
        CmArgs a{};
        a.seq = seq_planes(c);
        a.Pp = rs.planes;
        a.Pm = rs.planes + words;
        a.rank_p = rs.rank[0]; a.rank_m = rs.rank[1];
        a.base_p = rs.base[0]; a.base_m = rs.base[1];
        a.val_p = rs.val[0]; a.val_m = rs.val[1];
        a.chunk_contig = c->d_chunk_contig;
        a.n_chunks = c->n_chunks;
        a.n_contigs = nc;
        a.n_motifs = nb;
        a.programs = d_prog;
        a.seg_cnt = d_cnt; a.seg_valid = d_valid; a.seg_mod = d_mod; a.seg_off = d_off; a.seg_fill = d_fill;
        HIP_TRY(hipMemcpyAsync(d_prog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, (n_seg + 1) * 4, c->stream));
        HIP_TRY(hipMemsetAsync(d_fill, 0, n_seg * 4, c->stream));
        HIP_TRY(hipMemsetAsync(d_valid, 0, n_seg * 8, c->stream));
        HIP_TRY(hipMemsetAsync(d_mod, 0, n_seg * 8, c->stream));
        // pass 1: sites and coverage sums per (motif, contig)
        if (G == 3) launch_scan<Variant<3, 3, false, 1, false, false>>(a, false, c->stream);
        else if (G == 2) launch_scan<Variant<2, 2, false, 1, false, false>>(a, false, c->stream);
        else launch_scan<Variant<1, 1, false, 1, false, false>>(a, false, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(d_total, 0, 8, c->stream));
        hipLaunchKernelGGL(cm_total_kernel, dim3(256), dim3(256), 0, c->stream, d_cnt, (uint32_t)n_seg, d_total);
        HIP_TRY(hipGetLastError());
        HIP_TRY(rocprim::exclusive_scan(d_scan, scan_bytes, d_cnt, d_off, 0u, n_seg + 1, rocprim::plus<unsigned int>(), c->stream));
        unsigned long long total = 0;
        HIP_TRY(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (total > MAX_KEYS && nb == 1 && total >= 0xFFFFFFFFull)
            return fail(NM_ERANGE, "motif %u has %llu sites with records: more than one sort can hold", batch[0], total);
        if (total > MAX_KEYS && nb > 1) {                              // too many sites at once: halve the batch
            std::vector<uint32_t> lo(batch.begin(), batch.begin() + nb / 2), hi(batch.begin() + nb / 2, batch.end());
            batches[bi] = lo;
            batches.insert(batches.begin() + bi + 1, hi);
            --bi;
            continue;
        }
        if (total) {
            if (keys_cap < total) {
                keys_cap = std::max<uint64_t>(total, keys_cap * 2);
                HIP_TRY(alloc((void **)&d_keys, keys_cap * 8));
                HIP_TRY(alloc((void **)&d_sorted, keys_cap * 8));
            }
            a.keys = d_keys;
            // pass 2: the fractions into their segments
            if (G == 3) launch_scan<Variant<3, 3, false, 1, false, false>>(a, true, c->stream);
            else if (G == 2) launch_scan<Variant<2, 2, false, 1, false, false>>(a, true, c->stream);
            else launch_scan<Variant<1, 1, false, 1, false, false>>(a, true, c->stream);
            HIP_TRY(hipGetLastError());
            size_t need = 0;
            HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, need, d_keys, d_sorted, (unsigned int)total, (unsigned int)n_seg, d_off, d_off + 1, 0, 64, c->stream));
            if (need > sort_cap) {
                sort_cap = need + need / 4;
                HIP_TRY(alloc(&d_sort_tmp, sort_cap));
            }
            size_t have = sort_cap;
            HIP_TRY(rocprim::segmented_radix_sort_keys(d_sort_tmp, have, d_keys, d_sorted, (unsigned int)total, (unsigned int)n_seg, d_off, d_off + 1, 0, 64, c->stream));
        }
        hipLaunchKernelGGL(cm_finalize_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, c->stream, (uint32_t)n_seg, d_cnt, d_off,
                           d_valid, d_mod, d_sorted, d_on, d_ocov, d_omed, d_owm);
        HIP_TRY(hipGetLastError());
        // dense rows of the batch's motifs back to the caller's [motif][contig] tables
        std::vector<uint32_t> h_n(n_seg);
        std::vector<double> h_cov(n_seg), h_med(n_seg), h_wm(n_seg);
        HIP_TRY(hipMemcpyAsync(h_n.data(), d_on, n_seg * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_cov.data(), d_ocov, n_seg * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_med.data(), d_omed, n_seg * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_wm.data(), d_owm, n_seg * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (uint32_t k = 0; k < nb; ++k) {
            const size_t o = (size_t)batch[k] * nc, i = (size_t)k * nc;
            memcpy(out_n_obs + o, h_n.data() + i, (size_t)nc * 4);
            memcpy(out_mean_cov + o, h_cov.data() + i, (size_t)nc * 8);
            memcpy(out_median + o, h_med.data() + i, (size_t)nc * 8);
            memcpy(out_weighted_mean + o, h_wm.data() + i, (size_t)nc * 8);
        }
    }
    return NM_OK;
}

}  // extern "C"
