// libnmscan — device-side bedMethyl parser (C ABI: nm_bed_parse_device / nm_bedcols_*, include/nmscan.h).
//
// Replaces, for plain-text pileups, the host parser of nmbed.cpp on the ingestion side of the hot path (reference:
// polars' scan_csv of the 18-column modkit pileup, nanomotif/dataload.py:15-34, 72-100).  At 1 Gbp the text is ~80 GB:
// a host parser needs every core of the box for tens of seconds where the search itself takes a fraction of one.  Here the
// host only moves bytes: file -> pinned slabs (threads) -> HBM (one stream), and the GPU does the rest per slab:
//   line starts   popcount of "first byte of a non-empty line" per 16 KB block, prefix sum, one offset per line;
//   fields        one thread per line: the six columns the reference keeps (1 contig, 2 start, 4 mod code, 6 strand,
//                 10 N_valid_cov, 11 percent) with the host parser's own rules — integers digit by digit, the percentage
//                 through Clinger's fast path (mantissa < 2^53, <= 22 decimals: ONE correctly rounded division, the same
//                 bits as strtod), nulls "NA" / "null" / empty -> -1;
//   contigs       a 64-bit hash of the name per row; a row whose hash differs from the row before starts a RUN — only the
//                 runs (one per contig in a modkit file) go back to the host, which reads their names from the mapped file
//                 and numbers them in first-appearance order;
//   leftovers     rows the kernel will not decide (a mod code other than m / a / 21839, a number outside the fast path)
//                 are listed and parsed by the host parser's routines (nmbed_parse.h), then patched in.
// The columns stay in HBM in exactly the types nm_ingest_pileup takes (rows_on_device = 1): no pileup row ever exists
// as a host array.  nmbed.cpp's parser is the bit-exactness oracle for every row (tests/test_gpu_bed_device.py).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <rocprim/device/device_scan.hpp>

#include <sys/mman.h>

#include "nmbed_parse.h"
#include "nmbgzf.h"
#include "nmscan_internal.h"
#include "nmres.h"

using namespace nmdetail;

namespace {

constexpr uint32_t BLOCK_BYTES = 16384;           // bytes per workgroup in the line-start passes (256 threads x 64 B)
constexpr uint64_t SLAB_BYTES = 32ull << 20;            // per slab: 3 pinned + 2 device buffers of this size (pinning memory costs ~0.2 ms per MB)
constexpr uint32_t PATCH_CAP = 1u << 22;

enum RowError : uint32_t { E_NONE = 0, E_COLUMNS = 1, E_START = 2, E_COV = 3, E_PCT = 4, E_POS_RANGE = 5, E_STRAND = 6, E_START_NEG = 7 };

// bit k of the result = byte k of the 64 bytes at `base` is the first byte of a non-empty line.  The 64 bytes come as four
// 16-byte loads (the text buffers are 16-byte aligned and padded); the bytes before and after them decide the edges.
__device__ __forceinline__ unsigned long long line_start_mask(const uint8_t *__restrict__ b, uint64_t base, uint64_t n) {
    if (base >= n) return 0;
    unsigned long long nl = 0, cr = 0;                                  // bit k: byte k is '\n' / '\r'
    const uint4 *v = reinterpret_cast<const uint4 *>(b + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 x = v[q];
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t c = (w[j] >> (8 * k)) & 0xFFu;
                nl |= (unsigned long long)(c == '\n') << (q * 16 + j * 4 + k);
                cr |= (unsigned long long)(c == '\r') << (q * 16 + j * 4 + k);
            }
    }
    const uint64_t left = n - base;                                     // valid bytes here
    const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    const bool prev_nl = base == 0 || b[base - 1] == '\n';
    const unsigned long long after_nl = (nl << 1) | (prev_nl ? 1ull : 0ull);           // byte k follows a newline (or starts the text)
    // "\r\n" alone is an empty line: a '\r' whose next byte is '\n' or the end of the text
    const bool next_nl = left <= 64 ? true : b[base + 64] == '\n';
    const unsigned long long before_nl = (nl >> 1) | ((left <= 64 ? (1ull << (left - 1)) : 0ull)) | (next_nl && left >= 64 ? (1ull << 63) : 0ull);
    return after_nl & ~nl & ~(cr & before_nl) & valid;
}

// (1) non-empty line starts per 16 KB block
__global__ __launch_bounds__(256) void bed_count_kernel(const uint8_t *__restrict__ b, uint64_t n, uint32_t *__restrict__ block_cnt) {
    __shared__ uint32_t part[4];
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK_BYTES + (uint64_t)threadIdx.x * 64;
    uint32_t cnt = (uint32_t)__popcll(line_start_mask(b, base, n));
    for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// (2) the offsets of those starts, in order
__global__ __launch_bounds__(256) void bed_starts_kernel(const uint8_t *__restrict__ b, uint64_t n, const uint32_t *__restrict__ block_off,
                                                         uint32_t *__restrict__ line_start) {
    __shared__ uint32_t scan[256];
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK_BYTES + (uint64_t)threadIdx.x * 64;
    unsigned long long mask = line_start_mask(b, base, n);
    const uint32_t mine = (uint32_t)__popcll(mask);
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)d ? scan[threadIdx.x - d] : 0;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t at = block_off[blockIdx.x] + scan[threadIdx.x] - mine;
    while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        line_start[at++] = (uint32_t)(base + k);
    }
}

#include "nmbedinflate.h"

// a slab's results for the host, written straight into pinned memory (no copy engine on the inflate stream: see parse_device_impl)
__global__ void bed_report_kernel(const unsigned long long *__restrict__ tail, unsigned long long total, const unsigned int *__restrict__ status,
                                  unsigned long long *h_end_of_lines, unsigned long long *h_status) {
    *h_end_of_lines = tail ? *tail : total;
    *h_status = *status;
    __threadfence_system();
}

// the byte after the last '\n' of text[0, n) (0 when there is none in the last `window` bytes): what lies behind it is the
// beginning of a line that continues in the next slab.  One workgroup, walking BACK from the end in pieces of 4 KiB (16 bytes per
// thread) and stopping at the first piece that holds a newline — the last line of a pileup slab ends a few dozen bytes from the end —
// (round 6: it read the whole window, 1.2 ms on the inflate stream of every slab)
__global__ __launch_bounds__(256) void bed_tail_kernel(const uint8_t *__restrict__ text, uint64_t n, uint64_t window, unsigned long long *out) {
    __shared__ unsigned long long best;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    const uint64_t lo = n > window ? n - window : 0;
    for (uint64_t hi = n; hi > lo;) {
        const uint64_t base = hi - lo > 4096 ? hi - 4096 : lo;
        unsigned long long mine = 0;
        const uint64_t a = base + 16ull * threadIdx.x;
        for (uint64_t i = a; i < a + 16 && i < hi; ++i)
            if (text[i] == '\n') mine = i + 1;
        if (mine) atomicMax(&best, mine);
        __syncthreads();
        const unsigned long long seen = best;
        __syncthreads();                                                // (nobody adds to `best` for the next piece before everybody has read it)
        if (seen) break;
        hi = base;
    }
    if (threadIdx.x == 0) *out = best;
}


struct BedOut {
    uint64_t *hash;             // [line of the slab] name hash (scratch of one slab)
    uint32_t *position;
    int8_t *mod;
    uint8_t *strand;
    double *frac;
    int32_t *nvalid;
    unsigned long long *first_error;   // min over (row << 8 | code)
    unsigned int *n_patch;
    uint4 *patch;               // (row, flags, file offset lo, hi) — a pileup holds fewer than 2^32 rows (checked)
};

__device__ __forceinline__ bool field_is_null(const uint8_t *p, uint32_t n) {
    return n == 0 || (n == 2 && p[0] == 'N' && p[1] == 'A') || (n == 4 && p[0] == 'n' && p[1] == 'u' && p[2] == 'l' && p[3] == 'l');
}

__device__ __forceinline__ bool dev_parse_int(const uint8_t *p, uint32_t n, long long *out) {
    if (n == 0) return false;
    bool neg = false;
    uint32_t i = 0;
    if (p[0] == '-') { neg = true; i = 1; }
    long long v = 0;
    for (; i < n; ++i) {
        const uint32_t d = (uint32_t)p[i] - '0';
        if (d > 9) return false;
        v = v * 10 + d;
    }
    *out = neg ? -v : v;
    return true;
}

// bit k of the result: byte k of the eight bytes (lo, hi) equals the byte repeated in `cccc`.  Exact for every byte value: the byte-wise
// difference is zero <=> its low seven bits are zero (0x80 - them keeps bit 7, no borrow between bytes) and its bit 7 is clear; the
// eight marks (0x80 per byte) are weighted 1, 2, 4 ... 128 and summed by two v_dot4_u32_u8
__device__ __forceinline__ uint32_t bytes_equal_mask8(uint32_t lo, uint32_t hi, uint32_t cccc) {
    const uint32_t t0 = lo ^ cccc, t1 = hi ^ cccc;
    const uint32_t z0 = (0x80808080u - (t0 & 0x7F7F7F7Fu)) & ~t0 & 0x80808080u;
    const uint32_t z1 = (0x80808080u - (t1 & 0x7F7F7F7Fu)) & ~t1 & 0x80808080u;
    return __builtin_amdgcn_udot4(z1, 0x80402010u, __builtin_amdgcn_udot4(z0, 0x08040201u, 0u, false), false) >> 7;
}

// The fields of one line, as byte offsets from its first byte: b/e of the six columns that are read (1 contig, 2 start, 4 mod code,
// 6 strand, 10 N_valid_cov, 11 percent; numbered from 0 here) and the number of tab-separated fields up to the end of the line.
struct BedFields { uint32_t b0, e0, b1, e1, b3, e3, b5, e5, b9, e9, b10, e10, nf; };

// The line's tabs WITHOUT walking it byte by byte (round 6: the walk — ~80 dependent byte loads per line, a dozen instructions each —
// was two thirds of this kernel, and with the inflate kernels no longer waiting on memory beside it, this kernel became the largest
// consumer of the GPU in a from-files run): the line is loaded in pieces of 16 bytes (unaligned 16-byte loads), every piece gives 16
// bits "is a tab" and 16 bits "is a newline", the first newline ends the line, and the tabs below it are taken out of a 128-bit mask one
// by one.  false: no newline within 128 bytes (bed_split_walk does such lines).
__device__ __forceinline__ bool bed_split_masks(const uint8_t *__restrict__ p, uint64_t avail, BedFields &f) {
    const uint32_t want = avail < 128 ? (uint32_t)avail : 128u;
    unsigned long long tab[2] = {0, 0}, nl[2] = {0, 0};
    bool open = true;                                                   // no newline seen yet
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!__any(open && 16u * (unsigned)k < want)) break;            // (wave-uniform)
        if (open && 16u * (unsigned)k < want) {
            uint4 q;
            memcpy(&q, p + 16 * k, 16);                                 // (the text buffers are padded: up to 15 bytes behind the text may be read)
            const unsigned long long t = bytes_equal_mask8(q.x, q.y, 0x09090909u) | (bytes_equal_mask8(q.z, q.w, 0x09090909u) << 8);
            const unsigned long long e = bytes_equal_mask8(q.x, q.y, 0x0A0A0A0Au) | (bytes_equal_mask8(q.z, q.w, 0x0A0A0A0Au) << 8);
            tab[k >> 2] |= t << (16 * (k & 3));
            nl[k >> 2] |= e << (16 * (k & 3));
            open = e == 0;
        }
    }
    // bytes behind the text are not the line's
    if (want < 64) { const unsigned long long m = (1ull << want) - 1; nl[0] &= m; nl[1] = 0; }
    else if (want < 128) nl[1] &= (1ull << (want - 64)) - 1;
    uint32_t end;
    if (nl[0]) end = (uint32_t)__builtin_ctzll(nl[0]);
    else if (nl[1]) end = 64u + (uint32_t)__builtin_ctzll(nl[1]);
    else if (avail <= 128) end = want;                                  // the last line of the text, without a newline
    else return false;
    if (end < 64) { tab[0] &= (1ull << end) - 1; tab[1] = 0; }
    else if (end < 128) tab[1] &= (1ull << (end - 64)) - 1;
    f.nf = (uint32_t)__popcll(tab[0]) + (uint32_t)__popcll(tab[1]) + 1u;
    f.b0 = 0; f.e0 = f.b1 = f.e1 = f.b3 = f.e3 = f.b5 = f.e5 = f.b9 = f.e9 = f.b10 = f.e10 = 0;
    if (f.nf < 12) return true;                                         // (fewer than eleven tabs: the caller refuses the line)
    uint32_t t[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        if (tab[0]) { t[k] = (uint32_t)__builtin_ctzll(tab[0]); tab[0] &= tab[0] - 1; }
        else { t[k] = 64u + (uint32_t)__builtin_ctzll(tab[1]); tab[1] &= tab[1] - 1; }
    }
    f.e0 = t[0];
    f.b1 = t[0] + 1; f.e1 = t[1];
    f.b3 = t[2] + 1; f.e3 = t[3];
    f.b5 = t[4] + 1; f.e5 = t[5];
    f.b9 = t[8] + 1; f.e9 = t[9];
    f.b10 = t[9] + 1; f.e10 = t[10];
    return true;
}

// the same by walking the line (lines of more than 128 bytes)
__device__ __forceinline__ void bed_split_walk(const uint8_t *__restrict__ p, const uint8_t *__restrict__ end, BedFields &f) {
    uint32_t b0 = 0, e0 = 0, b1 = 0, e1 = 0, b3 = 0, e3 = 0, b5 = 0, e5 = 0, b9 = 0, e9 = 0, b10 = 0, e10 = 0;
    uint32_t nf = 0, at = 0, start = 0;
    for (;;) {
        const bool stop = p + at >= end || p[at] == '\n';
        if (stop || p[at] == '\t') {
            switch (nf) {
                case 0: b0 = start; e0 = at; break;
                case 1: b1 = start; e1 = at; break;
                case 3: b3 = start; e3 = at; break;
                case 5: b5 = start; e5 = at; break;
                case 9: b9 = start; e9 = at; break;
                case 10: b10 = start; e10 = at; break;
                default: break;
            }
            ++nf;
            if (stop) break;
            start = at + 1;
        }
        ++at;
    }
    f = BedFields{b0, e0, b1, e1, b3, e3, b5, e5, b9, e9, b10, e10, nf};
}

// (3) one thread per line
__global__ __launch_bounds__(256) void bed_parse_kernel(const uint8_t *__restrict__ b, uint64_t n, const uint32_t *__restrict__ line_start,
                                                        uint32_t n_lines, uint64_t row0, uint64_t slab_file_off, BedOut o) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_lines) return;
    const uint64_t row = row0 + li;
    const uint8_t *p = b + line_start[li];
    const uint8_t *end = b + n;
    // a bedMethyl row has exactly 18 tab-separated columns (dataload.py:15-34: the reference reads the file against a fixed 18-column
    // schema), so every tab of the line is counted
    BedFields f;
    if (!bed_split_masks(p, n - line_start[li], f)) bed_split_walk(p, end, f);
    const uint32_t b0 = f.b0, e0 = f.e0, b1 = f.b1, e1 = f.e1, b3 = f.b3, e3 = f.e3, b5 = f.b5, e5 = f.e5, b9 = f.b9, e9 = f.e9, b10 = f.b10, e10 = f.e10;
    const uint32_t nf = f.nf;
    // ("\r\n" line ends: the '\r' can only sit in the eighteenth field, which is not read)
    uint32_t err = E_NONE;
    if (nf != 18) err = E_COLUMNS;
    // contig name: FNV-1a over the bytes, the length folded in
    unsigned long long h = 1469598103934665603ull;
    if (!err) {
        for (uint32_t k = b0; k < e0; ++k) h = (h ^ p[k]) * 1099511628211ull;
        h = (h ^ (unsigned long long)(e0 - b0)) * 1099511628211ull;
    }
    long long pos = 0, cov = -1;
    double frac = -1.0;
    uint32_t flags = 0;                                                 // 1: mod code for the host, 2: percentage for the host
    int8_t mt = -1;
    uint8_t st = '?';
    if (!err) {
        if (!dev_parse_int(p + b1, e1 - b1, &pos)) err = E_START;
        else if (pos < 0) err = E_START_NEG;
        else if (pos > 0xFFFFFFFEll) err = E_POS_RANGE;
    }
    if (!err && !field_is_null(p + b9, e9 - b9) && !dev_parse_int(p + b9, e9 - b9, &cov)) err = E_COV;
    if (!err) {
        const uint8_t *q = p + b10;
        const uint32_t ln = e10 - b10;
        if (!field_is_null(q, ln)) {
            // Clinger's fast path, exactly as nmbedparse::parse_double takes it; everything else is the host's
            uint32_t i = 0;
            bool neg = false, dot = false, ok = true;
            if (q[0] == '-' || q[0] == '+') { neg = q[0] == '-'; i = 1; }
            unsigned long long mant = 0;
            int ndig = 0, dec = 0;
            for (; i < ln; ++i) {
                const uint32_t d = (uint32_t)q[i] - '0';
                if (d <= 9) {
                    if (mant > (0xFFFFFFFFFFFFFFFFull - 9) / 10) { ok = false; break; }
                    mant = mant * 10 + d;
                    dec += dot;
                    ++ndig;
                } else if (q[i] == '.' && !dot) {
                    dot = true;
                } else { ok = false; break; }
            }
            ok = ok && ndig > 0 && mant < (1ull << 53) && dec <= 22;
            if (ok) {
                double p10 = 1.0;
                for (int k = 0; k < dec; ++k) p10 *= 10.0;               // exact up to 1e22
                const double v = (double)mant / p10;
                frac = (neg ? -v : v) / 100.0;                           // dataload.py:85
            } else {
                flags |= 2u;
            }
        }
        const uint32_t ml = e3 - b3;
        const uint8_t *m = p + b3;
        if (ml == 1 && m[0] == 'm') mt = 0;
        else if (ml == 1 && m[0] == 'a') mt = 1;
        else if (ml == 5 && m[0] == '2' && m[1] == '1' && m[2] == '8' && m[3] == '3' && m[4] == '9') mt = 2;
        else flags |= 1u;
        // '+' or '-' and nothing else: the scoring path compares the column with exactly these (find_motifs_bin.py:1308-1314)
        if (e5 - b5 == 1 && (p[b5] == '+' || p[b5] == '-')) st = p[b5];
        else err = E_STRAND;
    }
    if (err) {
        atomicMin(o.first_error, ((unsigned long long)row << 8) | err);
        o.hash[li] = 0;
        return;
    }
    o.hash[li] = h;
    o.position[row] = (uint32_t)pos;
    o.mod[row] = mt;
    o.strand[row] = st;
    o.frac[row] = frac;
    o.nvalid[row] = cov < 0 ? -1 : (int32_t)(cov > 0x7FFFFFFF ? 0x7FFFFFFF : cov);
    if (flags) {
        const unsigned int k = atomicAdd(o.n_patch, 1u);
        const uint64_t off = slab_file_off + line_start[li];
        if (k < PATCH_CAP) o.patch[k] = make_uint4((uint32_t)row, flags, (uint32_t)off, (uint32_t)(off >> 32));
    }
}

// (4) rows whose contig name differs from the row before: the starts of the runs, found slab by slab (the hashes and the
// line starts of a slab are scratch).  prev_in / prev_out: the hash of the last row of the previous / of this slab.
struct BedRun { unsigned long long row, off; };

// A run's contig name goes along in a 64-byte slot (length, then up to 63 bytes; length 0xFF: longer than that — the host reads it from the
// file): the host used to fetch every run's name from the file — for a bgzip pileup one inflated BGZF block per run, 0.09 s for the 10 000
// contigs of a 1 Gbp metagenome (NM_BED_TIMING, round 6).
constexpr uint32_t RUN_NAME_SLOT = 64;
__global__ void bed_runs_kernel(const uint64_t *__restrict__ hash, const uint32_t *__restrict__ line_start, uint32_t n_lines, uint64_t row0,
                                uint64_t slab_file_off, int first_slab, const unsigned long long *__restrict__ prev_in,
                                unsigned long long *__restrict__ prev_out, unsigned int *n_runs, BedRun *runs, uint32_t cap,
                                const uint8_t *__restrict__ text, uint64_t len, uint8_t *__restrict__ run_names, uint32_t name_cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const unsigned long long before = i ? hash[i - 1] : *prev_in;
    if ((i == 0 && first_slab) || hash[i] != before) {
        const unsigned int k = atomicAdd(n_runs, 1u);
        if (k < cap) runs[k] = BedRun{row0 + i, slab_file_off + line_start[i]};
        if (k < name_cap) {
            uint8_t *slot = run_names + (size_t)k * RUN_NAME_SLOT;
            const uint8_t *p = text + line_start[i];
            const uint64_t room = len - line_start[i];
            uint32_t n = 0;
            while (n < RUN_NAME_SLOT - 1 && n < room && p[n] != '\t' && p[n] != '\n') { slot[1 + n] = p[n]; ++n; }
            slot[0] = (n < room && (p[n] == '\t' || p[n] == '\n')) ? (uint8_t)n : (uint8_t)0xFF;
        }
    }
    if (i == n_lines - 1) *prev_out = hash[i];
}

// (5) contig column from the run table (ascending run starts): id of the last run that starts at or before the row
__global__ void bed_fill_contig_kernel(uint64_t n_rows, const unsigned long long *__restrict__ run_row, const uint32_t *__restrict__ run_id,
                                       uint32_t n_runs, uint32_t *__restrict__ contig) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    uint32_t lo = 0, hi = n_runs - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (run_row[mid] <= i) lo = mid; else hi = mid - 1;
    }
    contig[i] = run_id[lo];
}

// device -> pinned host memory by a kernel (16-byte words): the first copy-engine transfer in that direction on a stream costs ~7 ms
__global__ void bed_copy_out_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

__global__ void bed_patch_kernel(uint32_t n, const unsigned long long *__restrict__ row, const int8_t *__restrict__ mod, const double *__restrict__ frac,
                                 const uint8_t *__restrict__ what, int8_t *__restrict__ out_mod, double *__restrict__ out_frac) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (what[k] & 1) out_mod[row[k]] = mod[k];
    if (what[k] & 2) out_frac[row[k]] = frac[k];
}

__global__ void bed_map_contigs_kernel(uint64_t n_rows, const uint32_t *__restrict__ lut, const uint32_t *__restrict__ file_id, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) out[i] = lut[file_id[i]];
}

}  // namespace

struct nm_bedcols {
    nm_ctx *ctx = nullptr;
    uint64_t n_rows = 0, cap = 0;
    uint32_t *d_file_contig = nullptr, *d_contig = nullptr, *d_position = nullptr;
    int8_t *d_mod = nullptr;
    uint8_t *d_strand = nullptr;
    double *d_frac = nullptr;
    int32_t *d_nvalid = nullptr;
    std::vector<std::string> names, other_mods;
    std::vector<const char *> name_ptrs;
    std::vector<uint64_t> run_row;          // ascending, + n_rows at the end
    std::vector<uint32_t> run_contig;
    double t_read = 0, t_total = 0, t_inflate = 0;
    bool read_beside = false;               // t_read was spent on a staging thread, beside the calling thread (device-inflate pipeline, round 6)
};

namespace {

void free_cols(nm_bedcols *b) {
    void *ptrs[] = {b->d_file_contig, b->d_contig, b->d_position, b->d_mod, b->d_strand, b->d_frac, b->d_nvalid};
    for (void *p : ptrs)
        if (p) (void)dev_free(p);
    b->d_file_contig = b->d_contig = b->d_position = nullptr;
    b->d_mod = nullptr; b->d_strand = nullptr; b->d_frac = nullptr; b->d_nvalid = nullptr;
}

// grow the six output columns (+ hash) to hold `rows`; device-to-device copies of what is there
int grow(nm_bedcols *b, uint64_t rows, hipStream_t s) {
    if (rows <= b->cap) return NM_OK;
    const uint64_t ncap = std::max<uint64_t>(rows, b->cap + b->cap / 2);
    auto regrow = [&](void **p, size_t esz) -> int {
        void *q = nullptr;
        HIP_TRY(device_alloc(&q, ncap * esz));
        if (*p && b->n_rows) HIP_TRY(hipMemcpyAsync(q, *p, b->n_rows * esz, hipMemcpyDeviceToDevice, s));
        if (*p) {
            HIP_TRY(hipStreamSynchronize(s));
            (void)dev_free(*p);
        }
        *p = q;
        return NM_OK;
    };
    int rc;
    if ((rc = regrow((void **)&b->d_position, 4)) || (rc = regrow((void **)&b->d_mod, 1)) || (rc = regrow((void **)&b->d_strand, 1)) ||
        (rc = regrow((void **)&b->d_frac, 8)) || (rc = regrow((void **)&b->d_nvalid, 4)))
        return rc;
    b->cap = ncap;
    return NM_OK;
}

const char *row_error_text(uint32_t code) {
    switch (code) {
        case E_COLUMNS: return "pileup line that does not have exactly 18 tab-separated columns (modkit bedMethyl)";
        case E_START: return "pileup column 2 (start) is not an integer";
        case E_COV: return "pileup column 10 (Nvalid_cov) is not an integer";
        case E_PCT: return "pileup column 11 (percent modified) is not a number";
        case E_POS_RANGE: return "pileup position beyond 4 Gbp";
        case E_STRAND: return "pileup column 6 (strand) is neither '+' nor '-'";
        case E_START_NEG: return "pileup column 2 (start) is negative";
        default: return "malformed pileup line";
    }
}

}  // namespace

extern "C" {

int nm_bedcols_close(nm_bedcols *b) {
    if (!b) return NM_OK;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    free_cols(b);
    delete b;
    return NM_OK;
}

}  // extern "C"

namespace {

// Where the bedMethyl TEXT comes from: a plain file (pread) or the wanted parts of the BGZF blocks of a bgzip file
// (nmbgzf.h: every block, or the blocks a tabix index names), inflated by the copy threads straight into the pinned slabs.
// Offsets are offsets into the text either way.
struct TextSource {
    int fd = -1;
    uint64_t n = 0;                               // bytes of text
    bool bgzf = false;
    const uint8_t *z = nullptr;                   // the mapped bgzip file
    size_t zn = 0;
    std::vector<nmbgzf::Piece> pieces;            // sorted by text_off, contiguous in the text
    ~TextSource() {
        // in pieces of 256 MiB: one munmap of a 14 GB pileup (3.5 million page-table entries where the copying threads' MADV_DONTNEED
        // did not take) holds the address-space lock for 0.2 s, and every allocation of the other threads — the pre-filters' scratch —
        // waits behind it; between two pieces they get their turn
        if (z) {
            constexpr size_t PIECE = 256u << 20;
            uint8_t *base = const_cast<uint8_t *>(z);
            for (size_t off = 0; off < zn; off += PIECE) munmap(base + off, std::min(PIECE, zn - off));
        }
        if (fd >= 0) close(fd);
    }
    size_t piece_at(uint64_t off) const {          // the piece that holds text offset `off`
        size_t lo = 0, hi = pieces.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (pieces[mid].text_off <= off) lo = mid;
            else hi = mid;
        }
        return lo;
    }
    // pieces first, first + step, ... that overlap the text range [off, off + len) -> dst (which holds the range)
    bool read_pieces(uint64_t off, uint64_t len, size_t first, size_t step, uint8_t *dst, std::vector<char> &tmp) const {
        for (size_t i = first; i < pieces.size() && pieces[i].text_off < off + len; i += step) {
            const nmbgzf::Piece &p = pieces[i];
            const uint64_t a = std::max<uint64_t>(off, p.text_off), e = std::min<uint64_t>(off + len, p.text_off + p.take);
            if (e <= a) continue;
            if (a == p.text_off && e == p.text_off + p.take) {
                if (!nmbgzf::inflate_piece(z, p, reinterpret_cast<char *>(dst + (a - off)), tmp)) return false;
            } else {                              // a piece cut by the range's end: its text through tmp
                tmp.resize(p.out_len);
                if (!nmbgzf::inflate_raw(z + p.in_off, p.in_len, tmp.data(), p.out_len, p.crc)) return false;
                memcpy(dst + (a - off), tmp.data() + p.skip + (a - p.text_off), (size_t)(e - a));
            }
        }
        return true;
    }
    bool read_at(uint64_t off, void *dst, uint64_t len) const {
        if (!bgzf) {
            uint8_t *d = static_cast<uint8_t *>(dst);
            while (len) {
                const ssize_t k = pread(fd, d, (size_t)std::min<uint64_t>(len, 1u << 30), (off_t)off);
                if (k <= 0) return false;
                d += k; off += (uint64_t)k; len -= (uint64_t)k;
            }
            return true;
        }
        if (len == 0) return true;
        if (off + len > n) return false;
        std::vector<char> tmp;
        return read_pieces(off, len, piece_at(off), 1, static_cast<uint8_t *>(dst), tmp);
    }
    // share t of nt of the range -> dst: a byte range of a plain file, every nt-th block of a bgzip file
    bool fill(uint64_t off, uint64_t len, unsigned t, unsigned nt, uint8_t *dst, std::vector<char> &tmp) const {
        if (!bgzf) {
            const uint64_t a = len * t / nt, e = len * (t + 1) / nt;
            return e <= a || read_at(off + a, dst + a, e - a);
        }
        if (len == 0) return true;
        return read_pieces(off, len, piece_at(off) + t, nt, dst, tmp);
    }
};

int map_file(const char *path, TextSource *src) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(NM_EINVAL, "cannot open pileup '%s'", path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(NM_EINVAL, "cannot stat pileup '%s'", path); }
    src->zn = (size_t)st.st_size;
    if (src->zn) {
        void *m = mmap(nullptr, src->zn, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); src->zn = 0; return fail(NM_EINVAL, "cannot map pileup '%s'", path); }
        src->z = static_cast<const uint8_t *>(m);
    }
    if (src->fd >= 0) close(src->fd);
    src->fd = fd;                                      // kept: the walk over the BGZF blocks reads their headers through it (nmbgzf.h)
    return NM_OK;
}

// *row_error (may be NULL): NM_EINVAL came from the TEXT — a line that is no bedMethyl row — not from the file, its blocks or an argument
int parse_device_impl(nm_ctx *c, const char *path, TextSource &src, uint32_t threads, nm_bedcols **out, bool *row_error = nullptr);

}  // namespace

struct nm_bedplan {                          // nm_bed_plan_indexed: the pieces of the wanted text, the file they sit in
    std::string path;
    TextSource src;
    std::unordered_map<std::string, uint32_t> want;
    uint64_t stats[4] = {0, 0, 0, 0};
};

extern "C" {

int nm_bed_parse_device(nm_ctx *c, const char *path, uint32_t threads, nm_bedcols **out) {
    if (!c || !path || !out) return fail(NM_EINVAL, "NULL argument");
    *out = nullptr;
    TextSource src;
    src.fd = open(path, O_RDONLY);
    if (src.fd < 0) return fail(NM_EINVAL, "cannot open pileup '%s'", path);
    struct stat st;
    if (fstat(src.fd, &st) != 0) return fail(NM_EINVAL, "cannot stat pileup '%s'", path);
    src.n = (uint64_t)st.st_size;
    // the file is READ (pread straight into the pinned slabs), not mapped: mapping 8 GB costs two million page faults on the
    // way in and as many page-table entries on the way out
    uint8_t magic[2] = {0, 0};
    if (src.n >= 2 && !src.read_at(0, magic, 2)) return fail(NM_EINVAL, "cannot read pileup '%s'", path);
    if (src.n >= 2 && magic[0] == 31 && magic[1] == 139) {
        // bgzip (what the reference recommends, docs/source/required_files.md:21): the blocks are inflated by the copy threads
        // into the same pinned slabs; any other gzip stream has no block structure to inflate in parallel
        close(src.fd);
        src.fd = -1;
        int rc = map_file(path, &src);
        if (rc) return rc;
        if (!nmbgzf::whole_file(src.z, src.zn, &src.pieces, &src.n))
            return fail(NM_EDECLINED, "%s: compressed input that is not bgzip: the device parser reads plain text and BGZF (use nm_bed_open)", path);
        src.bgzf = true;
    }
    return parse_device_impl(c, path, src, threads, out);
}

// The HOST-ONLY half of the indexed parse — the tabix index read, the wanted contigs' regions, the walk over their BGZF blocks (half a
// second of page faults for the pileup of a 1 Gbp metagenome) — as a call of its own: no GPU is involved, so a caller can run it on a
// thread while the HIP runtime comes up and the assembly is parsed (python -m nanomotif_amd does).
int nm_bed_plan_indexed(const char *path, const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset, uint32_t threads,
                        nm_bedplan **out, uint64_t stats[4]) {
    if (!path || !tbi_path || !out || (n_contigs && (!names || !name_offset))) return fail(NM_EINVAL, "NULL argument");
    *out = nullptr;
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // the index: a small bgzip file itself
    std::vector<char> index;
    {
        TextSource idx;
        int rc = map_file(tbi_path, &idx);
        if (rc) return fail(NM_EINDEX, "cannot open tabix index '%s'", tbi_path);
        if (idx.zn >= 2 && idx.z[0] == 31 && idx.z[1] == 139) {
            if (!nmbgzf::whole_file(idx.z, idx.zn, &idx.pieces, &idx.n)) return fail(NM_EINDEX, "%s: not a tabix index", tbi_path);
            idx.bgzf = true;
            index.resize(idx.n);
            if (idx.n && !idx.read_at(0, index.data(), idx.n)) return fail(NM_EINDEX, "%s: not a tabix index (corrupt BGZF block: deflate stream, size or CRC-32)", tbi_path);
        } else {
            index.assign(reinterpret_cast<const char *>(idx.z), reinterpret_cast<const char *>(idx.z) + idx.zn);
        }
    }
    nm_bedplan *p = new (std::nothrow) nm_bedplan();
    if (!p) return fail(NM_ENOMEM, "out of host memory");
    struct Guard { nm_bedplan *p; bool keep = false; ~Guard() { if (!keep) delete p; } } guard{p};
    p->path = path;
    for (uint32_t i = 0; i < n_contigs; ++i) p->want.emplace(std::string(names + name_offset[i], name_offset[i + 1] - name_offset[i]), i);
    std::vector<nmbgzf::Region> merged;
    std::vector<uint64_t> block_starts;
    uint64_t found = 0, inflated = 0;
    {
        const std::string what = nmbgzf::tabix_regions(reinterpret_cast<const uint8_t *>(index.data()), index.size(), p->want, &merged, &found, &block_starts);
        if (!what.empty()) return fail(NM_EINDEX, "%s: %s", tbi_path, what.c_str());
    }
    int rc = map_file(path, &p->src);
    if (rc) return rc;
    {
        // (the walk reads the block headers through the MAPPING: two small preads per block were measured at twice the page faults' time)
        bool index_problem = false;
        const std::string what = nmbgzf::region_pieces(p->src.z, p->src.zn, merged, &p->src.pieces, &p->src.n, &inflated, &block_starts,
                                                       threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency())), -1, &index_problem);
        if (!what.empty()) return fail(index_problem ? NM_EINDEX : NM_EINVAL, "%s: %s", path, what.c_str());
    }
    p->src.bgzf = true;
    p->stats[0] = inflated; p->stats[1] = p->src.zn; p->stats[2] = n_contigs - std::min<uint64_t>(found, n_contigs);
    p->stats[3] = (uint64_t)((std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin) * 1e6);   // microseconds in this call
    if (stats) for (int i = 0; i < 4; ++i) stats[i] = p->stats[i];
    guard.keep = true;
    *out = p;
    return NM_OK;
}

int nm_bedplan_close(nm_bedplan *p) {
    delete p;
    return NM_OK;
}

int nm_bed_parse_device_planned(nm_ctx *c, nm_bedplan *p, uint32_t threads, nm_bedcols **out) {
    if (!c || !p || !out) return fail(NM_EINVAL, "NULL argument");
    *out = nullptr;
    const char *path = p->path.c_str();
    bool row_error = false;
    int rc = parse_device_impl(c, path, p->src, threads, out, &row_error);
    if (rc == NM_EINVAL && row_error) {
        // blocks intact, LINES that do not parse (and only that: a damaged block, an unreadable file or a bad argument come back
        // as themselves): a region that starts or ends inside a line — the index is stale
        const std::string why = nm_last_error();
        return fail(NM_EINDEX, "%s: the text the tabix index names does not parse (%s): stale .tbi?", path, why.c_str());
    }
    if (rc) return rc;
    // the text the index pointed at must belong to the contigs that were asked for: a stale or foreign .tbi otherwise
    // yields a silently wrong subset of rows
    for (const std::string &nm : (*out)->names)
        if (!p->want.count(nm)) {
            const std::string culprit = nm;
            (void)nm_bedcols_close(*out);
            *out = nullptr;
            return fail(NM_EINDEX, "%s: the tabix index does not match the pileup (rows of contig '%s' where another contig was indexed): stale .tbi?",
                        path, culprit.c_str());
        }
    return NM_OK;
}

int nm_bed_parse_device_indexed(nm_ctx *c, const char *path, const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset,
                                uint32_t threads, nm_bedcols **out, uint64_t stats[4]) {
    if (!c || !path || !tbi_path || !out || (n_contigs && (!names || !name_offset))) return fail(NM_EINVAL, "NULL argument");
    *out = nullptr;
    nm_bedplan *plan = nullptr;
    int rc = nm_bed_plan_indexed(path, tbi_path, n_contigs, names, name_offset, threads, &plan, stats);
    if (rc) return rc;
    if (stats) stats[3] = 0;
    rc = nm_bed_parse_device_planned(c, plan, threads, out);
    (void)nm_bedplan_close(plan);
    return rc;
}

}  // extern "C"

namespace {

int parse_device_impl(nm_ctx *c, const char *path, TextSource &src, uint32_t threads, nm_bedcols **out, bool *row_error) {
    bool row_error_unused = false;
    if (!row_error) row_error = &row_error_unused;
    *row_error = false;
    auto fail_row = [&](int code) { *row_error = true; return code; };
    if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    HIP_TRY(hipSetDevice(c->device));
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const uint64_t n = src.n;
    auto read_at = [&](uint64_t off, void *dst, uint64_t len) -> bool { return src.read_at(off, dst, len); };
    nm_bedcols *b = new (std::nothrow) nm_bedcols();
    if (!b) return fail(NM_ENOMEM, "out of host memory");
    b->ctx = c;
    struct Fail { nm_bedcols *b; bool keep = false; ~Fail() { if (!keep) (void)nm_bedcols_close(b); } } guard{b};
    // bgzip: the blocks are inflated ON THE DEVICE (bed_inflate_kernel) unless NM_BED_HOST_INFLATE=1 asks for the copy threads
    const bool dev_inflate = src.bgzf && getenv("NM_BED_HOST_INFLATE") == nullptr;
    // device inflate: slabs of whole BLOCKS (up to 3 GiB of text: line offsets are 32-bit), a line that straddles two slabs is
    // carried over on the device
    constexpr uint64_t INF_SLAB_TEXT = 3ull << 30, CARRY_CAP = 1ull << 20;      // 3 GiB of text per slab (49 000 blocks = 768 workgroups), two slabs in flight — and inflating side by side
    // The compressed bytes of a slab travel as CHUNKS: byte ranges of the file (the few bytes of gzip header / trailer between two
    // blocks ride along), copied into the pinned buffers in a few large pieces — no per-block work on the host.
    struct InfChunk { size_t first, last; uint64_t file_lo, file_hi, dev_off; };
    struct InfSlab { size_t first, last; uint64_t text, comp; std::vector<InfChunk> chunks; };
    std::vector<InfSlab> inf_slabs;
    uint64_t inf_text_cap = 0, inf_comp_cap = 0;
    if (dev_inflate) {
        // (line starts are 32-bit offsets into a slab's text: a slab stays below 4 GiB whatever the environment asks for)
        const uint64_t cap = std::min<uint64_t>(0xE0000000ull, getenv("NM_BED_INFLATE_SLAB") ? std::max<uint64_t>(1u << 16, strtoull(getenv("NM_BED_INFLATE_SLAB"), nullptr, 10)) : INF_SLAB_TEXT);
        for (size_t i = 0; i < src.pieces.size();) {
            InfSlab sl{i, i, 0, 0};
            while (sl.last < src.pieces.size() && (sl.last == sl.first || sl.text + src.pieces[sl.last].take <= cap)) {
                sl.text += src.pieces[sl.last].take;
                sl.last += 1;
            }
            sl.comp = 0;
            for (size_t a = sl.first; a < sl.last;) {            // a chunk ends at SLAB_BYTES of file, or where the file skips more than 64 KiB
                size_t e = a;
                const uint64_t lo = src.pieces[a].in_off;
                uint64_t hi = lo;
                while (e < sl.last && (e == a || (src.pieces[e].in_off <= hi + (1u << 16) && src.pieces[e].in_off + src.pieces[e].in_len - lo <= SLAB_BYTES))) {
                    hi = src.pieces[e].in_off + src.pieces[e].in_len;
                    ++e;
                }
                sl.chunks.push_back(InfChunk{a, e, lo, hi, sl.comp});
                sl.comp += (hi - lo + 15) & ~15ull;
                a = e;
            }
            inf_text_cap = std::max(inf_text_cap, sl.text);
            inf_comp_cap = std::max(inf_comp_cap, sl.comp);
            inf_slabs.push_back(sl);
            i = sl.last;
        }
    }
    // slabs of whole lines (text read or inflated by the host)
    std::vector<uint64_t> cut(1, 0);
    while (!dev_inflate && cut.back() < n) {
        uint64_t e = std::min<uint64_t>(n, cut.back() + SLAB_BYTES);
        if (e < n) {                                       // back to the end of the last whole line (the tail of the slab is read in pieces)
            const uint64_t lo = cut.back();
            std::vector<uint8_t> win(1u << 16);
            bool found = false;
            while (e > lo && !found) {
                const uint64_t w0 = e - lo > win.size() ? e - win.size() : lo;
                if (!read_at(w0, win.data(), e - w0)) return fail(NM_EINVAL, "cannot read pileup '%s'", path);
                uint64_t k = e;
                while (k > w0 && win[k - 1 - w0] != '\n') --k;
                if (k > w0) { e = k; found = true; }
                else e = w0;
            }
            if (!found) return fail_row(fail(NM_EINVAL, "%s: a line longer than %llu bytes", path, (unsigned long long)SLAB_BYTES));
        }
        cut.push_back(e);
    }
    const size_t n_slabs = cut.size() - 1;
    // pinned ring + producer thread: file -> pinned, several copy threads per slab
    constexpr int RING = 3;
    uint8_t *h_ring[RING] = {nullptr, nullptr, nullptr};
    uint8_t *d_slab[2] = {nullptr, nullptr};
    uint32_t *d_line_start = nullptr, *d_block_cnt = nullptr, *d_block_off = nullptr;
    unsigned long long *d_first_error = nullptr, *d_run_row = nullptr, *d_prev = nullptr;
    uint64_t *d_hash = nullptr;
    BedRun *d_runs = nullptr;
    constexpr uint32_t RUN_CAP = 1u << 22;
    unsigned int *d_counters = nullptr;           // [0] patches, [1] runs
    uint4 *d_patch = nullptr;
    void *d_scan_tmp = nullptr;
    // the ctx's copy stream (idle during this call; a stream of its own only if the ctx has none: a new stream costs 10 - 15 ms, nmres.h)
    hipStream_t copy_stream = getenv("NM_OWN_COPY_STREAM") ? nullptr : c->copy_stream;        // (A/B: a stream of its own, as before round 6)
    const bool own_copy_stream = copy_stream == nullptr;
    hipEvent_t h2d_done[RING] = {nullptr, nullptr, nullptr}, parsed[2] = {nullptr, nullptr};
    std::vector<void *> dev_tmp;
    struct Cleanup {
        uint8_t **h; uint8_t **d; std::vector<void *> &tmp; hipStream_t &cs; bool own; hipEvent_t *e1; hipEvent_t *e2; nm_ctx *c;
        ~Cleanup() {
            (void)hipStreamSynchronize(c->stream);
            if (cs) (void)hipStreamSynchronize(cs);
            for (int i = 0; i < RING; ++i) { nmres::pinned_give(h[i]); if (e1[i]) (void)hipEventDestroy(e1[i]); }      // (kept for the next parser: nmres.h)
            for (int i = 0; i < 2; ++i) { if (d[i]) (void)dev_free(d[i]); if (e2[i]) (void)hipEventDestroy(e2[i]); }
            for (void *p : tmp) (void)dev_free(p);
            if (cs && own) (void)hipStreamDestroy(cs);
        }
    } cleanup{h_ring, d_slab, dev_tmp, copy_stream, own_copy_stream, h2d_done, parsed, c};
    const uint64_t slab_cap = dev_inflate ? inf_text_cap + CARRY_CAP + 64 : std::min<uint64_t>(SLAB_BYTES, std::max<uint64_t>(n, 1));
    const uint32_t max_blocks = (uint32_t)((slab_cap + BLOCK_BYTES - 1) / BLOCK_BYTES);
    const uint64_t max_lines = slab_cap / 2 + 1;                          // a non-empty line and its '\n'
    if (n_slabs) {
        for (int i = 0; i < RING && (size_t)i < n_slabs; ++i) HIP_TRY(nmres::pinned_take((void **)&h_ring[i], slab_cap));
        for (int i = 0; i < 2 && (size_t)i < n_slabs; ++i) HIP_TRY(dev_malloc(&d_slab[i], slab_cap + 128));
        for (int i = 0; i < RING; ++i) HIP_TRY(hipEventCreateWithFlags(&h2d_done[i], hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&parsed[i], hipEventDisableTiming));
        if (!copy_stream) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    }
    const bool alloc_timing = getenv("NM_BED_TIMING") != nullptr;
    auto tmp_alloc = [&](void **p, size_t bytes) -> hipError_t {
        const double t0 = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        const hipError_t e = device_alloc(p, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) dev_tmp.push_back(*p);
        if (alloc_timing) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
            if (dt > 0.003) fprintf(stderr, "[bed] an allocation of %.3f GB took %.3f s\n", (double)bytes / 1e9, dt);
        }
        return e;
    };
    // (line starts of a slab: worst case one per two bytes is absurd for a pileup; sized for lines of >= 16 bytes, checked)
    const uint64_t line_cap = std::min<uint64_t>(max_lines, slab_cap / 16 + 1024);
    // the two per-line arrays: sized for lines of >= 16 bytes when a slab is 32 MB; a multi-GiB slab of the device-inflate path
    // gets them once its line count is known (12 bytes per line: a 3 GiB slab of 75-byte lines needs 0.5 GB, not 2.4)
    uint64_t line_have = dev_inflate ? 0 : line_cap;
    if (line_have) HIP_TRY(tmp_alloc((void **)&d_line_start, line_have * 4));
    HIP_TRY(tmp_alloc((void **)&d_block_cnt, ((size_t)max_blocks + 1) * 4));
    HIP_TRY(tmp_alloc((void **)&d_block_off, ((size_t)max_blocks + 1) * 4));
    HIP_TRY(tmp_alloc((void **)&d_first_error, 8));
    HIP_TRY(tmp_alloc((void **)&d_counters, 8));
    HIP_TRY(tmp_alloc((void **)&d_patch, (size_t)PATCH_CAP * sizeof(uint4)));
    if (line_have) HIP_TRY(tmp_alloc((void **)&d_hash, line_have * 8));
    HIP_TRY(tmp_alloc((void **)&d_runs, (size_t)RUN_CAP * sizeof(BedRun)));
    constexpr uint32_t RUN_NAME_CAP = 1u << 20;           // runs whose names come back from the device (64 MB); further ones are read from the file
    uint8_t *d_run_names = nullptr;
    HIP_TRY(tmp_alloc((void **)&d_run_names, (size_t)RUN_NAME_CAP * RUN_NAME_SLOT));
    HIP_TRY(tmp_alloc((void **)&d_prev, 16));
    size_t scan_bytes = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_block_cnt, d_block_off, 0u, (size_t)max_blocks + 1, rocprim::plus<unsigned int>(), c->stream));
    HIP_TRY(tmp_alloc(&d_scan_tmp, scan_bytes));
    HIP_TRY(hipMemsetAsync(d_first_error, 0xFF, 8, c->stream));
    HIP_TRY(hipMemsetAsync(d_counters, 0, 8, c->stream));

    std::atomic<bool> read_failed{false};
    std::mutex mu;
    std::condition_variable cv;
    size_t filled = 0, consumed = 0;              // slabs copied into the ring / slabs whose H2D has been waited for
    bool stop = false;
    double t_read = 0;
    // The copy threads live for the whole file (a fresh set per slab was 15 thread starts for every 32 MB: ~90 ms of a
    // 7.8 GB parse): each takes its share of every slab in turn; a slab is full when its last share has arrived, and since
    // every thread walks the slabs in order they fill in order.
    const unsigned nt = std::max(1u, threads - 1);
    std::vector<unsigned> shares_done(n_slabs, 0);
    std::vector<std::thread> producers;
    for (unsigned t = 0; t < nt; ++t)
        producers.emplace_back([&, t] {
            std::vector<char> inflate_tmp;
            for (size_t k = 0; k < n_slabs; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < consumed + RING; });
                    if (stop) return;
                }
                const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                const uint64_t lo = cut[k], len = cut[k + 1] - cut[k];
                if (!src.fill(lo, len, t, nt, h_ring[k % RING], inflate_tmp)) read_failed = true;
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (t == 0) t_read += dt;
                    if (++shares_done[k] == nt) filled = k + 1;
                }
                cv.notify_all();
            }
        });
    struct Join {
        std::vector<std::thread> &ts; std::mutex &mu; std::condition_variable &cv; bool &stop;
        ~Join() {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            for (auto &t : ts) if (t.joinable()) t.join();
        }
    } join{producers, mu, cv, stop};

    bool first_rows = true;                 // the next slab with rows holds row 0
    size_t n_parsed = 0;                    // slabs with rows so far (ping-pong of the "hash of the row before")
    // the whole lines of text[0, len) (16-byte aligned, device memory) -> rows; text_base: the text offset of its first byte;
    // grow_hint: file size / slab size when this is the first of several slabs (sizes the columns once); counted(): called
    // as soon as the line count is back on the host
    auto parse_text = [&](const uint8_t *d_text, uint64_t len, uint64_t text_base, double grow_hint, const std::function<void()> &counted) -> int {
        const uint32_t nblk = (uint32_t)((len + BLOCK_BYTES - 1) / BLOCK_BYTES);
        auto clock_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tp0 = alloc_timing ? clock_s() : 0.0;
        hipLaunchKernelGGL(bed_count_kernel, dim3(nblk), dim3(256), 0, c->stream, d_text, len, d_block_cnt);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(d_block_cnt + nblk, 0, 4, c->stream));
        HIP_TRY(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_block_cnt, d_block_off, 0u, (size_t)nblk + 1, rocprim::plus<unsigned int>(), c->stream));
        uint32_t n_lines = 0;
        HIP_TRY(hipMemcpyAsync(&n_lines, d_block_off + nblk, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double tp1 = alloc_timing ? clock_s() : 0.0;
        counted();
        if (n_lines > line_cap) return fail_row(fail(NM_EINVAL, "%s: lines shorter than 16 bytes are no bedMethyl rows", path));
        if (n_lines > line_have) {                                    // (device-inflate slabs: the arrays follow the line count)
            line_have = std::min<uint64_t>(line_cap, (uint64_t)((double)n_lines * 1.1) + 1024);
            HIP_TRY(tmp_alloc((void **)&d_line_start, line_have * 4));            // (the smaller ones stay in dev_tmp until the call ends)
            HIP_TRY(tmp_alloc((void **)&d_hash, line_have * 8));
        }
        if (b->n_rows + n_lines >= 0xFFFFFFFFull) return fail(NM_ERANGE, "%s: more than 4G rows in one pileup", path);
        // the columns are sized ONCE, from the line density of the first slab (+ 3 % and one slab's worth of slack): every
        // regrow is a fresh allocation (scrubbed by the driver when the memory was used before) plus a copy
        uint64_t want = b->n_rows + n_lines;
        if (grow_hint > 0) want = std::max<uint64_t>(want, (uint64_t)((double)n_lines * grow_hint * 1.03) + n_lines);
        const double tp2 = alloc_timing ? clock_s() : 0.0;
        int rc = grow(b, want, c->stream);
        if (rc) return rc;
        if (alloc_timing && n_parsed == 0)
            fprintf(stderr, "[bed] first slab's parse: line count known after %.3f s, line arrays %.3f s, columns for %llu rows %.3f s\n", tp1 - tp0, tp2 - tp1,
                    (unsigned long long)want, clock_s() - tp2);
        if (n_lines) {
            hipLaunchKernelGGL(bed_starts_kernel, dim3(nblk), dim3(256), 0, c->stream, d_text, len, d_block_off, d_line_start);
            BedOut o{d_hash, b->d_position, b->d_mod, b->d_strand, b->d_frac, b->d_nvalid, d_first_error, d_counters, d_patch};
            hipLaunchKernelGGL(bed_parse_kernel, dim3((n_lines + 255) / 256), dim3(256), 0, c->stream, d_text, len, d_line_start, n_lines, b->n_rows,
                               text_base, o);
            hipLaunchKernelGGL(bed_runs_kernel, dim3((n_lines + 255) / 256), dim3(256), 0, c->stream, d_hash, d_line_start, n_lines, b->n_rows, text_base,
                               first_rows ? 1 : 0, d_prev + (n_parsed & 1), d_prev + ((n_parsed + 1) & 1), d_counters + 1, d_runs, RUN_CAP,
                               d_text, len, d_run_names, RUN_NAME_CAP);
            HIP_TRY(hipGetLastError());
            first_rows = false;
            n_parsed += 1;
        }
        b->n_rows += n_lines;
        return NM_OK;
    };
    // ---- bgzip, inflated on the device: per slab the compressed bytes of its blocks (packed, through the pinned ring) ->
    // bed_inflate_kernel -> the same line / field kernels
    if (dev_inflate && !inf_slabs.empty()) {
        // A PIPELINE over the slabs (round 5; the stages ran one after the other before: 0.098 s per 3 GiB slab, of which the inflate
        // kernel was 0.065): (1) a slab's compressed bytes -> pinned chunks -> device (copy stream); (2) its inflate streams: phase 1 into
        // the slab's token buffer, then phase 2 + CRC-32 + end of the last whole line into one of TWO text buffers; (3) ctx stream: the
        // line / field kernels.  A slab stays below 4 GiB of text (line starts are 32-bit offsets): 2 x 3 GiB of text buffers (slabs of
        // 1.5 GiB were 6 % slower — a slab should fill every lane slot phase 1's LDS tables leave: 768 workgroups, three per CU).
        constexpr uint64_t CHUNK = SLAB_BYTES;                        // compressed bytes per pinned buffer
        uint8_t *d_text[2] = {nullptr, nullptr};
        // Round 6: the blocks are inflated in TWO PHASES (nmbedinflate.h: one lane per block decodes the Huffman stream into tokens, one
        // wave per block turns the tokens into text) unless NM_BED_INFLATE_V1=1 asks for the single kernel of rounds 4 - 5 (A/B).  A
        // block's token region is `token_fraction` of its text (NM_BED_TOKEN_FRACTION; tests shrink it to send blocks down the fallback)
        const bool two_phase = getenv("NM_BED_INFLATE_V1") == nullptr;
        const double token_fraction = getenv("NM_BED_TOKEN_FRACTION") ? atof(getenv("NM_BED_TOKEN_FRACTION")) : 0.625;
        size_t max_tok = 0;
        unsigned int *d_status = nullptr;
        unsigned long long *d_tail = nullptr;                         // [2]
        size_t max_pieces = 0, max_partial = 0;
        for (const InfSlab &sl : inf_slabs) {
            max_pieces = std::max(max_pieces, sl.last - sl.first);
            size_t np = 0;
            for (size_t i = sl.first; i < sl.last; ++i) np += src.pieces[i].skip != 0 || src.pieces[i].take != src.pieces[i].out_len;
            max_partial = std::max(max_partial, np);
            size_t tk = 0;
            for (size_t i = sl.first; i < sl.last; ++i) tk += inf2_region_bytes((unsigned int)src.pieces[i].out_len, token_fraction);
            max_tok = std::max(max_tok, tk);
        }
        // Round 6: the pipeline is one stage deeper.  What a slab needs BEFORE its text can be written — its compressed bytes on the device,
        // its piece table, its tokens (phase 1) — exists three times (n_cmp) and is produced by a STAGING THREAD that runs up to two slabs
        // ahead of the slab being parsed; what needs the text buffer (phase 2 / the single kernel, CRC-32, the end of the last line) is
        // queued by the calling thread once the parse of the slab that used the buffer before has been queued.  Before, the calling thread
        // copied slab k+1 (12 ms of memcpy on 15 threads), THEN waited for slab k's inflate, THEN queued its parse: phase 1 of slab k+1
        // could not start before phase 2 of slab k - 1 ... had been waited for, and the two phases of neighbouring slabs never overlapped.
        constexpr int NC = 3;
        const size_t n_inf_slabs = inf_slabs.size();
        const int n_txt = n_inf_slabs > 1 ? 2 : 1, n_cmp = (int)std::min<size_t>(NC, n_inf_slabs);
        uint8_t *d_cmp_comp[NC] = {nullptr, nullptr, nullptr}, *d_cmp_scratch[NC] = {nullptr, nullptr, nullptr}, *d_cmp_tok[NC] = {nullptr, nullptr, nullptr};
        InfPiece *d_cmp_pieces[NC] = {nullptr, nullptr, nullptr};
        InfTokMeta *d_cmp_meta[NC] = {nullptr, nullptr, nullptr};
        const bool timing = getenv("NM_BED_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_alloc0 = now();
        for (int b = 0; b < n_txt; ++b) HIP_TRY(tmp_alloc((void **)&d_text[b], slab_cap + 128));
        if (n_txt == 1) d_text[1] = d_text[0];
        const double t_alloc_text = now();
        for (int b = 0; b < n_cmp; ++b) {
            HIP_TRY(tmp_alloc((void **)&d_cmp_comp[b], inf_comp_cap + INF_OVERRUN));     // (what a lane can read past a damaged stream before it notices)
            HIP_TRY(tmp_alloc((void **)&d_cmp_scratch[b], std::max<size_t>(max_partial, 1) << 16));
            HIP_TRY(tmp_alloc((void **)&d_cmp_pieces[b], max_pieces * sizeof(InfPiece)));
            if (two_phase) {
                HIP_TRY(tmp_alloc((void **)&d_cmp_tok[b], max_tok + 64));
                HIP_TRY(tmp_alloc((void **)&d_cmp_meta[b], max_pieces * sizeof(InfTokMeta)));
            }
        }
        HIP_TRY(tmp_alloc((void **)&d_status, 4));
        HIP_TRY(tmp_alloc((void **)&d_tail, 8 * NC));
        HIP_TRY(hipMemsetAsync(d_status, 0, 4, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double t_alloc_sets = now();
        uint8_t *h_chunk[2] = {nullptr, nullptr};
        hipEvent_t chunk_done[2] = {nullptr, nullptr}, parsed_ev[2] = {nullptr, nullptr}, inflated[NC] = {nullptr, nullptr, nullptr};
        // One inflate stream per set: a slab's phase 1, phase 2, CRC-32 and tail kernels follow each other on its own stream; the kernels of
        // neighbouring slabs run side by side as far as the device has room (three phase-1 workgroups per CU leave LDS for six phase-2 waves)
        hipStream_t inf_streams[NC] = {nullptr, nullptr, nullptr};
        int n_inf = 1, inf_priority = 0;
        bool inf_with_priority = false;
        struct Pinned { uint8_t **h; hipEvent_t *e, *e2, *e3; hipStream_t &cs; hipStream_t *is; ~Pinned() {
            if (cs) (void)hipStreamSynchronize(cs);
            for (int i = 0; i < NC; ++i) if (is[i]) (void)hipStreamSynchronize(is[i]);
            for (int i = 0; i < 2; ++i) {
                nmres::pinned_give(h[i]);
                if (e[i]) (void)hipEventDestroy(e[i]);
                if (e3[i]) (void)hipEventDestroy(e3[i]);
            }
            for (int i = 0; i < NC; ++i) if (e2[i]) (void)hipEventDestroy(e2[i]);
            for (int i = 0; i < NC; ++i) if (is[i]) (void)hipStreamDestroy(is[i]);
        } } pinned{h_chunk, chunk_done, inflated, parsed_ev, copy_stream, inf_streams};
        if (!copy_stream) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        {
            // The inflate streams get the LOWEST priority, which also gives them hardware queues of their own: the runtime deals streams of
            // one priority round-robin onto 4 hardware queues, and when an inflate stream landed on the copy stream's queue the next slab's
            // host-to-device copies only ran between two inflate kernels (measured, tools/leases/r5/gpu_r5k.sh: 0.047 s per slab against
            // 0.034 s with GPU_MAX_HW_QUEUES=8 — per slab inflate + copy instead of their maximum).  Long-running, latency-bound kernels
            // are the right thing to give way to copies and parse kernels in any case.
            int least = 0, greatest = 0;
            const bool prio = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest && getenv("NM_BED_FLAT_PRIORITY") == nullptr;
            n_inf = getenv("NM_BED_ONE_INFLATE_STREAM") ? 1 : n_cmp;        // (A/B: the slabs' kernels one after the other)
            inf_priority = prio ? least : 0;
            inf_with_priority = prio;
        }
        // A new stream costs 9 - 15 ms (tools/alloc_costs_probe.hip): the first set's is made here, the others by the staging thread when it
        // first gets to their set — while the first slab's phase 1 runs (NM_BED_EAGER_STREAMS=1: all of them here, as before)
        auto make_inf_stream = [&](int i) -> int {
            if (inf_streams[i] || i >= n_inf) return NM_OK;
            if (inf_with_priority) HIP_TRY(hipStreamCreateWithPriority(&inf_streams[i], hipStreamNonBlocking, inf_priority));
            else HIP_TRY(hipStreamCreateWithFlags(&inf_streams[i], hipStreamNonBlocking));
            return NM_OK;
        };
        for (int i = 0; i < (getenv("NM_BED_EAGER_STREAMS") ? n_inf : 1); ++i) {
            const int rcs = make_inf_stream(i);
            if (rcs) return rcs;
        }
        auto inf_stream_of = [&](int set) { return inf_streams[set] ? inf_streams[set] : inf_streams[0]; };
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(nmres::pinned_take((void **)&h_chunk[i], std::min<uint64_t>(CHUNK, inf_comp_cap) + (1u << 16)));       // (a chunk is at most SLAB_BYTES = CHUNK of file)
            HIP_TRY(hipEventCreateWithFlags(&chunk_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&parsed_ev[i], hipEventDisableTiming));
        }
        for (int i = 0; i < NC; ++i) HIP_TRY(hipEventCreateWithFlags(&inflated[i], hipEventDisableTiming));
        if (timing) {
            (void)hipStreamSynchronize(c->stream);
            fprintf(stderr, "[bed] device inflate: %zu slabs, buffers allocated %.3f s after entry (%.3f s before the first allocation; %d text buffers of %.2f GB %.3f s, "
                            "%d sets of %.2f GB compressed + %.2f GB tokens %.3f s, streams + pinned chunks %.3f s)\n", inf_slabs.size(), now() - t_begin, t_alloc0 - t_begin,
                    n_txt, (double)slab_cap / 1e9, t_alloc_text - t_alloc0, n_cmp, (double)inf_comp_cap / 1e9, (double)max_tok / 1e9, t_alloc_sets - t_alloc_text, now() - t_alloc_sets);
        }
        std::vector<InfPiece> hp[NC];
        const CrcConsts crc_k = crc_consts();
#ifdef NM_BED_PROBES                                                    // (timing probe builds only: what the check costs)
        const bool check_crc = getenv("NM_BED_NO_CRC") == nullptr;
#else
        constexpr bool check_crc = true;                                // the shipped library always checks every member's CRC-32
#endif
        const unsigned nt = std::max(1u, threads - 1);
        // where a slab's last whole line ends and the inflate status come back through PINNED words written by a kernel on the inflate
        // stream: a hipMemcpyAsync there would sit in a copy-engine queue behind the inflate kernel it waits for — and in front of the
        // next slab's compressed bytes on the copy stream, which then travel only after the inflate (measured: 0.036 s of 0.042 per slab)
        unsigned long long *h_words = nullptr;                         // [NC] end of lines, [NC] status
        HIP_TRY(hipHostMalloc((void **)&h_words, 64, hipHostMallocDefault));
        struct FreeWords { unsigned long long *p; hipStream_t *is; ~FreeWords() { for (int i = 0; i < NC; ++i) if (is[i]) (void)hipStreamSynchronize(is[i]); (void)hipHostFree(p); } } free_words{h_words, inf_streams};
        volatile unsigned long long *end_of_lines_h = h_words, *status_h = h_words + NC;
        double t_copy_slab[NC] = {0, 0, 0};
        // ---- the front half of slab si (STAGING THREAD): piece table and compressed bytes to the device, phase 1 queued on the slab's stream
        size_t n_chunk = 0;
        double t_read_staging = 0;
        auto stage_front = [&](size_t si) -> int {
            const InfSlab &sl = inf_slabs[si];
            const int cs = (int)(si % (size_t)n_cmp);
            {
                const int rcs = make_inf_stream(cs);                     // (the set's first slab: its stream is made now)
                if (rcs) return rcs;
            }
            const hipStream_t inf_stream = inf_stream_of(cs);
            // piece table of the slab: packed compressed offsets, where the text goes (behind the carry area)
            std::vector<InfPiece> &pieces = hp[cs];
            pieces.clear();
            uint64_t toff = CARRY_CAP, poff = 0, tok_off = 0;
            for (const InfChunk &ch : sl.chunks)
                for (size_t i = ch.first; i < ch.last; ++i) {
                    const nmbgzf::Piece &pp = src.pieces[i];
                    const bool partial = pp.skip != 0 || pp.take != pp.out_len;
                    const unsigned int tok_len = inf2_region_bytes((unsigned int)pp.out_len, token_fraction);
                    pieces.push_back({ch.dev_off + (pp.in_off - ch.file_lo), (unsigned int)pp.in_len, (unsigned int)pp.out_len, toff, pp.skip, pp.take, poff, pp.crc, tok_len, tok_off});
                    toff += pp.take;
                    tok_off += tok_len;
                    if (partial) poff += 1u << 16;
                }
            // (this set's buffers were last read by the kernels of slab si - n_cmp: the calling thread has waited for them)
            HIP_TRY(hipMemcpyAsync(d_cmp_pieces[cs], pieces.data(), pieces.size() * sizeof(InfPiece), hipMemcpyHostToDevice, copy_stream));
            // compressed bytes: chunks of whole pieces through two pinned buffers, memcpy on several threads
            const double t0 = now();
            for (const InfChunk &ch : sl.chunks) {
                const uint64_t bytes = ch.file_hi - ch.file_lo;
                uint8_t *dst = h_chunk[n_chunk % 2];
                if (n_chunk >= 2) HIP_TRY(hipEventSynchronize(chunk_done[n_chunk % 2]));
                std::vector<std::thread> pool;
                // (out of the MAPPING: its pages are in the page tables since the walk over the blocks — the kernel maps 64 KiB around
                //  every fault — and 15 threads copy 50 GB/s; preads of the same ranges were measured at 6.5 GB/s)
                for (unsigned t = 0; t < nt; ++t)
                    pool.emplace_back([&, t] {
                        const uint64_t a = bytes * t / nt, e = bytes * (t + 1) / nt;
                        if (e > a) memcpy(dst + a, src.z + ch.file_lo + a, (size_t)(e - a));
                        // the pages just copied leave the page tables HERE, a few thousand at a time on the copying threads (read lock
                        // of the address space): unmapping the whole file at the end — 3.5 million entries at 1 Gbp — held the
                        // address-space lock for 0.2 s, and whoever allocated or faulted meanwhile (the pre-filters' scratch) waited
                        const uintptr_t lo = ((uintptr_t)(src.z + ch.file_lo + a) + 4095u) & ~(uintptr_t)4095u, hi = (uintptr_t)(src.z + ch.file_lo + e) & ~(uintptr_t)4095u;
                        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_DONTNEED);
                    });
                for (auto &th : pool) th.join();
                HIP_TRY(hipMemcpyAsync(d_cmp_comp[cs] + ch.dev_off, dst, bytes, hipMemcpyHostToDevice, copy_stream));
                HIP_TRY(hipEventRecord(chunk_done[n_chunk % 2], copy_stream));
                n_chunk += 1;
            }
            t_copy_slab[cs] = now() - t0;
            t_read_staging += t_copy_slab[cs];
            HIP_TRY(hipStreamWaitEvent(inf_stream, chunk_done[(n_chunk - 1) % 2], 0));
            if (two_phase) {
                hipLaunchKernelGGL(bed_tokens_kernel, dim3((unsigned)((pieces.size() + INF_LANES - 1) / INF_LANES)), dim3(INF_LANES), 0, inf_stream, d_cmp_comp[cs], d_cmp_pieces[cs],
                                   (unsigned int)pieces.size(), d_cmp_tok[cs], d_cmp_meta[cs], d_status);
                HIP_TRY(hipGetLastError());
            }
            return NM_OK;
        };
        // ---- the back half of slab si (calling thread, after the parse of slab si - n_txt has been queued): everything that writes or
        // reads the slab's text buffer
        auto stage_back = [&](size_t si) -> int {
            const InfSlab &sl = inf_slabs[si];
            const int cs = (int)(si % (size_t)n_cmp), tb = (int)(si % (size_t)n_txt);
            const hipStream_t inf_stream = inf_stream_of(cs);
            uint8_t *text = d_text[tb];
            const unsigned int n_pieces = (unsigned int)hp[cs].size();
            if (si >= (size_t)n_txt) HIP_TRY(hipStreamWaitEvent(inf_stream, parsed_ev[tb], 0));      // the text buffer has been parsed
#ifdef NM_BED_PROBES
            {
                const int probe = getenv("NM_BED_INFLATE_PROBE") ? atoi(getenv("NM_BED_INFLATE_PROBE")) : 0;
                HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_inf_probe), &probe, sizeof probe));
            }
#endif
            // NM_BED_INFLATE_LDS_PAD (bytes of unused dynamic LDS per workgroup; A/B of the single kernel): fewer workgroups per CU
            static const unsigned lds_pad = getenv("NM_BED_INFLATE_LDS_PAD") ? (unsigned)atoi(getenv("NM_BED_INFLATE_LDS_PAD")) : 0u;
            const dim3 lane_grid((n_pieces + INF_LANES - 1) / INF_LANES);
            if (two_phase) {
                // (the blocks whose tokens did not fit their region, if any: the lanes of all others leave at once)
                hipLaunchKernelGGL(bed_inflate_kernel, lane_grid, dim3(INF_LANES), 0, inf_stream, d_cmp_comp[cs], d_cmp_pieces[cs], n_pieces, text, d_cmp_scratch[cs], d_status,
                                   (const InfTokMeta *)d_cmp_meta[cs]);
                HIP_TRY(hipGetLastError());
                hipLaunchKernelGGL(bed_resolve_kernel, dim3(n_pieces), dim3(64), 0, inf_stream, d_cmp_pieces[cs], n_pieces, d_cmp_tok[cs], d_cmp_meta[cs], text,
                                   d_cmp_scratch[cs], d_status);
            } else {
                hipLaunchKernelGGL(bed_inflate_kernel, lane_grid, dim3(INF_LANES), lds_pad, inf_stream, d_cmp_comp[cs], d_cmp_pieces[cs], n_pieces, text, d_cmp_scratch[cs], d_status,
                                   (const InfTokMeta *)nullptr);
            }
            HIP_TRY(hipGetLastError());
            if (check_crc) {
                hipLaunchKernelGGL(bed_crc_kernel, dim3((n_pieces + 3) / 4), dim3(256), 0, inf_stream, d_cmp_pieces[cs], n_pieces, text, d_cmp_scratch[cs], crc_k, d_status);
                HIP_TRY(hipGetLastError());
            }
            // where the last whole line ends; what follows it is carried into the next slab
            const uint64_t total = CARRY_CAP + sl.text;
            const bool want_tail = si + 1 != inf_slabs.size();
            if (want_tail) {
                hipLaunchKernelGGL(bed_tail_kernel, dim3(1), dim3(256), 0, inf_stream, text, total, CARRY_CAP, d_tail + cs);
                HIP_TRY(hipGetLastError());
            }
            hipLaunchKernelGGL(bed_report_kernel, dim3(1), dim3(1), 0, inf_stream, want_tail ? d_tail + cs : nullptr, (unsigned long long)total, d_status,
                               h_words + cs, h_words + NC + cs);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(inflated[cs], inf_stream));
            return NM_OK;
        };
        // ---- the staging thread: front halves in slab order, at most n_cmp slabs ahead of the last slab whose inflate has been waited for
        std::mutex pmu;
        std::condition_variable pcv;
        size_t fronts_done = 0, retired = 0;
        int front_rc = NM_OK;
        std::string front_error;
        bool stop_staging = false;
        std::thread stager([&] {
            (void)hipSetDevice(c->device);
            for (size_t si = 0; si < n_inf_slabs; ++si) {
                {
                    std::unique_lock<std::mutex> lk(pmu);
                    pcv.wait(lk, [&] { return stop_staging || si < retired + (size_t)n_cmp; });
                    if (stop_staging) return;
                }
                const int rc = stage_front(si);
                {
                    std::lock_guard<std::mutex> lk(pmu);
                    if (rc) { front_rc = rc; front_error = nm_last_error(); }
                    fronts_done = si + 1;
                }
                pcv.notify_all();
                if (rc) return;
            }
        });
        struct JoinStager {                                             // (declared last: leaves first, before the buffers and streams above it go)
            std::thread &t; std::mutex &m; std::condition_variable &cv; bool &stop;
            ~JoinStager() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); }
        } join_stager{stager, pmu, pcv, stop_staging};
        auto wait_front = [&](size_t si) -> int {
            std::unique_lock<std::mutex> lk(pmu);
            pcv.wait(lk, [&] { return fronts_done > si || front_rc != NM_OK; });
            if (front_rc != NM_OK) return fail(front_rc, "%s", front_error.c_str());       // (the message was made on the staging thread)
            return NM_OK;
        };
        uint64_t carry = 0;                                             // bytes of an unfinished line in front of the slab
        const double t_loop0 = now();
        for (size_t k = 0; k < std::min<size_t>(2, n_inf_slabs); ++k) {   // (both text buffers are free: the first two back halves at once)
            int rc0 = wait_front(k);
            if (rc0) return rc0;
            rc0 = stage_back(k);
            if (rc0) return rc0;
        }
        const double t_slabs0 = now();
        for (size_t si = 0; si < inf_slabs.size(); ++si) {
            const InfSlab &sl = inf_slabs[si];
            const int cs = (int)(si % (size_t)n_cmp), tb = (int)(si % (size_t)n_txt);
            uint8_t *text = d_text[tb];
            const bool last_slab = si + 1 == inf_slabs.size();
            const double t_wait = now();
            HIP_TRY(hipEventSynchronize(inflated[cs]));
            const double t_inflated = now();
            b->t_inflate += t_inflated - t_wait;
            {
                std::lock_guard<std::mutex> lk(pmu);                    // the set of slab si is free for slab si + n_cmp
                retired = si + 1;
            }
            pcv.notify_all();
#ifdef NM_BED_PROBES
            if (getenv("NM_BED_INFLATE_PROBE") && atoi(getenv("NM_BED_INFLATE_PROBE"))) {
                fprintf(stderr, "[bed] PROBE %s: slab %zu (%zu blocks, %.2f GB text): waited %.3f s for the inflate\n", getenv("NM_BED_INFLATE_PROBE"), si,
                        hp[cs].size(), sl.text / 1e9, t_inflated - t_wait);
                if (last_slab || si == 3) return fail(NM_EINVAL, "inflate probe: the text is garbage on purpose");
                HIP_TRY(hipEventRecord(parsed_ev[tb], c->stream));
                if (si + 2 < inf_slabs.size()) { int r2 = wait_front(si + 2); if (r2) return r2; r2 = stage_back(si + 2); if (r2) return r2; }
                continue;
            }
#endif
            const unsigned int status = (unsigned int)status_h[cs];
            const unsigned long long end_of_lines = end_of_lines_h[cs];
            const uint64_t begin = CARRY_CAP - carry, total = CARRY_CAP + sl.text;
            if (status) return fail(NM_EINVAL, "%s: corrupt BGZF block (block %u of the slab, %s %u)", path, status >> 8, (status & 255u) == 19u ? "CRC-32 mismatch, code" : "inflate error", status & 255u);
            if (!last_slab && (end_of_lines <= begin || total - end_of_lines > CARRY_CAP - 16))
                return fail_row(fail(NM_EINVAL, "%s: a line longer than %llu bytes", path, (unsigned long long)(CARRY_CAP - 16)));
            // the parse kernels want a 16-byte aligned start: the few bytes in front of the carried line become empty lines
            const uint64_t aligned = begin & ~15ull;
            if (aligned < begin) HIP_TRY(hipMemsetAsync(text + aligned, '\n', begin - aligned, c->stream));
            const uint64_t text_base = src.pieces[sl.first].text_off - carry - (begin - aligned);
            int rc = parse_text(text + aligned, end_of_lines - aligned, text_base, si == 0 && inf_slabs.size() > 1 ? (double)n / (double)sl.text : 0.0, [] {});
            if (rc) return rc;
            carry = total - end_of_lines;
            if (!last_slab && carry) HIP_TRY(hipMemcpyAsync(d_text[(si + 1) % (size_t)n_txt] + CARRY_CAP - carry, text + end_of_lines, carry, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipEventRecord(parsed_ev[tb], c->stream));
            const double t_parsed = now();
            if (si + 2 < inf_slabs.size()) {                            // this text buffer's next user: its back half behind the parse just queued
                rc = wait_front(si + 2);
                if (rc) return rc;
                rc = stage_back(si + 2);
                if (rc) return rc;
            }
            if (timing)
                fprintf(stderr, "[bed] slab %zu: %zu blocks, %.2f GB text: waited %.3f s for the inflate, parse queued in %.3f s, back half of slab %zu in %.3f s\n",
                        si, sl.last - sl.first, sl.text / 1e9, t_inflated - t_wait, t_parsed - t_inflated, si + 2, now() - t_parsed);
        }
        {
            std::lock_guard<std::mutex> lk(pmu);
            stop_staging = true;
        }
        pcv.notify_all();
        stager.join();
        t_read += t_read_staging;
        b->read_beside = true;
        const double t_loop1 = now();
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (timing)
            fprintf(stderr, "[bed] %.3f s after entry: first front half asked for; the first two back halves queued %.3f s later; all slabs queued %.3f s after that; "
                            "the last parse kernels done %.3f s later\n", t_loop0 - t_begin, t_slabs0 - t_loop0, t_loop1 - t_slabs0, now() - t_loop1);
    }
    for (size_t k = 0; k < n_slabs; ++k) {
        const uint64_t len = cut[k + 1] - cut[k];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled > k; });
        }
        if (read_failed) return fail(NM_EINVAL, src.bgzf ? "%s: corrupt BGZF block (deflate stream, size or CRC-32)" : "cannot read pileup '%s'", path);
        if (k >= 2) HIP_TRY(hipStreamWaitEvent(copy_stream, parsed[k % 2], 0));       // the device slab is free again
        HIP_TRY(hipMemcpyAsync(d_slab[k % 2], h_ring[k % RING], len, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipEventRecord(h2d_done[k % RING], copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, h2d_done[k % RING], 0));
        int rc = parse_text(d_slab[k % 2], len, cut[k], k == 0 && n_slabs > 1 ? (double)n / (double)len : 0.0, [&] {
            // the pinned buffer of this slab may be refilled once its H2D is done (it is: the count kernel ran after it)
            { std::lock_guard<std::mutex> lk(mu); consumed = k + 1; }
            cv.notify_all();
        });
        if (rc) return rc;
        HIP_TRY(hipEventRecord(parsed[k % 2], c->stream));
    }
    if (read_failed) return fail(NM_EINVAL, src.bgzf ? "%s: corrupt BGZF block (deflate stream, size or CRC-32)" : "cannot read pileup '%s'", path);
    unsigned long long first_error = ~0ull;
    HIP_TRY(hipMemcpyAsync(&first_error, d_first_error, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (first_error != ~0ull) return fail_row(fail(NM_EINVAL, "%s: %s", path, row_error_text((uint32_t)(first_error & 0xFF))));
    std::vector<char> line_buf(1u << 16);
    // k-th tab-separated field of the line at text offset `line` (buf: scratch of the calling thread).  A short window first:
    // in a bgzip file every window costs the inflation of the block(s) under it
    auto field_in = [&](std::vector<char> &buf, uint64_t line, int k, const char **fb, const char **fe) {
        if (buf.size() < (1u << 16)) buf.resize(1u << 16);
        uint64_t got = std::min<uint64_t>(1024, n - line);
        if (!read_at(line, buf.data(), got)) { *fb = *fe = buf.data(); return; }
        if (!memchr(buf.data(), '\n', (size_t)got) && got < n - line) {
            got = std::min<uint64_t>(buf.size(), n - line);
            if (!read_at(line, buf.data(), got)) { *fb = *fe = buf.data(); return; }
        }
        const char *p = buf.data(), *end = buf.data() + got;
        const char *le = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!le) le = end;
        if (le > p && le[-1] == '\r') --le;
        for (int i = 0; i < k; ++i) {
            const char *t = static_cast<const char *>(memchr(p, '\t', (size_t)(le - p)));
            p = t ? t + 1 : le;
        }
        const char *t = static_cast<const char *>(memchr(p, '\t', (size_t)(le - p)));
        *fb = p;
        *fe = t ? t : le;
    };
    auto field = [&](uint64_t line, int k, const char **fb, const char **fe) { field_in(line_buf, line, k, fb, fe); };
    // ---- runs of equal contig names -> names, ids, the contig column
    const double t_after_slabs = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (b->n_rows) {
        unsigned int n_runs = 0;
        HIP_TRY(hipMemcpyAsync(&n_runs, d_counters + 1, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (n_runs > RUN_CAP) return fail(NM_EDECLINED, "%s: more than %u runs of contig names (rows not grouped by contig): use nm_bed_open", path, RUN_CAP);
        const double tt0 = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        // (both tables through ONE pinned buffer and the ctx stream: a blocking hipMemcpy into pageable memory was 8 ms each — the runtime
        //  pins the destination on the fly)
        const unsigned int n_named = std::min<unsigned int>(n_runs, RUN_NAME_CAP);
        const size_t runs_bytes = (size_t)n_runs * sizeof(BedRun), names_bytes = (size_t)n_named * RUN_NAME_SLOT;
        struct PinnedStage { void *p = nullptr; ~PinnedStage() { nmres::pinned_give(p); } } stage;
        HIP_TRY(nmres::pinned_take(&stage.p, runs_bytes + names_bytes + 64));
        const double tta = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        hipLaunchKernelGGL(bed_copy_out_kernel, dim3((unsigned)((runs_bytes / 16 + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<const uint4 *>(d_runs),
                           static_cast<uint4 *>(stage.p), runs_bytes / 16);
        if (n_named)
            hipLaunchKernelGGL(bed_copy_out_kernel, dim3((unsigned)((names_bytes / 16 + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<const uint4 *>(d_run_names),
                               reinterpret_cast<uint4 *>(static_cast<uint8_t *>(stage.p) + runs_bytes), names_bytes / 16);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double ttb = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        if (alloc_timing) fprintf(stderr, "[bed] runs: pinned buffer %.4f s, copies back %.4f s\n", tta - tt0, ttb - tta);
        const BedRun *runs_found = static_cast<const BedRun *>(stage.p);
        const uint8_t *names_found = static_cast<const uint8_t *>(stage.p) + runs_bytes;
        std::vector<unsigned int> order(n_runs);                  // the runs were recorded in the order the kernels' threads found them
        for (unsigned int r = 0; r < n_runs; ++r) order[r] = r;
        std::sort(order.begin(), order.end(), [&](unsigned int x, unsigned int y) { return runs_found[x].row < runs_found[y].row; });
        std::vector<unsigned long long> rows(n_runs), offs(n_runs);
        for (unsigned int r = 0; r < n_runs; ++r) { rows[r] = runs_found[order[r]].row; offs[r] = runs_found[order[r]].off; }
        int rc = NM_OK;
        std::unordered_map<std::string, uint32_t> ids;
        b->run_row.assign(rows.begin(), rows.end());
        b->run_contig.resize(n_runs);
        // the name of every run: from its slot, or (longer than 63 bytes, beyond RUN_NAME_CAP, NM_BED_NAMES_FROM_FILE=1) read from the file —
        // for bgzip: inflated — on several threads
        const bool names_from_file = getenv("NM_BED_NAMES_FROM_FILE") != nullptr;          // (A/B and tests: every name read from the file)
        std::vector<std::string> run_names(n_runs);
        {
            const unsigned nt = n_runs >= 64 ? std::max(1u, threads) : 1u;
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; ++t)
                pool.emplace_back([&, t] {
                    std::vector<char> buf;
                    for (unsigned int r = t; r < n_runs; r += nt) {
                        const uint8_t *slot = order[r] < n_named && !names_from_file ? &names_found[(size_t)order[r] * RUN_NAME_SLOT] : nullptr;
                        if (slot && slot[0] != 0xFF) {           // (the name came back with the run)
                            run_names[r].assign(reinterpret_cast<const char *>(slot + 1), slot[0]);
                            continue;
                        }
                        const char *fb, *fe;
                        field_in(buf, offs[r], 0, &fb, &fe);
                        run_names[r].assign(fb, fe);
                    }
                });
            for (auto &th : pool) th.join();
        }
        for (unsigned int r = 0; r < n_runs; ++r) {
            const std::string &name = run_names[r];
            auto it = ids.find(name);
            if (it == ids.end()) {
                it = ids.emplace(name, (uint32_t)b->names.size()).first;
                b->names.push_back(name);
            }
            b->run_contig[r] = it->second;
        }
        b->run_row.push_back(b->n_rows);
        const double tt1 = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        uint32_t *d_run_id = nullptr;
        HIP_TRY(tmp_alloc((void **)&d_run_id, (size_t)n_runs * 4));
        HIP_TRY(dev_malloc(&b->d_file_contig, b->n_rows * 4));
        HIP_TRY(dev_malloc(&b->d_contig, b->n_rows * 4));
        HIP_TRY(tmp_alloc((void **)&d_run_row, (size_t)n_runs * 8));
        const double tt2 = alloc_timing ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0;
        HIP_TRY(hipMemcpyAsync(d_run_row, rows.data(), (size_t)n_runs * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_run_id, b->run_contig.data(), (size_t)n_runs * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(bed_fill_contig_kernel, dim3((unsigned)((b->n_rows + 255) / 256)), dim3(256), 0, c->stream, b->n_rows, d_run_row, d_run_id, n_runs,
                           b->d_file_contig);
        HIP_TRY(hipGetLastError());
        // ---- the rows left to the host parser's routines: other mod codes (numbered in first-appearance order, like the
        // host reader does), percentages outside the fast path
        unsigned int np = 0;
        HIP_TRY(hipMemcpyAsync(&np, d_counters, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (alloc_timing) {
            const double tt3 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            fprintf(stderr, "[bed] after the last slab: %u runs and their names %.3f s, the two contig columns allocated %.3f s, filled (kernel + wait) %.3f s, %u rows for the host's routines\n",
                    n_runs, tt1 - tt0, tt2 - tt1, tt3 - tt2, np);
        }
        if (np > PATCH_CAP) return fail(NM_EDECLINED, "%s: more than %u rows need the host parser (unusual mod codes / number formats): use nm_bed_open", path, PATCH_CAP);
        if (np) {
            std::vector<uint4> patch(np);
            HIP_TRY(hipMemcpy(patch.data(), d_patch, (size_t)np * sizeof(uint4), hipMemcpyDeviceToHost));
            std::sort(patch.begin(), patch.end(), [](const uint4 &x, const uint4 &y) { return x.x < y.x; });
            std::vector<unsigned long long> prow(np), poff(np);
            for (unsigned int i = 0; i < np; ++i) { prow[i] = patch[i].x; poff[i] = ((unsigned long long)patch[i].w << 32) | patch[i].z; }
            std::vector<int8_t> pmod(np, 0);
            std::vector<double> pfrac(np, 0.0);
            std::vector<uint8_t> pwhat(np, 0);
            for (unsigned int i = 0; i < np; ++i) {
                pwhat[i] = (uint8_t)patch[i].y;
                const char *fb, *fe;
                if (patch[i].y & 1u) {
                    field(poff[i], 3, &fb, &fe);
                    const std::string code(fb, fe);
                    size_t k = 0;
                    for (; k < b->other_mods.size(); ++k)
                        if (b->other_mods[k] == code) break;
                    if (k == b->other_mods.size()) b->other_mods.push_back(code);
                    if (k > 100) return fail_row(fail(NM_EINVAL, "%s: more than 100 distinct modification codes in column 4", path));
                    pmod[i] = (int8_t)(3 + k);
                }
                if (patch[i].y & 2u) {
                    field(poff[i], 10, &fb, &fe);
                    double pct = 0;
                    if (!nmbedparse::parse_double(fb, fe, &pct)) return fail_row(fail(NM_EINVAL, "%s: pileup column 11 (percent modified) is not a number", path));
                    pfrac[i] = pct / 100.0;
                }
            }
            unsigned long long *d_prow = nullptr;
            int8_t *d_pmod = nullptr;
            double *d_pfrac = nullptr;
            uint8_t *d_pwhat = nullptr;
            HIP_TRY(tmp_alloc((void **)&d_prow, (size_t)np * 8));
            HIP_TRY(tmp_alloc((void **)&d_pmod, np));
            HIP_TRY(tmp_alloc((void **)&d_pfrac, (size_t)np * 8));
            HIP_TRY(tmp_alloc((void **)&d_pwhat, np));
            HIP_TRY(hipMemcpyAsync(d_prow, prow.data(), (size_t)np * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_pmod, pmod.data(), np, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_pfrac, pfrac.data(), (size_t)np * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_pwhat, pwhat.data(), np, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(bed_patch_kernel, dim3((np + 255) / 256), dim3(256), 0, c->stream, np, d_prow, d_pmod, d_pfrac, d_pwhat, b->d_mod, b->d_frac);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (auto &s_ : b->names) b->name_ptrs.push_back(s_.c_str());
    b->t_read = t_read;
    b->t_total = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin;
    if (alloc_timing) fprintf(stderr, "[bed] %.3f s in the parser; the runs, names and patched rows after the last slab took %.3f s\n", b->t_total, b->t_total - (t_after_slabs - t_begin));
    guard.keep = true;
    *out = b;
    return NM_OK;
}

}  // namespace

extern "C" {

int nm_bedcols_shape(nm_bedcols *b, uint64_t *n_rows, uint32_t *n_contigs, uint32_t *n_runs, double times[2]) {
    if (!b || !n_rows || !n_contigs) return fail(NM_EINVAL, "NULL argument");
    *n_rows = b->n_rows;
    *n_contigs = (uint32_t)b->names.size();
    if (n_runs) *n_runs = (uint32_t)b->run_contig.size();
    if (times) { times[0] = b->t_total; times[1] = b->t_read; }
    return NM_OK;
}

int nm_bedcols_phase_seconds(nm_bedcols *b, double out[4]) {
    if (!b || !out) return fail(NM_EINVAL, "NULL argument");
    out[0] = b->t_total;
    out[1] = b->t_read;
    out[2] = b->t_inflate;
    out[3] = b->t_total - (b->read_beside ? 0.0 : b->t_read) - b->t_inflate;       // (a staging thread's copies ran beside everything else)
    return NM_OK;
}

int nm_bedcols_contig_name(nm_bedcols *b, uint32_t i, const char **name) {
    if (!b || !name || i >= b->name_ptrs.size()) return fail(NM_EINVAL, "bad contig index");
    *name = b->name_ptrs[i];
    return NM_OK;
}

int nm_bedcols_mod_code(nm_bedcols *b, uint32_t id, const char **code) {
    static const char *known[3] = {"m", "a", "21839"};
    if (!b || !code) return fail(NM_EINVAL, "NULL argument");
    if (id < 3) { *code = known[id]; return NM_OK; }
    if (id - 3 >= b->other_mods.size()) return fail(NM_EINVAL, "mod id %u not present", id);
    *code = b->other_mods[id - 3].c_str();
    return NM_OK;
}

int nm_bedcols_runs(nm_bedcols *b, uint64_t *run_row, uint32_t *run_contig) {
    if (!b || !run_row || !run_contig) return fail(NM_EINVAL, "NULL argument");
    memcpy(run_row, b->run_row.data(), b->run_row.size() * 8);
    memcpy(run_contig, b->run_contig.data(), b->run_contig.size() * 4);
    return NM_OK;
}

int nm_bedcols_map_contigs(nm_bedcols *b, const uint32_t *contig_lut, uint32_t n_lut) {
    if (!b || !contig_lut) return fail(NM_EINVAL, "NULL argument");
    if (n_lut != b->names.size()) return fail(NM_EINVAL, "contig_lut has %u entries, the pileup names %zu contigs", n_lut, b->names.size());
    if (b->n_rows == 0) return NM_OK;
    nm_ctx *c = b->ctx;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *d_lut = nullptr;
    HIP_TRY(dev_malloc(&d_lut, (size_t)std::max(n_lut, 1u) * 4));
    HIP_TRY(hipMemcpyAsync(d_lut, contig_lut, (size_t)n_lut * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bed_map_contigs_kernel, dim3((unsigned)((b->n_rows + 255) / 256)), dim3(256), 0, c->stream, b->n_rows, d_lut, b->d_file_contig, b->d_contig);
    const hipError_t e = hipGetLastError();
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)dev_free(d_lut);
    if (e != hipSuccess || e2 != hipSuccess) return fail(NM_EHIP, "contig mapping failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return NM_OK;
}

int nm_device_read(nm_ctx *c, void *host_dst, const void *device_src, uint64_t bytes) {
    if (!c || (bytes && (!host_dst || !device_src))) return fail(NM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NM_OK;
}

int nm_bedcols_device_columns(nm_bedcols *b, const uint32_t **contig_id, const uint32_t **file_contig_id, const uint32_t **position, const int8_t **mod_type,
                              const uint8_t **strand, const double **fraction_mod, const int32_t **nvalid_cov) {
    if (!b) return fail(NM_EINVAL, "NULL argument");
    if (contig_id) *contig_id = b->d_contig;
    if (file_contig_id) *file_contig_id = b->d_file_contig;
    if (position) *position = b->d_position;
    if (mod_type) *mod_type = b->d_mod;
    if (strand) *strand = b->d_strand;
    if (fraction_mod) *fraction_mod = b->d_frac;
    if (nvalid_cov) *nvalid_cov = b->d_nvalid;
    return NM_OK;
}

}  // extern "C"
