// Assembly FASTA parsed ON THE DEVICE (round 5) — replaces pyfastx / the line loop of fasta.py:35-49 and DNAsequence's
// checks (seq.py:53-71) for plain-text assemblies, like nmbedgpu.hip does for the pileup: the host only moves the file
// through a ring of pinned slabs into HBM, kernels find the records and compact their sequence bytes.
//
//   text (n bytes) --fa_count--> per 16 KiB tile: bytes that are neither '\n' nor '\r', header starts ('>' at a line start)
//        --scan--> C(tile) = such bytes before the tile (one 64-bit prefix: C(x) counts header bytes too, but only DIFFERENCES
//                  inside a record body are ever used, and a body holds no header byte), header starts before the tile
//        --fa_headers--> position of every header, in file order
//        --fa_records--> per record (one wave): end of its header line, first body byte, C at both ends of the body -> length
//        --scan--> destination of every record in the packed sequence, of every header line in the packed header text
//        --fa_compact--> body bytes upper-cased (seq.py:55), checked against ATGCRYSWKMBDHVN (seq.py:68-71), written back to
//                  back: exactly the array nm_fasta_sequence returns, but in device memory — nm_upload_contigs_fasta packs the
//                  wanted records into the planes from there (pack_kernel), no sequence byte ever becomes a host array
//        --fa_hdr_gather--> the header lines, copied to the host for the record names (first whitespace-delimited token)
//
// HBM-bound byte work: 1 B read per counting pass (3 passes over the text) + 1 B written per base; at 1 Gbp ~4 GB of traffic,
// < 2 ms — the wall of this phase is the file read (page cache -> pinned -> PCIe), which is why nothing is done on the host
// but the copy.  A gzip assembly keeps the host reader (nm_fasta_open).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <rocprim/device/device_scan.hpp>

#include "nmscan_internal.h"
#include "nmres.h"

using namespace nmdetail;

struct nm_fastadev {
    nm_ctx *ctx = nullptr;
    std::vector<std::string> names;
    std::string names_blob;                   // the names, each followed by a NUL (nm_fastadev_table)
    std::vector<uint64_t> offset;             // n + 1: where each record's bases start in d_seq
    uint8_t *d_seq = nullptr;                 // every record's bases, upper-cased, back to back (file order)
    double seconds = 0, seconds_reading = 0;
    uint64_t file_bytes = 0;
};

namespace {

constexpr uint32_t FA_TILE = 16384;           // text bytes per workgroup: 256 threads x 64
constexpr uint64_t FA_SLAB = 32ull << 20;     // pinned slab of the copy ring

// bit j of the result: byte j of the 64 bytes v[0..3] equals `c` (exact per-byte zero test, no borrow between bytes)
__device__ inline unsigned long long eq_mask64(const uint4 (&v)[4], uint32_t c) {
    const uint32_t pat = c * 0x01010101u;
    unsigned long long out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t y = w[d] ^ pat;
            uint32_t t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
            t = ~(t | y | 0x7F7F7F7Fu);                                  // 0x80 in every byte of y that is zero
            const uint32_t nib = ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
            out |= (unsigned long long)nib << (16 * q + 4 * d);
        }
    }
    return out;
}

struct FaMasks {
    unsigned long long seq;       // bytes that are neither '\n' nor '\r' (inside the file)
    unsigned long long hdr;       // '>' at the start of a line
};

// the 64 bytes at `base` (a multiple of 64; the buffer is readable up to the next multiple of FA_TILE)
__device__ inline FaMasks fa_masks(const uint8_t *__restrict__ text, uint64_t base, uint64_t n, uint4 (&v)[4]) {
    FaMasks m{0, 0};
    if (base >= n) return m;
    const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = p[q];
    const unsigned long long valid = n - base >= 64 ? ~0ull : ((1ull << (n - base)) - 1);
    const unsigned long long nl = eq_mask64(v, '\n'), cr = eq_mask64(v, '\r'), gt = eq_mask64(v, '>');
    const unsigned long long prev_nl = base == 0 ? 1ull : (text[base - 1] == '\n' ? 1ull : 0ull);
    m.seq = valid & ~nl & ~cr;
    m.hdr = valid & gt & ((nl << 1) | prev_nl);
    return m;
}

__device__ inline uint32_t wave_sum(uint32_t x) {
    for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// (1) per tile: sequence-alphabet bytes and header starts
__global__ __launch_bounds__(256) void fa_count_kernel(const uint8_t *__restrict__ text, uint64_t n, unsigned long long *__restrict__ tile_seq,
                                                       uint32_t *__restrict__ tile_hdr) {
    __shared__ uint32_t part[2][4];
    uint4 v[4];
    const FaMasks m = fa_masks(text, (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * 64, n, v);
    const uint32_t s = wave_sum((uint32_t)__popcll(m.seq)), h = wave_sum((uint32_t)__popcll(m.hdr));
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_seq[blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        tile_hdr[blockIdx.x] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}

// exclusive prefix over the 256 threads of a workgroup (value per thread -> sum of the threads before it)
__device__ inline uint32_t block_exclusive(uint32_t mine, uint32_t *scan) {
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)d ? scan[threadIdx.x - d] : 0;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    return scan[threadIdx.x] - mine;
}

// (2) the positions of the header starts, in file order (tiles without one leave at once)
__global__ __launch_bounds__(256) void fa_headers_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ tile_hdr_off,
                                                         unsigned long long *__restrict__ hdr_pos) {
    __shared__ uint32_t scan[256];
    if (tile_hdr_off[blockIdx.x + 1] == tile_hdr_off[blockIdx.x]) return;
    uint4 v[4];
    const uint64_t base = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * 64;
    unsigned long long h = fa_masks(text, base, n, v).hdr;
    uint32_t at = tile_hdr_off[blockIdx.x] + block_exclusive((uint32_t)__popcll(h), scan);
    while (h) {
        const int k = __ffsll((long long)h) - 1;
        h &= h - 1;
        hdr_pos[at++] = base + k;
    }
}

// C(x): bytes of [0, x) that are neither '\n' nor '\r' — the tile prefix plus the part of x's tile in front of x, counted by
// the 64 lanes of the calling wave (all lanes return the value)
__device__ inline unsigned long long fa_count_upto(const uint8_t *__restrict__ text, uint64_t x, const unsigned long long *__restrict__ tile_cum) {
    const uint64_t t0 = x / FA_TILE * FA_TILE;
    uint32_t cnt = 0;
    uint4 v[4];
    for (uint64_t b = t0 + (uint64_t)(threadIdx.x & 63) * 64; b < x; b += 64 * 64) cnt += (uint32_t)__popcll(fa_masks(text, b, x, v).seq);
    return tile_cum[x / FA_TILE] + wave_sum(cnt);
}

// (3) one wave per record: the end of the header line, where the body starts, C at both ends of the body
__global__ __launch_bounds__(256) void fa_records_kernel(const uint8_t *__restrict__ text, uint64_t n, const unsigned long long *__restrict__ hdr_pos,
                                                         uint32_t n_rec, const unsigned long long *__restrict__ tile_cum,
                                                         unsigned long long *__restrict__ body_beg, unsigned long long *__restrict__ rec_cb,
                                                         unsigned long long *__restrict__ rec_len, unsigned long long *__restrict__ hdr_len) {
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_rec) return;
    const uint64_t h = hdr_pos[k];
    uint64_t eol = n;
    for (uint64_t p = h; p < n; p += 64) {
        const bool is_nl = p + lane < n && text[p + lane] == '\n';
        const unsigned long long b = __ballot(is_nl);
        if (b) { eol = p + (uint64_t)(__ffsll((long long)b) - 1); break; }
    }
    const uint64_t body = eol < n ? eol + 1 : n;
    const uint64_t end = k + 1 < n_rec ? hdr_pos[k + 1] : n;
    const unsigned long long cb = fa_count_upto(text, body, tile_cum), ce = fa_count_upto(text, end, tile_cum);
    if (lane == 0) {
        body_beg[k] = body;
        rec_cb[k] = cb;
        rec_len[k] = ce - cb;
        hdr_len[k] = eol - (h + 1);                                    // without '>' and without the '\n'
    }
}

__device__ inline bool fa_letter_ok(uint32_t c) {
    // ATGCRYSWKMBDHVN (seq.py:68-71; DNAsequence upper-cases first, seq.py:55)
    constexpr uint32_t ok = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 6) | (1u << 7) | (1u << 10) | (1u << 12) | (1u << 13) | (1u << 17) |
                            (1u << 18) | (1u << 19) | (1u << 21) | (1u << 22) | (1u << 24);
    const uint32_t k = c - 'A';
    return k < 26 && ((ok >> k) & 1u);
}

// (4) body bytes -> the packed sequence.  A byte at text position p of record r lands at rec_dst[r] + C(p) - C(body start of r).
__global__ __launch_bounds__(256) void fa_compact_kernel(const uint8_t *__restrict__ text, uint64_t n, const unsigned long long *__restrict__ tile_cum,
                                                         const uint32_t *__restrict__ tile_hdr_off, const unsigned long long *__restrict__ hdr_pos,
                                                         const unsigned long long *__restrict__ body_beg, const unsigned long long *__restrict__ rec_cb,
                                                         const unsigned long long *__restrict__ rec_dst, uint8_t *__restrict__ out,
                                                         uint8_t *__restrict__ bad) {
    __shared__ uint32_t scan[256];
    uint4 v[4];
    const uint64_t base = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * 64;
    const FaMasks m = fa_masks(text, base, n, v);
    const unsigned long long c0 = tile_cum[blockIdx.x] + block_exclusive((uint32_t)__popcll(m.seq), scan);      // C(base)
    const uint32_t lo = tile_hdr_off[blockIdx.x], hi = tile_hdr_off[blockIdx.x + 1];     // headers inside this tile: [lo, hi)
    if (!m.seq) return;
    if (lo == hi) {
        // the whole tile belongs to record lo - 1 (or lies in front of the first header: no record)
        if (lo == 0) return;
        const uint32_t r = lo - 1;
        const uint64_t bb = body_beg[r];
        unsigned long long take = m.seq;
        if (bb > base) take = bb - base >= 64 ? 0ull : take & ~((1ull << (bb - base)) - 1);      // the tail of the header line
        if (!take) return;
        uint8_t *dst = out + rec_dst[r] + (c0 + (unsigned long long)__popcll(m.seq & ~take) - rec_cb[r]);
        bool wrong = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((take >> (16 * q + 4 * d + b)) & 1ull) {
                        uint32_t c = (w[d] >> (8 * b)) & 0xFFu;
                        if (c - 'a' < 26u) c -= 32;
                        wrong |= !fa_letter_ok(c);
                        *dst++ = (uint8_t)c;
                    }
        }
        if (wrong) bad[r] = 1;
        return;
    }
    // a tile with header starts (one in ~n_tiles / n_records): byte by byte, the record looked up as the bytes go by
    int64_t r = (int64_t)lo - 1;
    unsigned long long left = m.seq;
    while (left) {
        const int j = __ffsll((long long)left) - 1;
        left &= left - 1;
        const uint64_t p = base + j;
        while (r + 1 < (int64_t)hi && hdr_pos[r + 1] <= p) ++r;
        if (r < 0 || p < body_beg[r]) continue;
        uint32_t c = text[p];
        if (c - 'a' < 26u) c -= 32;
        if (!fa_letter_ok(c)) bad[r] = 1;
        out[rec_dst[r] + (c0 + (unsigned long long)__popcll(m.seq & ((1ull << j) - 1)) - rec_cb[r])] = (uint8_t)c;
    }
}

// (5) one wave per record: its header line (without '>') into the packed header text
__global__ __launch_bounds__(256) void fa_hdr_gather_kernel(const uint8_t *__restrict__ text, const unsigned long long *__restrict__ hdr_pos,
                                                            const unsigned long long *__restrict__ hdr_off, uint32_t n_rec, uint8_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_rec) return;
    const uint64_t src = hdr_pos[k] + 1, len = hdr_off[k + 1] - hdr_off[k];
    for (uint64_t i = lane; i < len; i += 64) out[hdr_off[k] + i] = text[src + i];
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool pread_all(int fd, uint8_t *dst, uint64_t off, uint64_t len) {
    while (len) {
        const ssize_t k = pread(fd, dst, (size_t)std::min<uint64_t>(len, 1u << 30), (off_t)off);
        if (k <= 0) return false;
        dst += k; off += (uint64_t)k; len -= (uint64_t)k;
    }
    return true;
}

}  // namespace

extern "C" {

int nm_fastadev_close(nm_fastadev *f) {
    if (!f) return NM_OK;
    if (f->d_seq && f->ctx) {
        (void)hipSetDevice(f->ctx->device);
        (void)hipStreamSynchronize(f->ctx->stream);
        (void)dev_free(f->d_seq);
    }
    delete f;
    return NM_OK;
}

int nm_fasta_parse_device(nm_ctx *c, const char *path, uint32_t threads, nm_fastadev **out) {
    if (!c || !path || !out) return fail(NM_EINVAL, "NULL argument");
    *out = nullptr;
    const double t_begin = now_s();
    const bool timing = getenv("NM_FASTA_TIMING") != nullptr;     // where the wall time of the call goes (stderr)
    if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(NM_EINVAL, "cannot open assembly '%s'", path);
    struct CloseFd { int fd; ~CloseFd() { close(fd); } } close_fd{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) return fail(NM_EINVAL, "cannot stat assembly '%s'", path);
    const uint64_t n = (uint64_t)st.st_size;
    uint8_t magic[2] = {0, 0};
    if (n >= 2 && !pread_all(fd, magic, 0, 2)) return fail(NM_EINVAL, "cannot read assembly '%s'", path);
    if (magic[0] == 31 && magic[1] == 139)
        return fail(NM_EDECLINED, "%s: compressed input: the device parser reads plain-text FASTA (use nm_fasta_open)", path);
    nm_fastadev *f = new (std::nothrow) nm_fastadev();
    if (!f) return fail(NM_ENOMEM, "out of host memory");
    f->ctx = c;
    f->file_bytes = n;
    f->offset.assign(1, 0);
    struct Guard { nm_fastadev *f; bool keep = false; ~Guard() { if (!keep) (void)nm_fastadev_close(f); } } guard{f};
    if (n == 0) {                                      // an empty file holds no record (fasta.py:35-49 returns an empty dict)
        guard.keep = true;
        *out = f;
        return NM_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n_tiles = (uint32_t)((n + FA_TILE - 1) / FA_TILE);
    if ((uint64_t)n_tiles * FA_TILE < n) return fail(NM_ERANGE, "%s: assembly file too large", path);
    // ---- the file -> device memory through a ring of pinned slabs filled by `threads` readers
    constexpr int RING = 3;
    uint8_t *h_ring[RING] = {nullptr, nullptr, nullptr};
    hipEvent_t h2d_done[RING] = {nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;                 // the ctx's copy stream (idle during this call); one of its own only if the ctx has none
    bool own_copy_stream = false;
    std::vector<void *> dev_tmp;
    struct Cleanup {
        uint8_t **h; hipEvent_t *e; hipStream_t &cs; bool &own; std::vector<void *> &tmp; nm_ctx *c;
        ~Cleanup() {
            (void)hipStreamSynchronize(c->stream);
            if (cs) (void)hipStreamSynchronize(cs);
            for (int i = 0; i < RING; ++i) { nmres::pinned_give(h[i]); if (e[i]) (void)hipEventDestroy(e[i]); }      // (kept for the next parser: nmres.h)
            for (void *p : tmp) (void)dev_free(p);
            if (cs && own) (void)hipStreamDestroy(cs);
        }
    } cleanup{h_ring, h2d_done, copy_stream, own_copy_stream, dev_tmp, c};
    auto tmp_alloc = [&](void **p, size_t bytes) -> hipError_t {
        const hipError_t e = device_alloc(p, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) dev_tmp.push_back(*p);
        return e;
    };
    uint8_t *d_text = nullptr;
    HIP_TRY(tmp_alloc((void **)&d_text, (size_t)n_tiles * FA_TILE));
    const double t_alloc_dev = now_s();
    const size_t n_slabs = (size_t)((n + FA_SLAB - 1) / FA_SLAB);
    const uint64_t slab_cap = std::min<uint64_t>(FA_SLAB, n);
    // (+ 64 KB: the size of the pileup parser's chunks, which take these buffers over from the cache)
    for (int i = 0; i < RING && (size_t)i < n_slabs; ++i) HIP_TRY(nmres::pinned_take((void **)&h_ring[i], slab_cap + (1u << 16)));
    for (int i = 0; i < RING; ++i) HIP_TRY(hipEventCreateWithFlags(&h2d_done[i], hipEventDisableTiming));
    copy_stream = getenv("NM_OWN_COPY_STREAM") ? nullptr : c->copy_stream;        // (A/B: a stream of its own, as before round 6)
    if (!copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        own_copy_stream = true;
    }
    const double t_alloc = now_s();
    std::atomic<bool> read_failed{false};
    std::mutex mu;
    std::condition_variable cv;
    size_t filled = 0, consumed = 0;                   // slabs complete in the ring / slabs whose copy to the device has finished
    bool stop = false;
    double t_read = 0;
    const unsigned nt = std::max(1u, std::min<unsigned>(threads, 16u));
    std::vector<unsigned> shares_done(n_slabs, 0);
    std::vector<std::thread> readers;
    for (unsigned t = 0; t < nt; ++t)
        readers.emplace_back([&, t] {
            for (size_t k = 0; k < n_slabs; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < consumed + RING; });
                    if (stop) return;
                }
                const double t0 = now_s();
                const uint64_t lo = (uint64_t)k * FA_SLAB, len = std::min<uint64_t>(FA_SLAB, n - lo);
                const uint64_t a = len * t / nt, e = len * (t + 1) / nt;
                if (e > a && !pread_all(fd, h_ring[k % RING] + a, lo + a, e - a)) read_failed = true;
                const double dt = now_s() - t0;
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
    } join{readers, mu, cv, stop};
    for (size_t k = 0; k < n_slabs; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled > k; });
        }
        if (read_failed) return fail(NM_EINVAL, "cannot read assembly '%s'", path);
        const uint64_t lo = (uint64_t)k * FA_SLAB, len = std::min<uint64_t>(FA_SLAB, n - lo);
        HIP_TRY(hipMemcpyAsync(d_text + lo, h_ring[k % RING], len, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipEventRecord(h2d_done[k % RING], copy_stream));
        if (k >= 1) {                                  // the slab before this one has left its pinned buffer: the readers may have it
            HIP_TRY(hipEventSynchronize(h2d_done[(k - 1) % RING]));
            { std::lock_guard<std::mutex> lk(mu); consumed = k; }
            cv.notify_all();
        }
    }
    HIP_TRY(hipStreamWaitEvent(c->stream, h2d_done[(n_slabs - 1) % RING], 0));
    f->seconds_reading = t_read;
    const double t_copied = now_s();
    // ---- records
    busy_begin(c);
    unsigned long long *d_tile_seq = nullptr, *d_tile_cum = nullptr;
    uint32_t *d_tile_hdr = nullptr, *d_tile_hdr_off = nullptr;
    HIP_TRY(tmp_alloc((void **)&d_tile_seq, ((size_t)n_tiles + 1) * 8));
    HIP_TRY(tmp_alloc((void **)&d_tile_cum, ((size_t)n_tiles + 1) * 8));
    HIP_TRY(tmp_alloc((void **)&d_tile_hdr, ((size_t)n_tiles + 1) * 4));
    HIP_TRY(tmp_alloc((void **)&d_tile_hdr_off, ((size_t)n_tiles + 1) * 4));
    HIP_TRY(hipMemsetAsync(d_tile_seq + n_tiles, 0, 8, c->stream));
    HIP_TRY(hipMemsetAsync(d_tile_hdr + n_tiles, 0, 4, c->stream));
    hipLaunchKernelGGL(fa_count_kernel, dim3(n_tiles), dim3(256), 0, c->stream, d_text, n, d_tile_seq, d_tile_hdr);
    HIP_TRY(hipGetLastError());
    size_t scan_bytes = 0, scan_bytes2 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_tile_seq, d_tile_cum, 0ull, (size_t)n_tiles + 1, rocprim::plus<unsigned long long>(), c->stream));
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes2, d_tile_hdr, d_tile_hdr_off, 0u, (size_t)n_tiles + 1, rocprim::plus<unsigned int>(), c->stream));
    void *d_scan_tmp = nullptr;
    HIP_TRY(tmp_alloc(&d_scan_tmp, std::max(scan_bytes, scan_bytes2)));
    HIP_TRY(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_tile_seq, d_tile_cum, 0ull, (size_t)n_tiles + 1, rocprim::plus<unsigned long long>(), c->stream));
    HIP_TRY(rocprim::exclusive_scan(d_scan_tmp, scan_bytes2, d_tile_hdr, d_tile_hdr_off, 0u, (size_t)n_tiles + 1, rocprim::plus<unsigned int>(), c->stream));
    uint32_t n_rec = 0;
    HIP_TRY(hipMemcpyAsync(&n_rec, d_tile_hdr_off + n_tiles, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_rec == 0) {                                  // text without a header line: no record
        busy_end(c);
        f->seconds = now_s() - t_begin;
        guard.keep = true;
        *out = f;
        return NM_OK;
    }
    unsigned long long *d_hdr_pos = nullptr, *d_body = nullptr, *d_cb = nullptr, *d_len = nullptr, *d_dst = nullptr, *d_hlen = nullptr, *d_hoff = nullptr;
    uint8_t *d_bad = nullptr;
    const size_t rec_bytes = ((size_t)n_rec + 1) * 8;
    HIP_TRY(tmp_alloc((void **)&d_hdr_pos, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_body, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_cb, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_len, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_dst, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_hlen, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_hoff, rec_bytes));
    HIP_TRY(tmp_alloc((void **)&d_bad, n_rec));
    HIP_TRY(hipMemsetAsync(d_bad, 0, n_rec, c->stream));
    HIP_TRY(hipMemsetAsync(d_len + n_rec, 0, 8, c->stream));
    HIP_TRY(hipMemsetAsync(d_hlen + n_rec, 0, 8, c->stream));
    hipLaunchKernelGGL(fa_headers_kernel, dim3(n_tiles), dim3(256), 0, c->stream, d_text, n, d_tile_hdr_off, d_hdr_pos);
    hipLaunchKernelGGL(fa_records_kernel, dim3((n_rec + 3) / 4), dim3(256), 0, c->stream, d_text, n, d_hdr_pos, n_rec, d_tile_cum, d_body, d_cb, d_len, d_hlen);
    HIP_TRY(hipGetLastError());
    size_t scan_bytes3 = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes3, d_len, d_dst, 0ull, (size_t)n_rec + 1, rocprim::plus<unsigned long long>(), c->stream));
    void *d_scan_tmp3 = nullptr;
    HIP_TRY(tmp_alloc(&d_scan_tmp3, scan_bytes3));
    HIP_TRY(rocprim::exclusive_scan(d_scan_tmp3, scan_bytes3, d_len, d_dst, 0ull, (size_t)n_rec + 1, rocprim::plus<unsigned long long>(), c->stream));
    HIP_TRY(rocprim::exclusive_scan(d_scan_tmp3, scan_bytes3, d_hlen, d_hoff, 0ull, (size_t)n_rec + 1, rocprim::plus<unsigned long long>(), c->stream));
    f->offset.assign((size_t)n_rec + 1, 0);
    std::vector<unsigned long long> hoff((size_t)n_rec + 1, 0);
    HIP_TRY(hipMemcpyAsync(f->offset.data(), d_dst, rec_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(hoff.data(), d_hoff, rec_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t total_bp = f->offset[n_rec], hdr_bytes = hoff[n_rec];
    HIP_TRY(dev_malloc(&f->d_seq, std::max<uint64_t>(total_bp, 16)));
    uint8_t *d_hdr_text = nullptr;
    HIP_TRY(tmp_alloc((void **)&d_hdr_text, hdr_bytes));
    hipLaunchKernelGGL(fa_compact_kernel, dim3(n_tiles), dim3(256), 0, c->stream, d_text, n, d_tile_cum, d_tile_hdr_off, d_hdr_pos, d_body, d_cb, d_dst,
                       f->d_seq, d_bad);
    hipLaunchKernelGGL(fa_hdr_gather_kernel, dim3((n_rec + 3) / 4), dim3(256), 0, c->stream, d_text, d_hdr_pos, d_hoff, n_rec, d_hdr_text);
    HIP_TRY(hipGetLastError());
    std::vector<uint8_t> bad(n_rec, 0);
    std::string hdr_text(hdr_bytes, '\0');
    HIP_TRY(hipMemcpyAsync(bad.data(), d_bad, n_rec, hipMemcpyDeviceToHost, c->stream));
    if (hdr_bytes) HIP_TRY(hipMemcpyAsync(&hdr_text[0], d_hdr_text, hdr_bytes, hipMemcpyDeviceToHost, c->stream));
    busy_end(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    // names: the first whitespace-delimited token of the header line (str.split(), like nm_fasta_open); the first empty or
    // non-IUPAC record in file order is the error
    auto is_space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v'; };
    f->names.reserve(n_rec);
    for (uint32_t i = 0; i < n_rec; ++i) {
        const char *h = hdr_text.data() + hoff[i], *he = hdr_text.data() + hoff[i + 1];
        while (h < he && is_space(*h)) ++h;
        const char *t = h;
        while (t < he && !is_space(*t)) ++t;
        f->names.emplace_back(h, (size_t)(t - h));
        const bool empty = f->offset[i + 1] == f->offset[i];
        if (empty || bad[i])
            return empty ? fail(NM_ESEQUENCE, "DNA sequence must not be empty (record '%s')", f->names.back().c_str())
                         : fail(NM_ESEQUENCE, "DNA sequence must be a nucleotide sequence of ATGCRYSWKMBDHVN (record '%s')", f->names.back().c_str());
    }
    f->seconds = now_s() - t_begin;
    if (timing)
        fprintf(stderr, "[fasta] %.2f GB, %u records: buffers %.3f s (the text's device buffer %.3f s, pinned ring + stream %.3f s), file -> device %.3f s "
                        "(a reader spent %.3f s in pread), kernels + tables %.3f s\n", n / 1e9, n_rec,
                t_alloc - t_begin, t_alloc_dev - t_begin, t_alloc - t_alloc_dev, t_copied - t_alloc, t_read, now_s() - t_copied);
    guard.keep = true;
    *out = f;
    return NM_OK;
}

int nm_warm_file_parsers(nm_ctx *c, uint64_t bytes_each, uint32_t count) {
    if (count > 8 || bytes_each > (256ull << 20)) return fail(NM_ERANGE, "nm_warm_file_parsers: at most 8 buffers of 256 MB");
    if (c) HIP_TRY(hipSetDevice(c->device));
    void *bufs[8] = {nullptr};
    int rc = NM_OK;
    for (uint32_t i = 0; i < count && rc == NM_OK; ++i)
        if (nmres::pinned_take(&bufs[i], (size_t)bytes_each) != hipSuccess) rc = fail(NM_ENOMEM, "cannot pin %llu bytes of host memory", (unsigned long long)bytes_each);
    if (rc == NM_OK && count && bytes_each >= 64 && c && c->copy_stream) {       // one transfer each way: the copy engines' queues of the stream come up now
        void *d = nullptr;
        if (device_alloc(&d, 64) == hipSuccess) {
            hipError_t e = hipMemcpyAsync(d, bufs[0], 64, hipMemcpyHostToDevice, c->copy_stream);
            if (e == hipSuccess) e = hipMemcpyAsync(bufs[0], d, 64, hipMemcpyDeviceToHost, c->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
            (void)dev_free(d);
            if (e != hipSuccess) rc = fail(NM_EHIP, "nm_warm_file_parsers: %s", hipGetErrorString(e));
        }
    }
    for (uint32_t i = 0; i < count; ++i) nmres::pinned_give(bufs[i]);
    return rc;
}

int nm_fastadev_shape(nm_fastadev *f, uint32_t *n_records, uint64_t *total_bp, double times[2]) {
    if (!f || !n_records || !total_bp) return fail(NM_EINVAL, "NULL argument");
    *n_records = (uint32_t)f->names.size();
    *total_bp = f->offset.back();
    if (times) { times[0] = f->seconds; times[1] = f->seconds_reading; }
    return NM_OK;
}

int nm_fastadev_record(nm_fastadev *f, uint32_t i, const char **name, uint64_t *offset, uint64_t *length) {
    if (!f || !name || !offset || !length || i >= f->names.size()) return fail(NM_EINVAL, "bad record index");
    *name = f->names[i].c_str();
    *offset = f->offset[i];
    *length = f->offset[i + 1] - f->offset[i];
    return NM_OK;
}

int nm_fastadev_table(nm_fastadev *f, const char **names, uint64_t *names_bytes, const uint64_t **offsets) {
    if (!f || !names || !names_bytes || !offsets) return fail(NM_EINVAL, "NULL argument");
    if (f->names_blob.empty())
        for (const std::string &nm : f->names) { f->names_blob.append(nm); f->names_blob.push_back('\0'); }
    *names = f->names_blob.data();
    *names_bytes = f->names_blob.size();
    *offsets = f->offset.data();
    return NM_OK;
}

int nm_fastadev_sequence_device(nm_fastadev *f, const uint8_t **d_seq_upper) {
    if (!f || !d_seq_upper) return fail(NM_EINVAL, "NULL argument");
    *d_seq_upper = f->d_seq;
    return NM_OK;
}

int nm_upload_contigs_fasta(nm_ctx *c, nm_fastadev *f, uint32_t n_contigs, const uint32_t *record, const uint32_t *bin_id, uint32_t n_bins) {
    if (!c || !f || (n_contigs && (!record || !bin_id))) return fail(NM_EINVAL, "NULL argument");
    if (f->ctx != c) return fail(NM_EINVAL, "the assembly was parsed on another context");
    std::vector<uint64_t> offsets((size_t)n_contigs + 1, 0), src((size_t)n_contigs, 0);
    for (uint32_t i = 0; i < n_contigs; ++i) {
        if (record[i] >= f->names.size()) return fail(NM_EINVAL, "contig %u: record %u >= %zu records of the assembly", i, record[i], f->names.size());
        src[i] = f->offset[record[i]];
        offsets[i + 1] = offsets[i] + (f->offset[record[i] + 1] - f->offset[record[i]]);
    }
    return upload_contigs_gather(c, n_contigs, offsets.data(), src.data(), bin_id, n_bins, f->d_seq);
}

}  // extern "C"
