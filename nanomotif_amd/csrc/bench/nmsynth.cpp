// libnmsynth — synthetic-data tooling of bench.py and the tests: NOT part of the product library (libnmscan.so) or of its C ABI.
//   nm_synth_write_bed   rows -> modkit bedMethyl text on several threads
//   nm_synth_bgzip       a text file -> bgzip (.gz) + tabix index (.gz.tbi), blocks deflated on several threads: what
//                        `bgzip p.bed; tabix -p bed p.bed.gz` produce (SAM spec 4.1 BGZF; tabix.pdf), htslib is not in the image
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {
thread_local std::string g_err;
int synth_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
}  // namespace

extern "C" {

const char *nm_synth_last_error(void) { return g_err.c_str(); }

// Synthetic-data tooling: rows -> modkit bedMethyl text (18 tab-separated columns, what synth.SynthMetagenome.write_bed
// writes row by row in Python), formatted on several threads.  pct_hundredths = percent modified in 1/100 %.
int nm_synth_write_bed(const char *path, uint64_t n_rows, uint32_t n_contigs, const char *names, const uint32_t *name_offset,
                       const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_type, const uint8_t *strand,
                       const int32_t *nvalid_cov, const int32_t *pct_hundredths, uint32_t threads) {
    if (!path || (n_rows && (!names || !name_offset || !contig_id || !position || !mod_type || !strand || !nvalid_cov || !pct_hundredths)))
        return synth_error("NULL argument");
    if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    static const char *codes[3] = {"m", "a", "21839"};
    FILE *f = fopen(path, "wb");
    if (!f) return synth_error("cannot create '%s'", path);
    const uint64_t piece = 1u << 20;                                   // rows per round and thread (~80 MB of text each)
    std::vector<std::string> buf(threads);
    int bad = 0;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += piece * threads) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                std::string &o = buf[t];
                o.clear();
                const uint64_t a = std::min<uint64_t>(n_rows, r0 + piece * t), e = std::min<uint64_t>(n_rows, a + piece);
                char tmp[256];
                for (uint64_t i = a; i < e; ++i) {
                    const uint32_t c = contig_id[i];
                    const int m = mod_type[i];
                    if (c >= n_contigs || m < 0 || m > 2) { bad = 1; return; }
                    const long long cov = nvalid_cov[i], pct = pct_hundredths[i], pos = position[i];
                    const long long nmod = (long long)std::nearbyint((double)(cov * pct) / 10000.0);    // Python's round(): half to even
                    o.append(names + name_offset[c], name_offset[c + 1] - name_offset[c]);
                    const int k = snprintf(tmp, sizeof tmp, "\t%lld\t%lld\t%s\t%lld\t%c\t%lld\t%lld\t255,0,0\t%lld\t%lld.%02lld\t%lld\t%lld\t0\t0\t0\t0\t0\n",
                                           pos, pos + 1, codes[m], cov, (char)strand[i], pos, pos + 1, cov, pct / 100, pct % 100, nmod, cov - nmod);
                    o.append(tmp, (size_t)k);
                }
            });
        for (auto &th : pool) th.join();
        if (bad) break;
        for (unsigned t = 0; t < threads; ++t)
            if (!buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), f) != buf[t].size()) bad = 2;
    }
    fclose(f);
    if (bad == 1) return synth_error("row with a contig id / mod code outside the tables");
    if (bad == 2) return synth_error("short write to '%s'", path);
    return 0;
}


// text file -> BGZF blocks of `block_size` text bytes (cut anywhere, like bgzip does) + the tabix index of a BED-like file
// (sequence name column 1, begin column 2, end column 3, 0-based): per sequence the bins with their chunks, the metadata
// pseudo-bin 37450 and the 16 kb linear index.
int nm_synth_bgzip(const char *text_path, const char *gz_path, uint32_t threads, int level, uint32_t block_size) {
    if (!text_path || !gz_path) return synth_error("NULL argument");
    if (threads == 0) threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (block_size == 0 || block_size > 0xFF00) block_size = 0xFF00;
    const int fd = open(text_path, O_RDONLY);
    if (fd < 0) return synth_error("cannot open '%s'", text_path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return synth_error("cannot stat '%s'", text_path); }
    const size_t n = (size_t)st.st_size;
    const uint8_t *text = nullptr;
    if (n) {
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return synth_error("cannot map '%s'", text_path); }
        text = static_cast<const uint8_t *>(m);
    }
    close(fd);
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { if (p) munmap(const_cast<uint8_t *>(p), n); } } unmap{text, n};
    auto block = [&](const uint8_t *src, size_t len, std::string *out) -> bool {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        std::vector<uint8_t> comp(deflateBound(&zs, (uLong)len) + 16);
        zs.next_in = const_cast<Bytef *>(src);
        zs.avail_in = (uInt)len;
        zs.next_out = comp.data();
        zs.avail_out = (uInt)comp.size();
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = comp.size() - zs.avail_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
        const uint16_t bsize = (uint16_t)(clen + 25);
        const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255), (uint8_t)(bsize >> 8)};
        out->append(reinterpret_cast<const char *>(hdr), 18);
        out->append(reinterpret_cast<const char *>(comp.data()), clen);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len), isize = (uint32_t)len;
        out->append(reinterpret_cast<const char *>(&crc), 4);
        out->append(reinterpret_cast<const char *>(&isize), 4);
        return true;
    };
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const size_t n_blocks = (n + block_size - 1) / block_size;
    std::vector<uint64_t> coff(n_blocks + 1, 0);
    FILE *f = fopen(gz_path, "wb");
    if (!f) return synth_error("cannot create '%s'", gz_path);
    // rounds of `threads` x 256 blocks: deflated in parallel, written in order
    const size_t per = 256;
    std::vector<std::string> buf(threads);
    std::vector<std::vector<uint32_t>> sizes(threads);
    bool bad = false;
    for (size_t b0 = 0; b0 < n_blocks && !bad; b0 += per * threads) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                buf[t].clear();
                sizes[t].clear();
                for (size_t b = b0 + per * t; b < std::min(n_blocks, b0 + per * (t + 1)); ++b) {
                    const size_t before = buf[t].size();
                    if (!block(text + b * block_size, std::min<size_t>(block_size, n - b * block_size), &buf[t])) { bad = true; return; }
                    sizes[t].push_back((uint32_t)(buf[t].size() - before));
                }
            });
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < threads && !bad; ++t) {
            size_t b = b0 + per * t;
            for (uint32_t sz : sizes[t]) { coff[b + 1] = coff[b] + sz; ++b; }
            if (!buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), f) != buf[t].size()) bad = true;
        }
    }
    if (!bad && fwrite(eof_block, 1, 28, f) != 28) bad = true;
    fclose(f);
    if (bad) return synth_error("cannot deflate / write '%s'", gz_path);
    const uint64_t c_end = coff[n_blocks];
    auto voff = [&](size_t text_off) -> uint64_t {
        if (text_off >= n) return c_end << 16;
        const size_t k = text_off / block_size;
        return (coff[k] << 16) | (uint64_t)(text_off - k * block_size);
    };
    auto reg2bin = [](int64_t beg, int64_t end) -> uint32_t {
        --end;
        if (beg >> 14 == end >> 14) return (uint32_t)(4681 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (uint32_t)(585 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (uint32_t)(73 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (uint32_t)(9 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (uint32_t)(1 + (beg >> 26));
        return 0;
    };
    struct Ref {
        std::string name;
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> lin;
        size_t first = 0, last = 0;
        uint64_t n_rec = 0;
    };
    std::vector<Ref> refs;
    std::map<std::string, size_t> ref_of;
    size_t pos = 0;
    Ref *cur = nullptr;
    while (pos < n) {
        const uint8_t *eol = static_cast<const uint8_t *>(memchr(text + pos, '\n', n - pos));
        const size_t len = eol ? (size_t)(eol - (text + pos)) : n - pos, next = pos + len + 1;
        if (len) {
            const uint8_t *p = text + pos, *e = p + len;
            const uint8_t *t1 = static_cast<const uint8_t *>(memchr(p, '\t', (size_t)(e - p)));
            if (!t1) return synth_error("'%s': a line without tabs at byte %zu", text_path, pos);
            if (!cur || cur->name.size() != (size_t)(t1 - p) || memcmp(cur->name.data(), p, (size_t)(t1 - p)) != 0) {
                const std::string name(reinterpret_cast<const char *>(p), (size_t)(t1 - p));
                auto it = ref_of.find(name);
                if (it == ref_of.end()) {
                    it = ref_of.emplace(name, refs.size()).first;
                    refs.emplace_back();
                    refs.back().name = name;
                    refs.back().first = pos;
                }
                cur = &refs[it->second];
            }
            int64_t beg = 0, end = 0;
            const uint8_t *q = t1 + 1;
            while (q < e && *q >= '0' && *q <= '9') beg = beg * 10 + (*q++ - '0');
            if (q < e && *q == '\t') ++q;
            while (q < e && *q >= '0' && *q <= '9') end = end * 10 + (*q++ - '0');
            if (end <= beg) end = beg + 1;
            const uint64_t v0 = voff(pos), v1 = voff(std::min(next, n));
            auto &chunks = cur->bins[reg2bin(beg, end)];
            if (!chunks.empty() && chunks.back().second == v0) chunks.back().second = v1;
            else chunks.emplace_back(v0, v1);
            for (int64_t w = beg >> 14; w <= (end - 1) >> 14; ++w) {
                if ((size_t)w >= cur->lin.size()) cur->lin.resize((size_t)w + 1, ~0ull);
                if (cur->lin[(size_t)w] == ~0ull) cur->lin[(size_t)w] = v0;
            }
            cur->last = std::min(next, n);
            cur->n_rec += 1;
        }
        pos = next;
    }
    std::string idx;
    auto put = [&](const void *p, size_t k) { idx.append(static_cast<const char *>(p), k); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    size_t l_nm = 0;
    for (const Ref &r : refs) l_nm += r.name.size() + 1;
    idx.append("TBI\1", 4);
    i32((int32_t)refs.size()); i32(0x10000); i32(1); i32(2); i32(3); i32('#'); i32(0); i32((int32_t)l_nm);
    for (const Ref &r : refs) idx.append(r.name.c_str(), r.name.size() + 1);
    for (const Ref &r : refs) {
        i32((int32_t)r.bins.size() + 1);
        for (const auto &kv : r.bins) {
            u32(kv.first);
            i32((int32_t)kv.second.size());
            for (const auto &c : kv.second) { u64(c.first); u64(c.second); }
        }
        u32(37450); i32(2); u64(voff(r.first)); u64(voff(r.last)); u64(r.n_rec); u64(0);
        i32((int32_t)r.lin.size());
        uint64_t last = 0;
        for (uint64_t v : r.lin) { if (v != ~0ull) last = v; u64(last); }
    }
    const std::string tbi_path = std::string(gz_path) + ".tbi";
    FILE *g = fopen(tbi_path.c_str(), "wb");
    if (!g) return synth_error("cannot create '%s'", tbi_path.c_str());
    std::string z;
    for (size_t o = 0; o < idx.size(); o += 0xFF00)
        if (!block(reinterpret_cast<const uint8_t *>(idx.data()) + o, std::min<size_t>(0xFF00, idx.size() - o), &z)) { fclose(g); return synth_error("cannot deflate the index"); }
    z.append(reinterpret_cast<const char *>(eof_block), 28);
    const bool ok = fwrite(z.data(), 1, z.size(), g) == z.size();
    fclose(g);
    return ok ? 0 : synth_error("short write to '%s'", tbi_path.c_str());
}

// ---- the same pair of files written from ROWS, part by part (bench `--extras cli1g`: a 1 Gbp pileup is 75 GB of text — it never
// exists as a file; each part is formatted, cut into blocks that continue the text stream, deflated on all threads and appended).
// nm_synth_bgz_open / _append_rows / _close produce byte for byte what nm_synth_write_bed + nm_synth_bgzip produce for the same
// rows (tests/test_bed_reader.py compares them).  One row per line with end = begin + 1, so a line's tabix bin is
// 4681 + (pos >> 14) and the index only needs the text offsets where (contig, pos >> 14) changes: the "runs" below.
struct nm_synth_bgz {
    FILE *f = nullptr, *tf = nullptr;
    std::string gz_path, text_path;
    uint32_t threads = 1, block_size = 0xFF00;
    int level = 6;
    std::string carry;                            // text not yet in a block (< block_size bytes)
    uint64_t text_total = 0, rows_total = 0;      // text bytes / rows appended so far (incl. the carry)
    std::vector<uint64_t> coff{0};                // compressed offset of every block written, + the end
    struct Run { std::string name; uint32_t window; uint64_t text_off, first_row; };
    std::vector<Run> runs;                        // maximal runs of lines with one (contig, pos >> 14)
};

static bool bgz_block(const uint8_t *src, size_t len, int level, std::string *out) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    std::vector<uint8_t> comp(deflateBound(&zs, (uLong)len) + 16);
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)len;
    zs.next_out = comp.data();
    zs.avail_out = (uInt)comp.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = comp.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
    const uint16_t bsize = (uint16_t)(clen + 25);
    const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 255), (uint8_t)(bsize >> 8)};
    out->append(reinterpret_cast<const char *>(hdr), 18);
    out->append(reinterpret_cast<const char *>(comp.data()), clen);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len), isize = (uint32_t)len;
    out->append(reinterpret_cast<const char *>(&crc), 4);
    out->append(reinterpret_cast<const char *>(&isize), 4);
    return true;
}

// text[0, n) as blocks of block_size (the last one may be shorter), deflated on all threads, written in order
static bool bgz_write_blocks(nm_synth_bgz *z, const uint8_t *text, size_t n) {
    const size_t bs = z->block_size, n_blocks = (n + bs - 1) / bs, per = 64;
    std::vector<std::string> buf(z->threads);
    std::vector<std::vector<uint32_t>> sizes(z->threads);
    bool bad = false;
    for (size_t b0 = 0; b0 < n_blocks && !bad; b0 += per * z->threads) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < z->threads; ++t)
            pool.emplace_back([&, t] {
                buf[t].clear();
                sizes[t].clear();
                for (size_t b = b0 + per * t; b < std::min(n_blocks, b0 + per * (t + 1)); ++b) {
                    const size_t before = buf[t].size();
                    if (!bgz_block(text + b * bs, std::min(bs, n - b * bs), z->level, &buf[t])) { bad = true; return; }
                    sizes[t].push_back((uint32_t)(buf[t].size() - before));
                }
            });
        for (auto &th : pool) th.join();
        for (unsigned t = 0; t < z->threads && !bad; ++t) {
            for (uint32_t sz : sizes[t]) z->coff.push_back(z->coff.back() + sz);
            if (!buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), z->f) != buf[t].size()) bad = true;
        }
    }
    return !bad;
}

// text_path (may be NULL): the plain-text twin of the stream is written there as well
int nm_synth_bgz_open(const char *gz_path, const char *text_path, uint32_t threads, int level, uint32_t block_size, nm_synth_bgz **out) {
    if (!gz_path || !out) return synth_error("NULL argument");
    *out = nullptr;
    nm_synth_bgz *z = new nm_synth_bgz();
    z->threads = threads ? threads : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    z->level = level;
    z->block_size = (block_size == 0 || block_size > 0xFF00) ? 0xFF00 : block_size;
    z->gz_path = gz_path;
    z->f = fopen(gz_path, "wb");
    if (!z->f) { delete z; return synth_error("cannot create '%s'", gz_path); }
    if (text_path) {
        z->text_path = text_path;
        z->tf = fopen(text_path, "wb");
        if (!z->tf) { fclose(z->f); delete z; return synth_error("cannot create '%s'", text_path); }
    }
    *out = z;
    return 0;
}

int nm_synth_bgz_append_rows(nm_synth_bgz *z, uint64_t n_rows, uint32_t n_contigs, const char *names, const uint32_t *name_offset,
                             const uint32_t *contig_id, const uint32_t *position, const int8_t *mod_type, const uint8_t *strand,
                             const int32_t *nvalid_cov, const int32_t *pct_hundredths) {
    if (!z || !z->f || (n_rows && (!names || !name_offset || !contig_id || !position || !mod_type || !strand || !nvalid_cov || !pct_hundredths)))
        return synth_error("NULL argument");
    static const char *codes[3] = {"m", "a", "21839"};
    const uint64_t piece = 1u << 19;                                   // rows per round and thread (~40 MB of text each)
    struct Mark { uint64_t row, local_off; };                          // a row that opens a run; where its line starts in the thread's text
    std::vector<std::string> buf(z->threads);
    std::vector<std::vector<Mark>> marks(z->threads);
    int bad = 0;
    for (uint64_t r0 = 0; r0 < n_rows && !bad; r0 += piece * z->threads) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < z->threads; ++t)
            pool.emplace_back([&, t] {
                std::string &o = buf[t];
                o.clear();
                marks[t].clear();
                const uint64_t a = std::min<uint64_t>(n_rows, r0 + piece * t), e = std::min<uint64_t>(n_rows, a + piece);
                char tmp[256];
                for (uint64_t i = a; i < e; ++i) {
                    const uint32_t c = contig_id[i];
                    const int m = mod_type[i];
                    if (c >= n_contigs || m < 0 || m > 2) { bad = 1; return; }
                    if (i == 0 || contig_id[i - 1] != c || (position[i - 1] >> 14) != (position[i] >> 14)) marks[t].push_back(Mark{i, o.size()});
                    const long long cov = nvalid_cov[i], pct = pct_hundredths[i], pos = position[i];
                    const long long nmod = (long long)std::nearbyint((double)(cov * pct) / 10000.0);    // Python's round(): half to even
                    o.append(names + name_offset[c], name_offset[c + 1] - name_offset[c]);
                    const int k = snprintf(tmp, sizeof tmp, "\t%lld\t%lld\t%s\t%lld\t%c\t%lld\t%lld\t255,0,0\t%lld\t%lld.%02lld\t%lld\t%lld\t0\t0\t0\t0\t0\n",
                                           pos, pos + 1, codes[m], cov, (char)strand[i], pos, pos + 1, cov, pct / 100, pct % 100, nmod, cov - nmod);
                    o.append(tmp, (size_t)k);
                }
            });
        for (auto &th : pool) th.join();
        if (bad) break;
        // the round's text behind the carry; the runs it opens; its whole blocks
        std::string text;
        size_t total = z->carry.size();
        for (unsigned t = 0; t < z->threads; ++t) total += buf[t].size();
        text.reserve(total);
        text.append(z->carry);
        uint64_t base = z->text_total;                                 // text offset of the next thread's first byte
        for (unsigned t = 0; t < z->threads; ++t) {
            for (const Mark &mk : marks[t]) {
                const uint32_t c = contig_id[mk.row];
                const size_t nlen = name_offset[c + 1] - name_offset[c];
                // the first row of a part may continue the run the part before it ended with
                if (mk.row == 0 && !z->runs.empty() && z->runs.back().window == (position[0] >> 14) && z->runs.back().name.size() == nlen &&
                    memcmp(z->runs.back().name.data(), names + name_offset[c], nlen) == 0)
                    continue;
                z->runs.push_back(nm_synth_bgz::Run{std::string(names + name_offset[c], nlen), position[mk.row] >> 14, base + mk.local_off,
                                                    z->rows_total + mk.row});
            }
            if (z->tf && !buf[t].empty() && fwrite(buf[t].data(), 1, buf[t].size(), z->tf) != buf[t].size()) bad = 2;
            text.append(buf[t]);
            base += buf[t].size();
        }
        z->text_total = base;
        const size_t whole = text.size() / z->block_size * z->block_size;
        if (whole && !bgz_write_blocks(z, reinterpret_cast<const uint8_t *>(text.data()), whole)) { bad = 3; break; }
        z->carry.assign(text, whole, std::string::npos);
    }
    z->rows_total += n_rows;
    if (bad == 1) return synth_error("row with a contig id / mod code outside the tables");
    if (bad == 2) return synth_error("short write to '%s'", z->text_path.c_str());
    if (bad == 3) return synth_error("cannot deflate / write '%s'", z->gz_path.c_str());
    return 0;
}

// the last (short) block, the EOF block, the tabix index; frees the handle
int nm_synth_bgz_close(nm_synth_bgz *z, uint64_t *text_bytes, uint64_t *gz_bytes) {
    if (!z) return 0;
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = true;
    if (!z->carry.empty()) ok = bgz_write_blocks(z, reinterpret_cast<const uint8_t *>(z->carry.data()), z->carry.size());
    if (ok && fwrite(eof_block, 1, 28, z->f) != 28) ok = false;
    fclose(z->f);
    if (z->tf) fclose(z->tf);
    const std::string gz_path = z->gz_path;
    const uint64_t n = z->text_total, c_end = z->coff.back(), bs = z->block_size;
    if (text_bytes) *text_bytes = n;
    if (gz_bytes) *gz_bytes = c_end + 28;
    auto voff = [&](uint64_t text_off) -> uint64_t {
        if (text_off >= n) return c_end << 16;
        const uint64_t k = text_off / bs;
        return (z->coff[k] << 16) | (text_off - k * bs);
    };
    struct Ref {
        std::string name;
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> lin;
        uint64_t first = 0, last = 0, n_rec = 0;
    };
    std::vector<Ref> refs;
    std::map<std::string, size_t> ref_of;
    for (size_t k = 0; ok && k < z->runs.size(); ++k) {
        const auto &run = z->runs[k];
        const uint64_t next_off = k + 1 < z->runs.size() ? z->runs[k + 1].text_off : n;
        const uint64_t rows = (k + 1 < z->runs.size() ? z->runs[k + 1].first_row : z->rows_total) - run.first_row;
        auto it = ref_of.find(run.name);
        if (it == ref_of.end()) {
            it = ref_of.emplace(run.name, refs.size()).first;
            refs.emplace_back();
            refs.back().name = run.name;
            refs.back().first = run.text_off;
        }
        Ref &r = refs[it->second];
        const uint64_t v0 = voff(run.text_off), v1 = voff(next_off);
        auto &chunks = r.bins[4681u + run.window];
        if (!chunks.empty() && chunks.back().second == v0) chunks.back().second = v1;
        else chunks.emplace_back(v0, v1);
        if (run.window >= r.lin.size()) r.lin.resize((size_t)run.window + 1, ~0ull);
        if (r.lin[run.window] == ~0ull) r.lin[run.window] = v0;
        r.last = next_off;
        r.n_rec += rows;
    }
    std::string idx;
    auto put = [&](const void *p, size_t k) { idx.append(static_cast<const char *>(p), k); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto u32 = [&](uint32_t v) { put(&v, 4); };
    auto u64 = [&](uint64_t v) { put(&v, 8); };
    size_t l_nm = 0;
    for (const Ref &r : refs) l_nm += r.name.size() + 1;
    idx.append("TBI\1", 4);
    i32((int32_t)refs.size()); i32(0x10000); i32(1); i32(2); i32(3); i32('#'); i32(0); i32((int32_t)l_nm);
    for (const Ref &r : refs) idx.append(r.name.c_str(), r.name.size() + 1);
    for (const Ref &r : refs) {
        i32((int32_t)r.bins.size() + 1);
        for (const auto &kv : r.bins) {
            u32(kv.first);
            i32((int32_t)kv.second.size());
            for (const auto &c : kv.second) { u64(c.first); u64(c.second); }
        }
        u32(37450); i32(2); u64(voff(r.first)); u64(voff(r.last)); u64(r.n_rec); u64(0);
        i32((int32_t)r.lin.size());
        uint64_t last = 0;
        for (uint64_t v : r.lin) { if (v != ~0ull) last = v; u64(last); }
    }
    const int level = z->level;
    delete z;
    if (!ok) return synth_error("cannot deflate / write '%s'", gz_path.c_str());
    const std::string tbi_path = gz_path + ".tbi";
    FILE *g = fopen(tbi_path.c_str(), "wb");
    if (!g) return synth_error("cannot create '%s'", tbi_path.c_str());
    std::string zz;
    for (size_t o = 0; o < idx.size(); o += 0xFF00)
        if (!bgz_block(reinterpret_cast<const uint8_t *>(idx.data()) + o, std::min<size_t>(0xFF00, idx.size() - o), level, &zz)) { fclose(g); return synth_error("cannot deflate the index"); }
    zz.append(reinterpret_cast<const char *>(eof_block), 28);
    const bool wrote = fwrite(zz.data(), 1, zz.size(), g) == zz.size();
    fclose(g);
    return wrote ? 0 : synth_error("short write to '%s'", tbi_path.c_str());
}


}  // extern "C"
