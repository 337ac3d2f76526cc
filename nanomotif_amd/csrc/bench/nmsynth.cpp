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

}  // extern "C"
