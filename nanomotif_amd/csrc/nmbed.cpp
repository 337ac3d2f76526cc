// nmbed — native modkit bedMethyl reader (plain text, gzip, bgzip) for libnmscan.  C ABI: include/nmscan.h.
//
// Replaces, on the ingestion side of the hot path, polars' scan_csv of the 18-column pileup
// (nanomotif/dataload.py:15-34, 72-100) and the epymetheus / pysam tabix reader of the bgzip path
// (dataload.py:102-152).  Only the six columns the reference keeps are materialised, as struct-of-arrays:
// contig id, start (col 2), mod code (col 4), strand (col 6), Nvalid_cov (col 10), percent modified (col 11) / 100.
//
// The file is inflated (BGZF blocks in parallel) or mapped, cut at line ends into one slab per thread, and every
// thread parses its slab into private columns that are concatenated at the end; contig names are interned per thread
// (rows of one contig are consecutive in a pileup, so the last name is cached) and merged in first-appearance order.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/nmscan.h"
#include "nmbed_parse.h"
#include "nmbgzf.h"

int nm_set_error(int code, const char *fmt, ...);   // defined in nmscan.hip

namespace {

// std::vector whose resize() leaves new elements uninitialised: the merged columns are sized once and then filled by
// all parser threads at the same time (no serial zero-fill of tens of gigabytes first)
template <class T>
struct NoInit {
    using value_type = T;
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t) { ::operator delete(p); }
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const NoInit<U> &) const { return true; }
    template <class U> bool operator!=(const NoInit<U> &) const { return false; }
};
template <class T> using Vec = std::vector<T, NoInit<T>>;

struct Columns {
    Vec<uint32_t> contig;
    Vec<int64_t> position, nvalid;
    Vec<int8_t> mod_type;
    Vec<uint8_t> strand;
    Vec<double> fraction;
    Vec<int32_t> n_mod, n_diff;                   // columns 12 / 17, only when the reader was opened with NM_BED_COUNTS
    bool want_counts = false;
    std::vector<std::string> names;               // local id -> name
    std::vector<std::string> other_mods;          // local mod id 3 + k -> code (anything but m, a, 21839)
    std::string error;
};

using namespace nmbedparse;

void parse_slab(const char *beg, const char *end, Columns *c) {
    std::unordered_map<std::string, uint32_t> ids;
    std::string last_name;
    uint32_t last_id = 0;
    bool have_last = false;
    size_t line_no = 0;
    const char *p = beg;
    while (p < end) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char *le = eol;
        if (le > p && le[-1] == '\r') --le;
        ++line_no;
        if (le > p) {
            // exactly 18 tab-separated columns, like the fixed schema the reference reads the file against (dataload.py:15-34): a
            // damaged file is refused, not half-read
            const char *f[18];
            const char *fe[18];
            int nf = 0;
            const char *q = p;
            for (;;) {
                const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
                if (nf < 18) { f[nf] = q; fe[nf] = t ? t : le; }
                ++nf;
                if (!t) break;
                q = t + 1;
            }
            if (nf != 18) { c->error = "pileup line that does not have exactly 18 tab-separated columns (modkit bedMethyl)"; return; }
            // contig
            const size_t nl = (size_t)(fe[0] - f[0]);
            if (!have_last || nl != last_name.size() || memcmp(last_name.data(), f[0], nl) != 0) {
                last_name.assign(f[0], nl);
                auto it = ids.find(last_name);
                if (it == ids.end()) {
                    it = ids.emplace(last_name, (uint32_t)c->names.size()).first;
                    c->names.push_back(last_name);
                }
                last_id = it->second;
                have_last = true;
            }
            int64_t pos = 0, cov = -1;
            double pct = 0;
            if (!parse_int(f[1], fe[1], &pos)) { c->error = "pileup column 2 (start) is not an integer"; return; }
            if (pos < 0) { c->error = "pileup column 2 (start) is negative"; return; }
            if (pos > 0xFFFFFFFEll) { c->error = "pileup position beyond 4 Gbp"; return; }      // (the engine's coordinates are 32-bit; the device parser says the same)
            // '+' or '-' and nothing else: the scoring path compares the column with exactly these (find_motifs_bin.py:1308-1314)
            if (fe[5] - f[5] != 1 || (f[5][0] != '+' && f[5][0] != '-')) { c->error = "pileup column 6 (strand) is neither '+' nor '-'"; return; }
            if (!is_null(f[9], fe[9]) && !parse_int(f[9], fe[9], &cov)) { c->error = "pileup column 10 (Nvalid_cov) is not an integer"; return; }
            bool pct_null = is_null(f[10], fe[10]);
            if (!pct_null && !parse_double(f[10], fe[10], &pct)) { c->error = "pileup column 11 (percent modified) is not a number"; return; }
            const size_t ml = (size_t)(fe[3] - f[3]);
            int8_t mt = -1;
            if (ml == 1 && f[3][0] == 'm') mt = 0;
            else if (ml == 1 && f[3][0] == 'a') mt = 1;
            else if (ml == 5 && memcmp(f[3], "21839", 5) == 0) mt = 2;
            else {
                size_t k = 0;
                for (; k < c->other_mods.size(); ++k)
                    if (c->other_mods[k].size() == ml && memcmp(c->other_mods[k].data(), f[3], ml) == 0) break;
                if (k == c->other_mods.size()) c->other_mods.emplace_back(f[3], ml);
                if (k > 100) { c->error = "more than 100 distinct modification codes in column 4"; return; }
                mt = (int8_t)(3 + k);
            }
            c->contig.push_back(last_id);
            c->position.push_back(pos);
            c->mod_type.push_back(mt);
            c->strand.push_back((uint8_t)f[5][0]);
            c->nvalid.push_back(cov);
            c->fraction.push_back(pct_null ? -1.0 : pct / 100.0);     // dataload.py:85
            if (c->want_counts) {
                int64_t nm = 0, nd = 0;
                if (!parse_int(f[11], fe[11], &nm) || !parse_int(f[16], fe[16], &nd) || nm < 0 || nd < 0 || nm > 0x7FFFFFFF || nd > 0x7FFFFFFF) {
                    c->error = "pileup column 12 (N_mod) or 17 (N_diff) is not a non-negative integer";
                    return;
                }
                c->n_mod.push_back((int32_t)nm);
                c->n_diff.push_back((int32_t)nd);
            }
        }
        p = eol + 1;
    }
}

// ---- input: plain (mmap), BGZF (parallel inflate), generic gzip (serial)
struct Buffer {
    const char *data = nullptr;
    size_t size = 0;
    void *map = nullptr;
    size_t map_size = 0;
    std::vector<char> owned;
};

using nmbgzf::inflate_raw;

bool load_gzip(const uint8_t *z, size_t zn, Buffer *b, unsigned threads, std::string *err) {
    // BGZF: every member is a gzip block with FEXTRA 'B','C' subfield holding the block size (nmbgzf.h)
    std::vector<nmbgzf::Piece> blocks;
    uint64_t out = 0;
    if (nmbgzf::whole_file(z, zn, &blocks, &out)) {
        b->owned.resize(out);
        std::vector<std::thread> pool;
        std::vector<int> okv(threads, 1);
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                for (size_t i = t; i < blocks.size(); i += threads)
                    if (!inflate_raw(z + blocks[i].in_off, blocks[i].in_len, b->owned.data() + blocks[i].text_off, blocks[i].out_len, blocks[i].crc)) okv[t] = 0;
            });
        for (auto &th : pool) th.join();
        for (int v : okv)
            if (!v) { *err = "corrupt BGZF block (deflate stream, size or CRC-32)"; return false; }
        b->data = b->owned.data();
        b->size = b->owned.size();
        return true;
    }
    // generic (possibly multi-member) gzip
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) { *err = "zlib init failed"; return false; }
    zs.next_in = const_cast<Bytef *>(z);
    zs.avail_in = (uInt)std::min<size_t>(zn, 0xFFFFFFFFu);
    size_t consumed_total = 0;
    std::vector<char> &o = b->owned;
    o.resize(std::max<size_t>(zn * 4, 1 << 20));
    size_t w = 0;
    for (;;) {
        if (w == o.size()) o.resize(o.size() * 2);
        zs.next_out = reinterpret_cast<Bytef *>(o.data() + w);
        zs.avail_out = (uInt)std::min<size_t>(o.size() - w, 0x40000000u);
        const uInt before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        w += before - zs.avail_out;
        if (rc == Z_STREAM_END) {
            consumed_total = zn - zs.avail_in;
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); *err = "zlib reset failed"; return false; }
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); *err = "corrupt gzip stream"; return false; }
        if (rc == Z_BUF_ERROR && zs.avail_in == 0) { inflateEnd(&zs); *err = "truncated gzip stream"; return false; }
    }
    (void)consumed_total;
    inflateEnd(&zs);
    o.resize(w);
    b->data = o.data();
    b->size = w;
    return true;
}

// map a file and, when it is gzip / BGZF, inflate it; *buf owns whatever it needs
static int load_file(const char *path, unsigned threads, Buffer *buf, const char *what) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nm_set_error(NM_EINVAL, "cannot open %s '%s'", what, path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return nm_set_error(NM_EINVAL, "cannot stat %s '%s'", what, path); }
    if (st.st_size > 0) {
        buf->map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        buf->map_size = (size_t)st.st_size;
        if (buf->map == MAP_FAILED) { buf->map = nullptr; close(fd); return nm_set_error(NM_EINVAL, "cannot map %s '%s'", what, path); }
    }
    close(fd);
    const uint8_t *raw = static_cast<const uint8_t *>(buf->map);
    std::string err;
    if (buf->map_size >= 2 && raw[0] == 31 && raw[1] == 139) {
        if (!load_gzip(raw, buf->map_size, buf, threads, &err)) {
            munmap(buf->map, buf->map_size);
            buf->map = nullptr;
            return nm_set_error(NM_EINVAL, "%s: %s", path, err.c_str());
        }
    } else {
        buf->data = static_cast<const char *>(buf->map);
        buf->size = buf->map_size;
    }
    return NM_OK;
}

// map a file as it is (no inflation)
static int load_file_raw(const char *path, Buffer *buf) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nm_set_error(NM_EINVAL, "cannot open '%s'", path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return nm_set_error(NM_EINVAL, "cannot stat '%s'", path); }
    if (st.st_size > 0) {
        buf->map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        buf->map_size = (size_t)st.st_size;
        if (buf->map == MAP_FAILED) { buf->map = nullptr; close(fd); return nm_set_error(NM_EINVAL, "cannot map '%s'", path); }
    }
    close(fd);
    return NM_OK;
}

}  // namespace

struct nm_bed {
    Columns all;
    std::vector<const char *> name_ptrs;
    // the columns in the exact types nm_ingest_pileup takes (nm_bed_ingest_columns)
    Vec<uint32_t> ing_contig, ing_position;
    Vec<int32_t> ing_nvalid;
};

extern "C" {

// bedMethyl text (whole file or the tabix-selected regions of it) -> columns
static int parse_text(const char *path, Buffer &buf, uint32_t threads, nm_bed **out, bool want_counts = false) {
    // slabs cut at line ends
    std::vector<size_t> cut(1, 0);
    for (unsigned t = 1; t < threads; ++t) {
        size_t at = buf.size / threads * t;
        if (at <= cut.back()) continue;
        const char *nl = (const char *)memchr(buf.data + at, '\n', buf.size - at);
        if (!nl) break;
        cut.push_back((size_t)(nl - buf.data) + 1);
    }
    cut.push_back(buf.size);
    std::vector<Columns> parts(cut.size() - 1);
    for (auto &p : parts) p.want_counts = want_counts;
    std::vector<std::thread> pool;
    for (size_t i = 0; i + 1 < cut.size(); ++i)
        pool.emplace_back(parse_slab, buf.data + cut[i], buf.data + cut[i + 1], &parts[i]);
    for (auto &th : pool) th.join();
    if (buf.map) munmap(buf.map, buf.map_size);
    buf.map = nullptr;
    for (auto &p : parts)
        if (!p.error.empty()) return nm_set_error(NM_EINVAL, "%s: %s", path, p.error.c_str());
    nm_bed *b = new (std::nothrow) nm_bed();
    if (!b) return nm_set_error(NM_ENOMEM, "out of host memory");
    Columns &a = b->all;
    // serial and cheap: global contig / mod ids of every part, and where its rows go; then all parts copy at once
    std::vector<size_t> at(parts.size() + 1, 0);
    std::vector<std::vector<uint32_t>> remap(parts.size());
    std::vector<std::vector<int8_t>> mod_remap(parts.size());
    std::unordered_map<std::string, uint32_t> ids;
    for (size_t k = 0; k < parts.size(); ++k) {
        Columns &p = parts[k];
        at[k + 1] = at[k] + p.position.size();
        remap[k].resize(p.names.size());
        for (size_t i = 0; i < p.names.size(); ++i) {
            auto it = ids.find(p.names[i]);
            if (it == ids.end()) {
                it = ids.emplace(p.names[i], (uint32_t)a.names.size()).first;
                a.names.push_back(p.names[i]);
            }
            remap[k][i] = it->second;
        }
        mod_remap[k].resize(3 + p.other_mods.size());
        for (int j = 0; j < 3; ++j) mod_remap[k][j] = (int8_t)j;
        for (size_t j = 0; j < p.other_mods.size(); ++j) {
            size_t g = 0;
            for (; g < a.other_mods.size(); ++g)
                if (a.other_mods[g] == p.other_mods[j]) break;
            if (g == a.other_mods.size()) a.other_mods.push_back(p.other_mods[j]);
            mod_remap[k][3 + j] = (int8_t)(3 + g);
        }
    }
    const size_t n = at.back();
    a.contig.resize(n); a.position.resize(n); a.nvalid.resize(n); a.mod_type.resize(n); a.strand.resize(n); a.fraction.resize(n);
    a.want_counts = want_counts;
    if (want_counts) { a.n_mod.resize(n); a.n_diff.resize(n); }
    {
        std::vector<std::thread> copiers;
        for (size_t k = 0; k < parts.size(); ++k)
            copiers.emplace_back([&, k] {
                Columns &p = parts[k];
                const size_t o = at[k], m = p.position.size();
                for (size_t i = 0; i < m; ++i) a.contig[o + i] = remap[k][p.contig[i]];
                for (size_t i = 0; i < m; ++i) a.mod_type[o + i] = mod_remap[k][(size_t)p.mod_type[i]];
                if (m) {
                    memcpy(a.position.data() + o, p.position.data(), m * sizeof(int64_t));
                    memcpy(a.nvalid.data() + o, p.nvalid.data(), m * sizeof(int64_t));
                    memcpy(a.strand.data() + o, p.strand.data(), m);
                    memcpy(a.fraction.data() + o, p.fraction.data(), m * sizeof(double));
                    if (a.want_counts) {
                        memcpy(a.n_mod.data() + o, p.n_mod.data(), m * sizeof(int32_t));
                        memcpy(a.n_diff.data() + o, p.n_diff.data(), m * sizeof(int32_t));
                    }
                }
                Columns().contig.swap(p.contig);
                Columns().position.swap(p.position);
                Columns().nvalid.swap(p.nvalid);
                Columns().fraction.swap(p.fraction);
            });
        for (auto &th : copiers) th.join();
    }
    for (auto &s : a.names) b->name_ptrs.push_back(s.c_str());
    *out = b;
    return NM_OK;
}

static int bed_open_impl(const char *path, uint32_t threads, bool want_counts, nm_bed **out) {
    if (!path || !out) return nm_set_error(NM_EINVAL, "NULL argument");
    *out = nullptr;
    if (threads == 0) threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    Buffer buf;
    const int rc = load_file(path, threads, &buf, "pileup");
    if (rc) return rc;
    return parse_text(path, buf, threads, out, want_counts);
}

int nm_bed_open(const char *path, uint32_t threads, nm_bed **out) { return bed_open_impl(path, threads, false, out); }

int nm_bed_open_counts(const char *path, uint32_t threads, nm_bed **out) { return bed_open_impl(path, threads, true, out); }

int nm_bed_count_columns(nm_bed *b, const int32_t **n_modified, const int32_t **n_diff) {
    if (!b || !n_modified || !n_diff) return nm_set_error(NM_EINVAL, "NULL argument");
    if (!b->all.want_counts) return nm_set_error(NM_ESTATE, "the pileup was not opened with nm_bed_open_counts");
    *n_modified = b->all.n_mod.data();
    *n_diff = b->all.n_diff.data();
    return NM_OK;
}

// Tabix-indexed read (dataload.py:102-152 / find_motifs_bin.py:233-246: the reference fetches the records of a bin's
// contigs through the .tbi index instead of reading the file): only the BGZF blocks that hold the wanted contigs are
// inflated and parsed.  The index (tabix format: magic TBI\1, per reference the bins with their chunks of virtual
// offsets) gives every contig its [begin, end) virtual offsets — from the pseudo-bin 37450 when present, else the
// hull of its chunks.  NM_EINDEX when the .tbi is not one or does not fit the file (the caller may read the whole file).
int nm_bed_open_indexed(const char *path, const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset,
                        uint32_t threads, nm_bed **out, uint64_t stats[4]) {
    if (!path || !tbi_path || !out || (n_contigs && (!names || !name_offset))) return nm_set_error(NM_EINVAL, "NULL argument");
    *out = nullptr;
    if (threads == 0) threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    Buffer idx;
    int rc = load_file(tbi_path, threads, &idx, "tabix index");
    if (rc) {                       // an index that cannot be read or inflated is an index problem, not a pileup problem
        const std::string why = nm_last_error();
        return nm_set_error(NM_EINDEX, "%s", why.c_str());
    }
    struct Unmap {
        Buffer &b;
        ~Unmap() { if (b.map) munmap(b.map, b.map_size); b.map = nullptr; }
    } unmap_idx{idx};
    std::unordered_map<std::string, uint32_t> want;
    for (uint32_t i = 0; i < n_contigs; ++i) want.emplace(std::string(names + name_offset[i], name_offset[i + 1] - name_offset[i]), i);
    std::vector<nmbgzf::Region> merged;
    std::vector<uint64_t> block_starts;
    uint64_t found_in_index = 0;
    {
        const std::string what = nmbgzf::tabix_regions(reinterpret_cast<const uint8_t *>(idx.data), idx.size, want, &merged, &found_in_index, &block_starts);
        if (!what.empty()) return nm_set_error(NM_EINDEX, "%s: %s", tbi_path, what.c_str());
    }
    Buffer file;
    rc = load_file_raw(path, &file);
    if (rc) return rc;
    Unmap unmap_file{file};
    const uint8_t *z = static_cast<const uint8_t *>(file.map);
    const size_t zn = file.map_size;
    // the blocks of every region, where their text goes, and which part of it is wanted
    std::vector<nmbgzf::Piece> blocks;
    uint64_t text_size = 0, inflated = 0;
    {
        bool index_problem = false;
        const std::string what = nmbgzf::region_pieces(z, zn, merged, &blocks, &text_size, &inflated, &block_starts, threads, -1, &index_problem);
        if (!what.empty()) return nm_set_error(index_problem ? NM_EINDEX : NM_EINVAL, "%s: %s", path, what.c_str());
    }
    Buffer text;
    text.owned.resize(text_size);
    {
        std::vector<std::thread> pool;
        std::vector<int> okv(threads, 1);
        for (unsigned w = 0; w < threads; ++w)
            pool.emplace_back([&, w] {
                std::vector<char> tmp;
                for (size_t i = w; i < blocks.size(); i += threads)
                    if (!nmbgzf::inflate_piece(z, blocks[i], text.owned.data() + blocks[i].text_off, tmp)) okv[w] = 0;
            });
        for (auto &th : pool) th.join();
        for (int v : okv)
            if (!v) return nm_set_error(NM_EINVAL, "%s: corrupt BGZF block (deflate stream, size or CRC-32)", path);
    }
    text.data = text.owned.data();
    text.size = text.owned.size();
    if (stats) { stats[0] = inflated; stats[1] = zn; stats[2] = n_contigs - std::min<uint64_t>(found_in_index, n_contigs); stats[3] = 0; }
    rc = parse_text(path, text, threads, out);
    if (rc == NM_EINVAL) {          // blocks intact, lines that do not parse: a region that starts inside a line — the index is stale
        const std::string why = nm_last_error();
        return nm_set_error(NM_EINDEX, "%s: the text the tabix index names does not parse (%s): stale .tbi?", path, why.c_str());
    }
    if (rc) return rc;
    // the text the index pointed at must belong to the contigs that were asked for: a stale or foreign .tbi otherwise
    // yields a silently wrong subset of rows
    for (const std::string &nm : (*out)->all.names)
        if (!want.count(nm)) {
            const std::string culprit = nm;
            delete *out;
            *out = nullptr;
            return nm_set_error(NM_EINDEX, "%s: the tabix index does not match the pileup (rows of contig '%s' where another contig was indexed): stale .tbi?",
                                path, culprit.c_str());
        }
    return NM_OK;
}

// The [begin, end) virtual offsets a tabix index holds for the named sequences (host only, no pileup needed): what the reference's
// reader asks pysam for before it fetches a bin's contigs (dataload.py:102-152).  The one htslib-made index in the reference tree
// (datasets/geobacillus-plasmids.pileup.bed.gz.tbi) is read through this call in tests/test_bed_reader.py.
int nm_tabix_regions(const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset, uint64_t *begin, uint64_t *end,
                     uint8_t *present) {
    if (!tbi_path || !begin || !end || !present || (n_contigs && (!names || !name_offset))) return nm_set_error(NM_EINVAL, "NULL argument");
    Buffer idx;
    int rc = load_file(tbi_path, 1, &idx, "tabix index");
    if (rc) {
        const std::string why = nm_last_error();
        return nm_set_error(NM_EINDEX, "%s", why.c_str());
    }
    struct Unmap {
        Buffer &b;
        ~Unmap() { if (b.map) munmap(b.map, b.map_size); b.map = nullptr; }
    } unmap_idx{idx};
    std::unordered_map<std::string, uint32_t> want;
    for (uint32_t i = 0; i < n_contigs; ++i) {
        begin[i] = end[i] = 0;
        present[i] = 0;
        // a name given twice answers at its first position only
        want.emplace(std::string(names + name_offset[i], name_offset[i + 1] - name_offset[i]), i);
    }
    std::vector<nmbgzf::Region> merged;
    uint64_t found = 0;
    // tabix_regions sizes its per-name tables by want.size(): number the distinct names densely and map back
    std::vector<nmbgzf::Region> per_w;
    std::vector<uint8_t> have_w;
    std::unordered_map<std::string, uint32_t> dense;
    std::vector<uint32_t> back;
    for (const auto &kv : want) { dense.emplace(kv.first, (uint32_t)back.size()); back.push_back(kv.second); }
    const std::string what = nmbgzf::tabix_regions(reinterpret_cast<const uint8_t *>(idx.data), idx.size, dense, &merged, &found, nullptr, &per_w, &have_w);
    if (!what.empty()) return nm_set_error(NM_EINDEX, "%s: %s", tbi_path, what.c_str());
    for (size_t k = 0; k < back.size(); ++k) {
        begin[back[k]] = per_w[k].beg;
        end[back[k]] = per_w[k].end;
        present[back[k]] = have_w[k];
    }
    return NM_OK;
}

int nm_bed_shape(nm_bed *b, uint64_t *n_rows, uint32_t *n_contigs) {
    if (!b || !n_rows || !n_contigs) return nm_set_error(NM_EINVAL, "NULL argument");
    *n_rows = b->all.position.size();
    *n_contigs = (uint32_t)b->all.names.size();
    return NM_OK;
}

int nm_bed_contig_name(nm_bed *b, uint32_t i, const char **name) {
    if (!b || !name || i >= b->name_ptrs.size()) return nm_set_error(NM_EINVAL, "bad contig index");
    *name = b->name_ptrs[i];
    return NM_OK;
}

int nm_bed_mod_code(nm_bed *b, uint32_t id, const char **code) {
    static const char *known[3] = {"m", "a", "21839"};
    if (!b || !code) return nm_set_error(NM_EINVAL, "NULL argument");
    if (id < 3) { *code = known[id]; return NM_OK; }
    if (id - 3 >= b->all.other_mods.size()) return nm_set_error(NM_EINVAL, "mod id %u not present", id);
    *code = b->all.other_mods[id - 3].c_str();
    return NM_OK;
}

int nm_bed_columns(nm_bed *b, const uint32_t **contig_id, const int64_t **position, const int8_t **mod_type,
                   const uint8_t **strand, const double **fraction_mod, const int64_t **nvalid_cov) {
    if (!b) return nm_set_error(NM_EINVAL, "NULL argument");
    if (contig_id) *contig_id = b->all.contig.data();
    if (position) *position = b->all.position.data();
    if (mod_type) *mod_type = b->all.mod_type.data();
    if (strand) *strand = b->all.strand.data();
    if (fraction_mod) *fraction_mod = b->all.fraction.data();
    if (nvalid_cov) *nvalid_cov = b->all.nvalid.data();
    return NM_OK;
}

int nm_bed_ingest_columns(nm_bed *b, const uint32_t *contig_lut, uint32_t n_lut, const uint32_t **contig_id,
                          const uint32_t **position, const int8_t **mod_type, const uint8_t **strand,
                          const double **fraction_mod, const int32_t **nvalid_cov) {
    if (!b || !contig_lut || !contig_id || !position || !mod_type || !strand || !fraction_mod || !nvalid_cov)
        return nm_set_error(NM_EINVAL, "NULL argument");
    if (n_lut != b->all.names.size()) return nm_set_error(NM_EINVAL, "contig_lut has %u entries, the pileup names %zu contigs", n_lut, b->all.names.size());
    const size_t n = b->all.contig.size();
    // a second call (another lut) only maps the contigs again: the 64-bit originals were given up by the first
    const bool again = n > 0 && b->all.position.size() != n && b->ing_position.size() == n;
    b->ing_contig.resize(n);
    if (again) {
        for (size_t i = 0; i < n; ++i) b->ing_contig[i] = contig_lut[b->all.contig[i]];
        *contig_id = b->ing_contig.data();
        *position = b->ing_position.data();
        *mod_type = b->all.mod_type.data();
        *strand = b->all.strand.data();
        *fraction_mod = b->all.fraction.data();
        *nvalid_cov = b->ing_nvalid.data();
        return NM_OK;
    }
    b->ing_position.resize(n);
    b->ing_nvalid.resize(n);
    unsigned threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < (1u << 20)) threads = 1;
    std::vector<int> bad(threads, 0);
    auto work = [&](unsigned t) {
        const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
        for (size_t i = lo; i < hi; ++i) {
            const int64_t p = b->all.position[i], v = b->all.nvalid[i];
            const int8_t m = b->all.mod_type[i];
            if (p < 0 || p > 0xFFFFFFFEll) bad[t] |= 1;
            if (m < 0) bad[t] |= 2;
            b->ing_contig[i] = contig_lut[b->all.contig[i]];
            b->ing_position[i] = (uint32_t)p;
            // a null coverage can never pass Nvalid_cov > 5: such rows leave through the coverage filter; a null PERCENTAGE
            // with enough coverage keeps its coverage and its fraction of -1: it counts as a position of its
            // (contig, mod code) group in the frequency filter (pl.count(), dataload.py:216) and leaves at the adjacency filter
            b->ing_nvalid[i] = v < 0 ? -1 : (int32_t)std::min<int64_t>(v, 0x7FFFFFFF);
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    int any = 0;
    for (int x : bad) any |= x;
    if (any & 1) return nm_set_error(NM_ERANGE, "pileup position beyond 4 Gbp");
    if (any & 2) return nm_set_error(NM_ERANGE, "modification code id out of range");
    Vec<int64_t>().swap(b->all.position);          // the 64-bit originals are no longer needed
    Vec<int64_t>().swap(b->all.nvalid);
    *contig_id = b->ing_contig.data();
    *position = b->ing_position.data();
    *mod_type = b->all.mod_type.data();
    *strand = b->all.strand.data();
    *fraction_mod = b->all.fraction.data();
    *nvalid_cov = b->ing_nvalid.data();
    return NM_OK;
}

int nm_bed_close(nm_bed *b) {
    delete b;
    return NM_OK;
}

// ------------------------------------------------------------------------------------------------------
// FASTA reader (reference: nanomotif/fasta.py:35-49 + DNAsequence checks, seq.py:53-71): names = first whitespace token
// of the header, sequences upper-cased and validated against the IUPAC alphabet, all records back to back.
// ------------------------------------------------------------------------------------------------------
struct nm_fasta {
    std::vector<std::string> names;
    std::vector<uint64_t> offset;         // n + 1
    std::vector<uint8_t> seq;
};

int nm_fasta_open(const char *path, uint32_t threads, nm_fasta **out) {
    if (!path || !out) return nm_set_error(NM_EINVAL, "NULL argument");
    *out = nullptr;
    if (threads == 0) threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    Buffer buf;
    int rc = load_file(path, threads, &buf, "assembly");
    if (rc) return rc;
    auto release = [&]() { if (buf.map) munmap(buf.map, buf.map_size); };
    const char *d = buf.data, *end = buf.data + buf.size;
    // record starts: '>' at the beginning of a line
    struct Rec { const char *hdr, *body, *stop; uint64_t len; };
    std::vector<Rec> recs;
    for (const char *p = d; p < end;) {
        const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        if (*p == '>') {
            if (!recs.empty()) recs.back().stop = p;
            recs.push_back(Rec{p, eol < end ? eol + 1 : end, end, 0});
        }
        p = eol < end ? eol + 1 : end;
    }
    nm_fasta *f = new (std::nothrow) nm_fasta();
    if (!f) { release(); return nm_set_error(NM_ENOMEM, "out of host memory"); }
    static bool iupac[256];
    static bool init = false;
    if (!init) { for (const char *c = "ATGCRYSWKMBDHVN"; *c; ++c) iupac[(unsigned char)*c] = true; init = true; }
    const size_t n = recs.size();
    threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n));
    auto for_records = [&](auto fn) {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] { for (size_t i = t; i < n; i += threads) fn(i); });
        for (auto &th : pool) th.join();
    };
    for_records([&](size_t i) {                                   // pass 1: sequence lengths (bytes that are not \r \n)
        uint64_t len = 0;
        for (const char *p = recs[i].body; p < recs[i].stop; ++p) len += (*p != '\n' && *p != '\r');
        recs[i].len = len;
    });
    f->offset.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) f->offset[i + 1] = f->offset[i] + recs[i].len;
    f->seq.resize(f->offset[n]);
    std::vector<int> bad(n, 0);
    for_records([&](size_t i) {                                   // pass 2: copy, upper-case, validate
        uint8_t *o = f->seq.data() + f->offset[i];
        int b = 0;
        for (const char *p = recs[i].body; p < recs[i].stop; ++p) {
            unsigned char c = (unsigned char)*p;
            if (c == '\n' || c == '\r') continue;
            if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
            b |= !iupac[c];
            *o++ = c;
        }
        bad[i] = b;
    });
    for (size_t i = 0; i < n; ++i) {
        const char *h = recs[i].hdr + 1, *he = static_cast<const char *>(memchr(h, '\n', (size_t)(end - h)));
        if (!he) he = end;
        while (h < he && (*h == ' ' || *h == '\t' || *h == '\r' || *h == '\f' || *h == '\v')) ++h;      // str.split(): leading whitespace
        const char *t = h;
        while (t < he && !(*t == ' ' || *t == '\t' || *t == '\r' || *t == '\f' || *t == '\v')) ++t;
        f->names.emplace_back(h, (size_t)(t - h));
        if (recs[i].len == 0 || bad[i]) {
            const std::string name = f->names.back();            // copied before the file mapping goes away
            const bool empty = recs[i].len == 0;
            delete f;
            release();
            return empty ? nm_set_error(NM_ESEQUENCE, "DNA sequence must not be empty (record '%s')", name.c_str())
                         : nm_set_error(NM_ESEQUENCE, "DNA sequence must be a nucleotide sequence of ATGCRYSWKMBDHVN (record '%s')", name.c_str());
        }
    }
    release();
    *out = f;
    return NM_OK;
}

int nm_fasta_shape(nm_fasta *f, uint32_t *n_records, uint64_t *total_bp) {
    if (!f || !n_records || !total_bp) return nm_set_error(NM_EINVAL, "NULL argument");
    *n_records = (uint32_t)f->names.size();
    *total_bp = f->offset.back();
    return NM_OK;
}

int nm_fasta_record(nm_fasta *f, uint32_t i, const char **name, uint64_t *offset, uint64_t *length) {
    if (!f || !name || !offset || !length || i >= f->names.size()) return nm_set_error(NM_EINVAL, "bad record index");
    *name = f->names[i].c_str();
    *offset = f->offset[i];
    *length = f->offset[i + 1] - f->offset[i];
    return NM_OK;
}

int nm_fasta_sequence(nm_fasta *f, const uint8_t **seq_upper) {
    if (!f || !seq_upper) return nm_set_error(NM_EINVAL, "NULL argument");
    *seq_upper = f->seq.data();
    return NM_OK;
}

int nm_fasta_close(nm_fasta *f) {
    delete f;
    return NM_OK;
}

}  // extern "C"