// BGZF blocks and tabix indexes, shared by the host bedMethyl reader (nmbed.cpp) and the device-side parser
// (nmbedgpu.hip).  Internal: not part of the C ABI.
//
// A bgzip file is a sequence of gzip members ("blocks", <= 64 KiB of text each) whose FEXTRA 'B','C' subfield holds the
// block size; a tabix index (.tbi) gives, per reference sequence, virtual offsets (file offset of a block << 16 | offset
// inside its text) — the reference reads a bin's contigs through it (dataload.py:102-152, find_motifs_bin.py:233-246).
// Here the text that has to be parsed is described as a list of PIECES: a block and the part of its text that is wanted.
#pragma once
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace nmbgzf {

struct Piece {
    size_t in_off, in_len;      // the raw deflate stream of the block inside the file
    size_t out_len;             // text bytes of the whole block
    uint32_t skip, take;        // the wanted part of that text
    uint64_t text_off;          // where the wanted part starts in the concatenated text
    uint32_t crc;               // CRC-32 of the block's whole text (the member's trailer)
};

// one block's deflate stream -> dst; the text must have the size AND the CRC-32 the member's trailer states (gzip readers —
// Python's gzip, htslib — refuse a block whose checksum is off; so does this one)
inline bool inflate_raw(const uint8_t *src, size_t n, char *dst, size_t dst_n, uint32_t crc) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)n;
    zs.next_out = reinterpret_cast<Bytef *>(dst);
    zs.avail_out = (uInt)dst_n;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0 && (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(dst), (uInt)dst_n) == crc;
}

// header of the block at `off`: its size in the file, where its deflate stream sits, its text size.  false: not a BGZF block
// (or one whose trailer states more than the format's 64 KiB of text)
inline bool block_at(const uint8_t *z, size_t zn, size_t off, size_t *bsize, size_t *in_off, size_t *in_len, size_t *isize, uint32_t *crc) {
    if (off + 18 > zn) return false;
    if (!(z[off] == 31 && z[off + 1] == 139 && z[off + 2] == 8 && (z[off + 3] & 4))) return false;
    const size_t xlen = z[off + 10] | (z[off + 11] << 8);
    size_t x = off + 12, xe = x + xlen, bs = 0;
    if (xe > zn) return false;
    while (x + 4 <= xe) {
        const size_t slen = z[x + 2] | (z[x + 3] << 8);
        if (z[x] == 'B' && z[x + 1] == 'C' && slen == 2 && x + 6 <= xe) bs = (size_t)(z[x + 4] | (z[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bs < 12 + xlen + 8 || off + bs > zn) return false;
    *bsize = bs;
    *in_off = off + 12 + xlen;
    *in_len = bs - 12 - xlen - 8;
    *isize = z[off + bs - 4] | (z[off + bs - 3] << 8) | (z[off + bs - 2] << 16) | ((size_t)z[off + bs - 1] << 24);
    *crc = (uint32_t)z[off + bs - 8] | ((uint32_t)z[off + bs - 7] << 8) | ((uint32_t)z[off + bs - 6] << 16) | ((uint32_t)z[off + bs - 5] << 24);
    // a BGZF block holds at most 64 KiB of text (SAM spec 4.1): the device inflate sizes its per-block scratch and the
    // padding behind the compressed bytes for exactly that, so a trailer that claims more is a damaged file, not a big block
    if (*isize > 65536) return false;
    return true;
}

// the same through pread (fd >= 0): walking a mapped file touches one or two fresh pages per block — a page fault each, 2.3 million of
// them for the pileup of a 1 Gbp metagenome —, two small reads per block cost a third of that
inline bool block_at_fd(int fd, size_t zn, size_t off, size_t *bsize, size_t *in_off, size_t *in_len, size_t *isize, uint32_t *crc, bool *is_bgzf) {
    uint8_t h[64];
    *is_bgzf = false;
    if (off + 18 > zn) return false;
    const size_t hn = std::min<size_t>(sizeof h, zn - off);
    if (pread(fd, h, hn, (off_t)off) != (ssize_t)hn) return false;
    if (!(h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4))) return false;
    *is_bgzf = true;
    const size_t xlen = h[10] | (h[11] << 8);
    if (12 + xlen > hn) return false;                       // (extra fields beyond 52 bytes: not what bgzip writes)
    size_t x = 12, xe = 12 + xlen, bs = 0;
    while (x + 4 <= xe) {
        const size_t slen = h[x + 2] | (h[x + 3] << 8);
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= xe) bs = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bs < 12 + xlen + 8 || off + bs > zn) return false;
    uint8_t t[8];
    if (pread(fd, t, 8, (off_t)(off + bs - 8)) != 8) return false;
    *bsize = bs;
    *in_off = off + 12 + xlen;
    *in_len = bs - 12 - xlen - 8;
    *crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    *isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((size_t)t[7] << 24);
    return *isize <= 65536;
}

// every block of the file, whole.  false when the file is not BGZF from the first to the last byte (a plain gzip stream)
inline bool whole_file(const uint8_t *z, size_t zn, std::vector<Piece> *pieces, uint64_t *text_size) {
    pieces->clear();
    size_t off = 0;
    uint64_t out = 0;
    while (off < zn) {
        size_t bsize, in_off, in_len, isize;
        uint32_t crc;
        if (!block_at(z, zn, off, &bsize, &in_off, &in_len, &isize, &crc)) return false;
        if (isize) pieces->push_back({in_off, in_len, isize, 0u, (uint32_t)isize, out, crc});
        out += isize;
        off += bsize;
    }
    *text_size = out;
    return !pieces->empty() || zn == 0 || out == 0;
}

struct Region { uint64_t beg, end; };    // virtual offsets

// The [begin, end) virtual offsets of the wanted reference sequences of a tabix index held in memory (already inflated):
// from the metadata pseudo-bin 37450 when present, else the hull of the sequence's chunks; regions sorted, neighbours
// merged.  *found = wanted names the index knows.  Returns an empty string, or what is wrong with the index.
// block_starts (may be NULL): the file offsets at which the wanted sequences' regions begin, ascending, unique — every one the first
// byte of a BGZF block: where region_pieces may take up the walk on another thread.
// per_want (may be NULL): the region of every wanted name by its number in `want` ({0, 0} and present = 0 for names the index lacks;
// a sequence without records is present with beg == end).
inline std::string tabix_regions(const uint8_t *t, size_t tn, const std::unordered_map<std::string, uint32_t> &want,
                                 std::vector<Region> *merged, uint64_t *found, std::vector<uint64_t> *block_starts = nullptr,
                                 std::vector<Region> *per_want = nullptr, std::vector<uint8_t> *present = nullptr) {
    const std::string bad = "not a tabix index";
    if (tn < 36 || memcmp(t, "TBI\1", 4) != 0) return bad;
    auto i32 = [&](size_t o) { int32_t v; memcpy(&v, t + o, 4); return v; };
    auto u64 = [&](size_t o) { uint64_t v; memcpy(&v, t + o, 8); return v; };
    const int32_t n_ref = i32(4), l_nm = i32(32);
    if (n_ref < 0 || l_nm < 0 || 36 + (size_t)l_nm > tn) return bad;
    std::vector<std::string> ref_names;
    for (size_t o = 36; o < 36 + (size_t)l_nm;) {
        const char *z = reinterpret_cast<const char *>(t + o);
        const size_t len = strnlen(z, 36 + (size_t)l_nm - o);
        ref_names.emplace_back(z, len);
        o += len + 1;
    }
    if ((int32_t)ref_names.size() != n_ref) return bad;
    std::vector<Region> regions;
    *found = 0;
    if (per_want) per_want->assign(want.size(), Region{0, 0});
    if (present) present->assign(want.size(), 0);
    size_t o = 36 + (size_t)l_nm;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (o + 4 > tn) return bad;
        const int32_t n_bin = i32(o);
        o += 4;
        uint64_t lo = ~0ull, hi = 0;
        bool pseudo = false;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (o + 8 > tn) return bad;
            uint32_t bin;
            memcpy(&bin, t + o, 4);
            const int32_t n_chunk = i32(o + 4);
            o += 8;
            if (n_chunk < 0 || o + (size_t)n_chunk * 16 > tn) return bad;
            if (bin == 37450 && n_chunk >= 1) {              // metadata pseudo-bin: chunk 0 = [begin, end) of the reference
                lo = u64(o);
                hi = u64(o + 8);
                pseudo = true;
            } else if (!pseudo) {
                for (int32_t k = 0; k < n_chunk; ++k) {
                    lo = std::min(lo, u64(o + (size_t)k * 16));
                    hi = std::max(hi, u64(o + (size_t)k * 16 + 8));
                }
            }
            o += (size_t)n_chunk * 16;
        }
        if (o + 4 > tn) return bad;
        const int32_t n_intv = i32(o);
        o += 4;
        if (n_intv < 0 || o + (size_t)n_intv * 8 > tn) return bad;
        o += (size_t)n_intv * 8;
        const auto w = want.find(ref_names[r]);
        if (w != want.end()) {
            *found += 1;
            if (hi > lo) regions.push_back({lo, hi});
            if (per_want && w->second < per_want->size()) (*per_want)[w->second] = hi > lo ? Region{lo, hi} : Region{0, 0};
            if (present && w->second < present->size()) (*present)[w->second] = 1;
        }
    }
    std::sort(regions.begin(), regions.end(), [](const Region &x, const Region &y) { return x.beg < y.beg; });
    if (block_starts) {
        block_starts->clear();
        for (const Region &g : regions)
            if (block_starts->empty() || block_starts->back() != (g.beg >> 16)) block_starts->push_back(g.beg >> 16);
    }
    merged->clear();
    for (const Region &g : regions) {
        if (!merged->empty() && g.beg <= merged->back().end) merged->back().end = std::max(merged->back().end, g.end);
        else merged->push_back(g);
    }
    return std::string();
}

// the pieces of the regions' text.  Returns an empty string, or the error.
// The walk from block to block reads every block's header and trailer: one or two pages of the mapped file per block, 1.2 million
// blocks in the pileup of a 1 Gbp metagenome — a second of page faults on one thread.  With the block offsets the index itself
// names (block_starts of tabix_regions) the walk is cut into stretches that `threads` workers take in turn; a stretch must end
// exactly where the next one begins (it does when the index belongs to the file).
inline std::string region_pieces(const uint8_t *z, size_t zn, const std::vector<Region> &merged, std::vector<Piece> *pieces,
                                 uint64_t *text_size, uint64_t *inflated, const std::vector<uint64_t> *block_starts = nullptr,
                                 unsigned threads = 1, int fd = -1, bool *index_problem = nullptr) {
    // *index_problem: the error is the INDEX's (it points off the blocks of this file: a stale .tbi, the caller may read the whole
    // file), not the file's (a damaged block) — callers branch on this flag, never on the message
    if (index_problem) *index_problem = false;
    pieces->clear();
    // stretches: [first block, stop) of one region; stop = the next restart point, or beyond the region's last block
    struct Stretch { size_t region; size_t off, stop; bool to_region_end; std::vector<Piece> pieces; std::string error; bool index_error = false; };
    std::vector<Stretch> work;
    for (size_t r = 0; r < merged.size(); ++r) {
        const Region &g = merged[r];
        const size_t first = (size_t)(g.beg >> 16), last = (size_t)(g.end >> 16);
        std::vector<size_t> cuts(1, first);
        if (block_starts && threads > 1) {
            auto lo = std::upper_bound(block_starts->begin(), block_starts->end(), (uint64_t)first);
            auto hi = std::upper_bound(block_starts->begin(), block_starts->end(), (uint64_t)last);
            const size_t n_points = (size_t)(hi - lo), want = (size_t)threads * 8;
            const size_t step = std::max<size_t>(1, n_points / std::max<size_t>(want, 1));
            for (size_t k = step; k < n_points; k += step) cuts.push_back((size_t)lo[k]);
        }
        for (size_t k = 0; k < cuts.size(); ++k)
            work.push_back(Stretch{r, cuts[k], k + 1 < cuts.size() ? cuts[k + 1] : last + 1, k + 1 == cuts.size(), {}, {}, false});
    }
    auto walk = [&](Stretch &w) {
        const Region &g = merged[w.region];
        const size_t first = (size_t)(g.beg >> 16), last = (size_t)(g.end >> 16);
        const uint32_t u_beg = (uint32_t)(g.beg & 0xFFFF), u_end = (uint32_t)(g.end & 0xFFFF);
        size_t off = w.off;
        while (off < w.stop && off + 18 <= zn) {
            size_t bsize, in_off, in_len, isize;
            uint32_t crc;
            if (fd >= 0) {
                bool is_bgzf = false;
                if (!block_at_fd(fd, zn, off, &bsize, &in_off, &in_len, &isize, &crc, &is_bgzf)) {
                    w.error = is_bgzf ? "corrupt BGZF block" : "the index points outside a BGZF block";
                    w.index_error = !is_bgzf;
                    return;
                }
            } else {
                if (!(z[off] == 31 && z[off + 1] == 139 && z[off + 2] == 8 && (z[off + 3] & 4))) { w.error = "the index points outside a BGZF block"; w.index_error = true; return; }
                if (!block_at(z, zn, off, &bsize, &in_off, &in_len, &isize, &crc)) { w.error = "corrupt BGZF block"; return; }
            }
            const uint32_t skip = off == first ? u_beg : 0u;
            const uint32_t stop = off == last ? u_end : (uint32_t)isize;
            if (skip > isize || stop > isize) { w.error = "the index points beyond a BGZF block"; w.index_error = true; return; }
            if (stop > skip) w.pieces.push_back({in_off, in_len, isize, skip, stop - skip, 0, crc});
            off += bsize;
        }
        // a stretch that is followed by another one of its region must end on that one's first block
        if (!w.to_region_end && off != w.stop) { w.error = "the index points outside a BGZF block"; w.index_error = true; }
    };
    if (threads > 1 && work.size() > 1) {
        std::vector<std::thread> pool;
        const unsigned nt = (unsigned)std::min<size_t>(threads, work.size());
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t] { for (size_t i = t; i < work.size(); i += nt) walk(work[i]); });
        for (auto &th : pool) th.join();
    } else {
        for (auto &w : work) walk(w);
    }
    uint64_t text = 0, infl = 0;
    size_t total = 0;
    for (const auto &w : work) {
        if (!w.error.empty()) {
            if (index_problem) *index_problem = w.index_error;
            return w.error;
        }
        total += w.pieces.size();
    }
    pieces->reserve(total);
    for (auto &w : work)
        for (Piece &p : w.pieces) {
            p.text_off = text;
            text += p.take;
            infl += p.out_len;
            pieces->push_back(p);
        }
    *text_size = text;
    *inflated = infl;
    return std::string();
}

// the wanted text of one piece -> dst (take bytes); tmp is scratch for pieces that are not taken whole
inline bool inflate_piece(const uint8_t *z, const Piece &p, char *dst, std::vector<char> &tmp) {
    if (p.skip == 0 && p.take == p.out_len) return inflate_raw(z + p.in_off, p.in_len, dst, p.out_len, p.crc);
    tmp.resize(p.out_len);
    if (!inflate_raw(z + p.in_off, p.in_len, tmp.data(), p.out_len, p.crc)) return false;
    memcpy(dst, tmp.data() + p.skip, p.take);
    return true;
}

}  // namespace nmbgzf
