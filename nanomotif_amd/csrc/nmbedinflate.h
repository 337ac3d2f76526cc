// DEFLATE on the device: the kernels that inflate the BGZF blocks of a bgzip pileup and check their CRC-32 (dataload.py:102-152: the
// reference reads such files through gzip / htslib).  Internal, not part of the C ABI.  Included by nmbedgpu.hip INSIDE its anonymous
// namespace (and by tools/inflate2_proto.hip, the stand-alone harness that times these kernels on synthetic slabs): no namespace and
// no includes of its own here.
#pragma once

// ---- DEFLATE on the device (round 4): the BGZF blocks of a bgzip pileup (<= 64 KiB of text each, independent streams) are
// inflated by the GPU — one LANE per block, canonical Huffman decoding from per-length code counts (held in registers) and a
// symbol table per lane in LDS, output straight into the text buffer the parse kernels read.  A lane takes ~40 ms for a
// block (every back-reference is a round trip through memory), but a 3 GiB slab is 48 000 blocks in flight together:
// tools/inflate_proto.hip measures 27 GB/s of text on one wave per CU against 0.9 GB/s for zlib on a host thread, and only
// the compressed bytes (a fifth of the text) cross PCIe.
struct InfPiece {
    unsigned long long in_off;          // raw deflate stream inside the packed compressed buffer
    unsigned int in_len, out_len;       // compressed bytes, text bytes of the whole block
    unsigned long long dst_off;         // where the WANTED text goes in the slab
    unsigned int skip, take;            // the wanted part of the block's text
    unsigned long long full_off;        // partial blocks (skip / take cut them): the whole text goes to scratch + full_off first
    unsigned int crc;                   // CRC-32 the member's trailer states for the block's whole text
    unsigned int tok_len;               // two-phase inflate (round 6): bytes of the block's token region (a multiple of 16) ...
    unsigned long long tok_off;         // ... and where it starts in the token buffer
};

constexpr int INF_LANES = 64, INF_MAXL = 288, INF_MAXD = 30;
constexpr int INF_TURN_SYMBOLS = 1;       // symbols a lane may decode per turn of its wave (bed_inflate_kernel)
// A damaged deflate block is noticed at its end at the latest (the text cannot grow beyond the member's size: at most 65 536
// symbols of at most 48 bits) and at the head of the next one: the bit reader never gets further than this past the stream.
constexpr size_t INF_OVERRUN = 512u << 10;
// Entry-major: lane l of the wave touches [entry][l].  A symbol of the literal / length code is 9 bits: its low byte in `lo`, bit 8
// in a bit plane of 288 bits per lane — 25.1 KB per workgroup instead of the 44.8 KB of 16-bit entries, six workgroups on a
// compute unit's 160 KB instead of three (round 5: the lanes wait on memory most of the time; a CU held three waves for four SIMDs).
struct InfSymbols {
    unsigned char lo[INF_MAXL][INF_LANES];
    unsigned int hi[(INF_MAXL + 31) / 32][INF_LANES];
    __device__ __forceinline__ void clear(int lane) {
#pragma unroll
        for (int w = 0; w < (INF_MAXL + 31) / 32; ++w) hi[w][lane] = 0;
    }
    __device__ __forceinline__ void put(int idx, int lane, int sym) {
        lo[idx][lane] = (unsigned char)sym;
        if (sym & 256) hi[idx >> 5][lane] |= 1u << (idx & 31);
    }
    __device__ __forceinline__ int get(int idx, int lane) const { return (int)lo[idx][lane] | (int)(((hi[idx >> 5][lane] >> (idx & 31)) & 1u) << 8); }
};
struct InfDistSymbols {                 // distance symbols are below 30
    unsigned char lo[INF_MAXD][INF_LANES];
    __device__ __forceinline__ void clear(int) {}
    __device__ __forceinline__ void put(int idx, int lane, int sym) { lo[idx][lane] = (unsigned char)sym; }
    __device__ __forceinline__ int get(int idx, int lane) const { return (int)lo[idx][lane]; }
};
struct InfTables {
    unsigned short lcount[16][INF_LANES];
    unsigned char dcount[16][INF_LANES];
    InfSymbols lsym;
    InfDistSymbols dsym;
};

struct InfBits {
    const unsigned char *p;
    unsigned long long buf;
    int cnt;
    // The 8 bytes at p, LOADED AHEAD: a refill takes them from this register and at once asks memory for the 8 bytes at the new p,
    // which nobody looks at before the next refill, several symbols later.  (Round 5: the wave waited for a refill load of SOME lane
    // in nearly every iteration — 64 lanes, a refill every two or three symbols each — one round trip through L2 per symbol.)
    unsigned long long ahead;
    __device__ __forceinline__ void start() { memcpy(&ahead, p, 8); }   // (the compressed buffer has 8 readable bytes after its end)
    __device__ __forceinline__ void refill() {
        buf |= ahead << cnt;
        const int take = (63 - cnt) >> 3;
        p += take;
        cnt += take * 8;
        memcpy(&ahead, p, 8);
    }
    __device__ __forceinline__ unsigned int get(int n) {       // n <= 16
        if (cnt < 32) refill();
        const unsigned int v = (unsigned int)(buf & ((1ull << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
};

// canonical code from code lengths; > 0: incomplete, < 0: over-subscribed (zlib contrib/puff: construct)
template <typename Count, typename Symbols>
__device__ int inf_construct(Count (*count)[INF_LANES], Symbols &symbol, const unsigned char *length, int n, int lane) {
    symbol.clear(lane);
    for (int len = 0; len <= 15; ++len) count[len][lane] = 0;
    for (int s = 0; s < n; ++s) count[length[s]][lane] += 1;
    if (count[0][lane] == n) return 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= count[len][lane];
        if (left < 0) return left;
    }
    unsigned short offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len][lane];
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) symbol.put(offs[length[s]]++, lane, s);
    return left;
}

struct InfCounts { unsigned int c[16]; };
template <typename Count>
__device__ __forceinline__ InfCounts inf_counts(Count (*count)[INF_LANES], int lane) {
    InfCounts k;
#pragma unroll
    for (int len = 0; len < 16; ++len) k.c[len] = count[len][lane];
    return k;
}

template <typename Symbols>
__device__ __forceinline__ int inf_decode(InfBits &b, const InfCounts &k, const Symbols &symbol, int lane) {
    if (b.cnt < 32) b.refill();
    int code = 0, first = 0, index = 0;
    unsigned int bits = (unsigned int)b.buf;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int c = (int)k.c[len];
        if (code - c < first) {
            b.buf >>= len;
            b.cnt -= len;
            return symbol.get(index + (code - first), lane);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__constant__ unsigned short INF_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ unsigned char INF_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ unsigned short INF_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ unsigned char INF_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ unsigned char INF_CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// what phase 1 of the two-phase inflate (below) leaves per block; INF2_FULL / INF2_WIDE in `err` are not errors of the stream: the block's
// tokens did not fit its region / it uses more literal-length codes than phase 1's table holds — bed_inflate_kernel inflates such blocks
struct InfTokMeta {
    unsigned int n_seq, n_lit, err, pad;
};
constexpr unsigned int INF2_FULL = 21, INF2_WIDE = 22;

#ifdef NM_BED_PROBES
// probe builds (NM_CXXFLAGS=-DNM_BED_PROBES, tools/gpu_r5ac.sh): NM_BED_INFLATE_PROBE = 1: matches are not copied, + 2: literals are not
// stored, + 4: a block ends behind its first pair of Huffman tables — what each part of bed_inflate_kernel costs (the text is garbage: the
// call fails on purpose after printing the slab's time)
__device__ int g_inf_probe = 0;
#endif

// status: 0, or (piece index << 8 | what went wrong) of the first bad block
__global__ __launch_bounds__(INF_LANES) void bed_inflate_kernel(const unsigned char *__restrict__ in, const InfPiece *__restrict__ pieces, unsigned int n_pieces,
                                                                unsigned char *__restrict__ text, unsigned char *__restrict__ scratch,
                                                                unsigned int *__restrict__ status, const InfTokMeta *__restrict__ only_full) {
    __shared__ InfTables T;
#ifdef NM_BED_PROBES
    const int probe = g_inf_probe;
#else
    constexpr int probe = 0;
#endif
    const int lane = threadIdx.x;
    const unsigned int i = blockIdx.x * INF_LANES + lane;
    if (i >= n_pieces) return;
    if (only_full) {                                                    // (two-phase inflate: the blocks phase 1 gave up, nothing else)
        const unsigned int gave_up = only_full[i].err;
        if (gave_up != INF2_FULL && gave_up != INF2_WIDE) return;
    }
    const InfPiece pc = pieces[i];
    const bool partial = pc.skip != 0 || pc.take != pc.out_len;
    InfBits b{in + pc.in_off, 0ull, 0, 0ull};
    b.start();
    unsigned char *dst = partial ? scratch + pc.full_off : text + pc.dst_off;
    unsigned int o = 0;
    int err = 0, last = 0;
    unsigned char lengths[INF_MAXL + INF_MAXD];
    while (!last && !err) {
        if ((size_t)(b.p - (in + pc.in_off)) > (size_t)pc.in_len + 8) { err = 18; break; }      // ran past the block's stream
        last = (int)b.get(1);
        const int type = (int)b.get(2);
        if (type == 0) {                                         // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const unsigned int len = b.get(16), nlen = b.get(16);
            if ((len ^ 0xFFFFu) != nlen || o + len > pc.out_len) { err = 2; break; }
            for (unsigned int k = 0; k < len; ++k) dst[o++] = (unsigned char)b.get(8);
            continue;
        }
        if (type == 3) { err = 3; break; }
        if (type == 1) {                                         // fixed code
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            inf_construct(T.lcount, T.lsym, lengths, 288, lane);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            inf_construct(T.dcount, T.dsym, lengths, 30, lane);
        } else {                                                 // dynamic code
            const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
            if (nlen > 286 || ndist > 30) { err = 4; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[INF_CLORDER[idx]] = (unsigned char)b.get(3);
            for (; idx < 19; ++idx) lengths[INF_CLORDER[idx]] = 0;
            if (inf_construct(T.lcount, T.lsym, lengths, 19, lane) != 0) { err = 5; break; }
            const InfCounts kc = inf_counts(T.lcount, lane);
            idx = 0;
            while (idx < nlen + ndist) {
                const int sym = inf_decode(b, kc, T.lsym, lane);
                if (sym < 0) { err = 6; break; }
                if (sym < 16) lengths[idx++] = (unsigned char)sym;
                else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { err = 7; break; }
                        len = lengths[idx - 1];
                        rep = 3 + (int)b.get(2);
                    } else if (sym == 17) rep = 3 + (int)b.get(3);
                    else rep = 11 + (int)b.get(7);
                    if (idx + rep > nlen + ndist) { err = 8; break; }
                    while (rep--) lengths[idx++] = (unsigned char)len;
                }
            }
            if (err) break;
            if (lengths[256] == 0) { err = 9; break; }
            int r = inf_construct(T.lcount, T.lsym, lengths, nlen, lane);
            if (r < 0 || (r > 0 && nlen - T.lcount[0][lane] != 1)) { err = 10; break; }
            r = inf_construct(T.dcount, T.dsym, lengths + nlen, ndist, lane);
            if (r < 0 || (r > 0 && ndist - T.dcount[0][lane] != 1)) { err = 11; break; }
        }
        const InfCounts kl = inf_counts(T.lcount, lane), kd = inf_counts(T.dcount, lane);
        if (probe & 4) break;
        // The 64 lanes of a wave decode 64 blocks in lock-step: whatever ONE lane does in an iteration, the others wait for.  A match
        // used to be copied whole inside the iteration that decoded it — every iteration then cost the LONGEST match among the lanes,
        // one round trip through memory per 8 (or 32) bytes of it (round 5: that was most of a lane's 64 ms per block).  Now a match is
        // a STATE of the lane: an iteration copies one bounded piece of it — at most 32 bytes, all loaded before any is stored: one
        // round trip — and lanes without a match in progress decode their next symbol meanwhile.
        unsigned int pend = 0, pdist = 0;                        // bytes of the match in progress still to copy, its distance
        // A piece's source is LOADED at the end of one turn and STORED at the beginning of the next: the round trip to L2 / HBM that
        // every turn used to wait out (some lane of the 64 always has a copy whose source missed L2; the counters of
        // profiles/r5/bed_device/inflate_pmc.txt: 71 % of a wave's life waiting, one exposed round trip per turn) now runs under the
        // other lanes' decoding.  Same-lane order keeps it exact: a piece is stored before the next piece's (or the next match's)
        // source is asked for, and a lane decodes nothing while it has a match in progress.
        struct Q { unsigned long long a, b; };
        Q q0 = {0, 0}, q1 = {0, 0};
        unsigned int fl = 0, fl_kind = 0;                        // bytes loaded and not yet stored (0: none); 1: 16-byte words, 2: one 8-byte word, 3: periodic
        for (;;) {                                               // the block's symbols, one TURN of the wave per pass
            if (fl) {                                            // (1) the piece loaded in the last turn goes out
                unsigned char *d = dst + o;
                if (fl_kind == 1) {
                    memcpy(d, &q0, 16);
                    if (fl > 16) memcpy(d + 16, &q1, 16);
                } else if (fl_kind == 2) {
                    memcpy(d, &q0.a, 8);
                } else {
                    // a distance below 8 repeats its last `pdist` bytes: 16 bytes of the periodic sequence in two registers, the
                    // words of the piece are cut out of them at the phase they start with
                    const unsigned long long s8 = q0.a;
                    unsigned long long lo = 0, hi = 0;
                    unsigned int ph = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const unsigned long long byte = (s8 >> (8 * ph)) & 0xFFull;
                        if (j < 8) lo |= byte << (8 * j);
                        else hi |= byte << (8 * (j - 8));
                        ph = ph + 1 == pdist ? 0 : ph + 1;
                    }
                    unsigned int r = 0;                           // phase of the next word = (bytes written so far) mod pdist
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((unsigned int)(8 * k) < fl) {
                            const unsigned long long w = r ? (lo >> (8 * r)) | (hi << (64 - 8 * r)) : lo;
                            memcpy(d + 8 * k, &w, 8);
                        }
                        r = (r + 8) % pdist;
                    }
                }
                o += fl;
                pend -= fl;
                fl = 0;
            }
            // (2) a lane without a match in progress decodes its next symbol (INF_TURN_SYMBOLS > 1: several — measured slower, 0.66 ->
            // 0.85 s at 1 Gbp: the wave then pays the longest run of literals among its lanes in every turn)
            bool stop = false;                                   // end of block, or an error
            for (int rep = 0; rep < INF_TURN_SYMBOLS && !pend && !stop; ++rep) {
                int sym = inf_decode(b, kl, T.lsym, lane);
                if (sym < 0) { err = 12; stop = true; }
                else if (sym < 256) {
                    if (o >= pc.out_len) { err = 13; stop = true; }
                    else if (probe & 2) o += 1;
                    else dst[o++] = (unsigned char)sym;
                } else if (sym == 256) stop = true;
                else {
                    sym -= 257;
                    if (sym >= 29) { err = 14; stop = true; }
                    else {
                        // base and extra bits of a length / distance code by ARITHMETIC (RFC 1951, 3.2.5: the codes come in groups of
                        // four / two per extra bit): the __constant__ tables were four DEPENDENT per-lane loads per match — length
                        // base, its extra bits, distance base, its extra bits, each a trip to L1 — in front of the copy's own
                        const unsigned int lx = sym < 8 || sym == 28 ? 0u : ((unsigned int)sym >> 2) - 1u;
                        const unsigned int lbase = sym < 8 ? 3u + (unsigned int)sym : sym == 28 ? 258u : 3u + ((4u + ((unsigned int)sym & 3u)) << lx);
                        const unsigned int len = lbase + b.get((int)lx);
                        const int ds = inf_decode(b, kd, T.dsym, lane);
                        if (ds < 0 || ds >= 30) { err = 15; stop = true; }
                        else {
                            const unsigned int dx = ds < 4 ? 0u : ((unsigned int)ds >> 1) - 1u;
                            const unsigned int dbase = ds < 4 ? 1u + (unsigned int)ds : 1u + ((2u + ((unsigned int)ds & 1u)) << dx);
                            const unsigned int dist = dbase + b.get((int)dx);
                            if (dist > o || o + len > pc.out_len) { err = 16; stop = true; }
                            else {
                                pend = len;                      // copied piece by piece, from this turn on
                                pdist = dist;
                            }
                        }
                    }
                }
            }
            if (stop) break;                                     // (no match is in progress then: a lane decodes only without one)
            if (probe & 1) { o += pend; pend = 0; }
            if (pend) {                                          // (3) the next piece's source is asked for
                unsigned char *d = dst + o;
                const unsigned char *src = d - pdist;
                if (o + 40 > pc.out_len) {                       // at the block's end there is no room for whole words: byte by byte, at once
                    for (unsigned int k = 0; k < pend; ++k) d[k] = src[k];
                    o += pend;
                    pend = 0;
                } else if (pdist >= 16) {
                    // one or two 16-byte words, as many as lie wholly in front of the piece's first store
                    const unsigned int nq = pdist >= 32 ? 2u : 1u;
                    fl = pend < 16 * nq ? pend : 16 * nq;
                    fl_kind = 1;
                    memcpy(&q0, src, 16);
                    if (fl > 16) memcpy(&q1, src + 16, 16);
                } else {
                    fl = pdist >= 8 ? (pend < 8 ? pend : 8) : (pend < 32 ? pend : 32);
                    fl_kind = pdist >= 8 ? 2 : 3;
                    memcpy(&q0.a, src, 8);
                }
            }
        }
    }
    if (probe) return;
    if (!err && o != pc.out_len) err = 17;
    if (!err && (size_t)(b.p - (in + pc.in_off)) > (size_t)pc.in_len + 8) err = 18;      // ran past the block's stream
    if (err) { atomicCAS(status, 0u, (unsigned int)err | (i << 8)); return; }
    if (partial) {
        unsigned char *out = text + pc.dst_off;
        for (unsigned int k = 0; k < pc.take; ++k) out[k] = dst[pc.skip + k];
    }
}

// ---- Two-phase inflate (round 6).  One LANE per block was the shape of everything above: 64 unrelated byte streams per wave, every
// store a partial line, every match source a private round trip (profiles/r5/bed_device/inflate_pmc.txt: a wave waits on memory 71 % of
// its life, DRAM sees 3.5 x the text in writes).  Here the two jobs part:
//   phase 1  bed_tokens_kernel   still one lane per block — Huffman decoding IS serial per stream — but it only DECODES: the compressed
//                                bytes come through a ring per lane in LDS that is topped up 64 bytes at a time, far ahead of the bit
//                                reader (no turn of the wave waits for some lane's refill any more), and what leaves the lane is a
//                                compact token stream written front to back: 4-byte sequence records (literal run, match length,
//                                distance) growing up from the start of the block's token region, the literal bytes growing down from
//                                its end.  No match is copied, nothing is read back.
//   phase 2  bed_resolve_kernel  one WAVE per block: 64 sequences at a time (<= INF2_CAP bytes of text), staged in LDS.  Every byte of
//                                the chunk gets the DISTANCE to a byte of equal value (0 for a literal); distances that land on another
//                                match byte of the chunk are added up by pointer doubling (a chain of k matches takes log2 k rounds, a
//                                run-length match log2(length / distance)), until every match byte points at a literal of the chunk (in
//                                LDS) or in front of the chunk (text this wave stored earlier: one batch of gathers per chunk).  Then
//                                the chunk leaves as whole 16-byte stores, 1 KiB per wave instruction.
// A block's text is written exactly once and in whole lines of the memory system; the only scattered reads left are phase 2's gathers.
constexpr unsigned int INF2_CAP = 1536;            // text bytes of a chunk of phase 2 (a sequence is at most 254 + 258 bytes)
constexpr int INF2_ROWS = (int)(INF2_CAP / 64);     // 64-byte rows of a chunk


// Bytes of a block's token region.  4 bytes per match (>= 3 bytes of text each), 4 per 255 literals that meet no match, 1 per literal,
// the last literal word written whole: never more than 4/3 of the text + 32.  bedMethyl text needs 0.4 of its size (5 700 records +
// 3 000 literals per 65 280 bytes), and the buffers are fresh device memory on the critical path of a run: the host sizes a region at
// `fraction` of the text (5/8 by default), phase 1 gives a block up when its tokens do not fit (INF2_FULL) and the one-lane-per-block
// kernel above inflates exactly those blocks (only_full) — any deflate stream stays readable, the common one pays for what it needs.
__host__ __device__ inline unsigned int inf2_region_bytes(unsigned int out_len, double fraction) {
    const unsigned long long worst = ((unsigned long long)out_len * 4u / 3u + 32u + 15u) & ~15ull;
    const unsigned long long want = ((unsigned long long)((double)out_len * fraction) + 64u + 15u) & ~15ull;
    return (unsigned int)(want < worst ? want : worst);
}


// sequence record: literals in front of the match (0..254; 255 = 255 literals and no match), bit 8 = no match (the tail of a block),
// match length - 3, distance - 1
__device__ __forceinline__ unsigned int inf2_record(unsigned int lit, bool no_match, unsigned int len, unsigned int dist) {
    return lit | (no_match ? 256u : 0u) | ((len - 3u) << 9) | ((dist - 1u) << 17);
}

constexpr unsigned int INF2_RING = 16;                             // dwords of compressed bytes per lane in the ring (a power of two, >= 16)
struct InfRing { unsigned int w[INF2_RING][INF_LANES]; };          // [word][lane]

// bit reader of phase 1: the stream arrives in pieces of 32 bytes — two 16-byte loads issued at EVERY service call (every eight turns of
// the wave), written into the ring (INF2_RING = 16 dwords per lane) at the next one, when they have long arrived, provided the ring had room for them when they were asked
// for (otherwise the same bytes are asked for again: no state hangs on a condition, so the loads stay in flight across the turns instead
// of being waited for on the spot) — and is consumed a dword at a time; the dword the next refill will take is read from the ring one
// refill ahead, so that no turn waits for it either
struct InfBits2 {
    const uint4 *src;                   // the lane's stream from the 16-byte boundary in front of its first byte
    unsigned int fill, rd;              // dwords written into the ring / taken out of it so far (ring[rd] is in `ahead`)
    unsigned long long buf;
    int cnt;
    unsigned int ahead;
    uint4 q0, q1;
    bool room;                          // the ring had room for q0, q1 when they were asked for
    __device__ __forceinline__ void land(InfRing &r, int lane) {
        if (room) {
            const unsigned int f = fill;
            constexpr unsigned int M = INF2_RING - 1;
            r.w[(f + 0) & M][lane] = q0.x; r.w[(f + 1) & M][lane] = q0.y; r.w[(f + 2) & M][lane] = q0.z; r.w[(f + 3) & M][lane] = q0.w;
            r.w[(f + 4) & M][lane] = q1.x; r.w[(f + 5) & M][lane] = q1.y; r.w[(f + 6) & M][lane] = q1.z; r.w[(f + 7) & M][lane] = q1.w;
            fill = f + 8;
        }
    }
    __device__ __forceinline__ void ask() {
        room = fill - rd <= INF2_RING - 8u;
        const uint4 *p = src + (fill >> 2);
        q0 = p[0]; q1 = p[1];
    }
    // every eight turns: on bedMethyl text a turn takes 13 bits, eight turns 3 - 4 dwords of the 8 that arrive; a stream that takes more
    // than 32 bits per turn for long (long matches at far distances, one after the other) drains the ring and pays refill()'s slow path
    __device__ __forceinline__ void service(InfRing &r, int lane) {
        land(r, lane);
        ask();
    }
    __device__ __forceinline__ void start(InfRing &r, int lane, unsigned int skip) {
        fill = 0; rd = 0; buf = 0; cnt = 0;
        ask();
        for (unsigned int k = 0; k < INF2_RING / 8; ++k) service(r, lane);      // the ring is full (a ring of 32: 24 dwords and 8 on their way)
        rd = skip >> 2;
        ahead = r.w[rd & (INF2_RING - 1)][lane];
        if (skip & 3u) { refill(r, lane); buf >>= 8 * (skip & 3u); cnt -= 8 * (int)(skip & 3u); }
    }
    __device__ __forceinline__ void refill(InfRing &r, int lane) {
        buf |= (unsigned long long)ahead << cnt;
        rd += 1;
        cnt += 32;
        if (rd == fill) {                                              // the ring is empty (header loops; streams of > 32 bits per turn)
            service(r, lane);
            if (rd == fill) service(r, lane);
        }
        ahead = r.w[rd & (INF2_RING - 1)][lane];
    }
    __device__ __forceinline__ unsigned int get(int n, InfRing &r, int lane) {       // n <= 16
        if (cnt < 32) refill(r, lane);
        const unsigned int v = (unsigned int)(buf & ((1ull << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
    __device__ __forceinline__ long long consumed_bytes(unsigned int skip) const { return (long long)rd * 4 - (long long)skip - (cnt >> 3); }
};

// A canonical code WITHOUT a loop over the lengths that ends early (64 lanes in lock-step paid the longest code of the turn, a dozen
// branches each): lim[L - 1] = the first 15 bits of the stream, read most significant bit first, are below it for every code of at most L
// bits.  The length of the code in front of the reader is 1 + the number of limits the 15 bits reach — fifteen compares, no branch —
// and tab[len - 1] = (lower limit of that length's codes | index of its first symbol << 16) gives the symbol's place.
struct InfCanon { unsigned int lim[15]; };

// 1 + the number of limits x reaches.  The limits grow with the length, so the signs of lim - 1 - x (x, lim < 2^16) are ones up to the
// code's length and zeros from there on: they are shifted into one word (v_sub + v_alignbit per limit, no compare / select pairs with
// their wait states) and counted
__device__ __forceinline__ unsigned int inf2_code_length(unsigned int x, const InfCanon &k) {
    unsigned int signs = 0;
#pragma unroll
    for (int l = 0; l < 15; ++l) signs = __builtin_amdgcn_alignbit(signs, k.lim[l] - 1u - x, 31);
    return 1u + (unsigned int)__popc(signs & 0x7FFFu);
}

// Phase 1 keeps its LDS at 39 040 bytes per workgroup — a ring of 16 dwords, the codes per length counted in the rows of the table being
// built — so that FOUR workgroups fit a CU's 160 KB, one per SIMD (with 45 KB — ring of 32, a count table of its own — three did: the
// stand-alone harness decodes 65 536 blocks in the 9.7 ms it takes for 49 152).  A slab is 771 workgroups; the fourth slot of a CU takes
// workgroups of the next slab, which the staging thread has queued on another stream; phase-2 waves (4.6 KB each) get their LDS as phase-1
// workgroups retire.  The
// literal / length table holds INF2_LSYM codes: all 288 (a table of 224 would make room for two more phase-2 waves, but every block
// in the FIXED code — the short last block of most files — would then go through bed_inflate_kernel, ~40 ms for however few blocks;
// the INF2_WIDE path below stays for whoever shrinks the table).
constexpr int INF2_LSYM = 288;
struct InfSymbols2 {
    unsigned char lo[INF2_LSYM][INF_LANES];
    unsigned int hi[INF2_LSYM / 32][INF_LANES];
    __device__ __forceinline__ void clear(int lane) {
#pragma unroll
        for (int w = 0; w < INF2_LSYM / 32; ++w) hi[w][lane] = 0;
    }
    __device__ __forceinline__ void put(int idx, int lane, int sym) {
        lo[idx][lane] = (unsigned char)sym;
        if (sym & 256) hi[idx >> 5][lane] |= 1u << (idx & 31);
    }
    __device__ __forceinline__ int get(int idx, int lane) const { return (int)lo[idx][lane] | (int)(((hi[idx >> 5][lane] >> (idx & 31)) & 1u) << 8); }
};

struct InfTables2 {
    unsigned int ltab[16][INF_LANES], dtab[16][INF_LANES];      // (while a table is built its rows hold the number of codes per length)
    InfSymbols2 lsym;
    InfDistSymbols dsym;
};

// canonical code from code lengths; > 0: incomplete, < 0: over-subscribed (zlib contrib/puff: construct)
// (*too_wide: more codes in use than the symbol table holds — nothing is written then)
// (the codes per length are counted in the rows of `tab` itself: row len is read before row len - 1 is written — a count table of its own
//  was 2 KB of the workgroup's LDS; *n_codes: how many symbols have a code)
template <typename Symbols>
__device__ int inf2_construct(unsigned int (*tab)[INF_LANES], Symbols &symbol, InfCanon &canon, const unsigned char *length, int n, int lane,
                              int capacity, bool *too_wide, int *n_codes) {
    symbol.clear(lane);
    unsigned int (*count)[INF_LANES] = tab;
    for (int len = 0; len <= 15; ++len) count[len][lane] = 0;
    for (int s = 0; s < n; ++s) count[length[s]][lane] += 1;
    *n_codes = n - (int)count[0][lane];
    if (*n_codes > capacity) { *too_wide = true; return 0; }
    int left = 1;
    if (*n_codes == 0) left = 0;                                        // (no code at all: every decode fails, as in puff)
    else
        for (int len = 1; len <= 15; ++len) {
            left <<= 1;
            left -= (int)count[len][lane];
            if (left < 0) return left;
        }
    unsigned short offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + (unsigned short)count[len][lane];
    unsigned int acc = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        const unsigned int c = count[len][lane];                       // (row len: still a count; row len - 1 becomes the table's entry)
        tab[len - 1][lane] = acc | ((unsigned int)offs[len] << 16);
        acc += c << (15 - len);
        canon.lim[len - 1] = acc;
    }
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) symbol.put(offs[length[s]]++, lane, s);
    return left;
}

template <typename Symbols>
__device__ __forceinline__ int inf2_decode(InfBits2 &b, InfRing &ring, const InfCanon &k, const unsigned int (*tab)[INF_LANES], const Symbols &symbol, int lane) {
    if (b.cnt < 32) b.refill(ring, lane);
    const unsigned int x = __builtin_bitreverse32((unsigned int)b.buf) >> 17;
    const unsigned int len = inf2_code_length(x, k);
    if (len > 15u) return -1;
    const unsigned int t = tab[len - 1][lane];
    const unsigned int idx = (t >> 16) + ((x - (t & 0xFFFFu)) >> (15u - len));
    b.buf >>= len;
    b.cnt -= (int)len;
    return symbol.get((int)idx, lane);
}

struct InfShared2 {
    InfTables2 T;
    InfRing ring;
    unsigned int rec[16][INF_LANES];            // a ring of the lane's latest sequence records: they leave eight at a time, as one whole 32-byte sector
};

// phase 1.  status: as bed_inflate_kernel (0, or piece index << 8 | code of the first bad block); meta[i].err repeats the code per block
__global__ __launch_bounds__(INF_LANES) void bed_tokens_kernel(const unsigned char *__restrict__ in, const InfPiece *__restrict__ pieces, unsigned int n_pieces,
                                                               unsigned char *__restrict__ tokens, InfTokMeta *__restrict__ meta, unsigned int *__restrict__ status) {
    __shared__ InfShared2 S;
    InfTables2 &T = S.T;
    InfRing &ring = S.ring;
    const int lane = threadIdx.x;
    const unsigned int i = blockIdx.x * INF_LANES + lane;
    if (i >= n_pieces) return;
    const InfPiece pc = pieces[i];
    const unsigned char *first_byte = in + pc.in_off;
    const unsigned int skip = (unsigned int)((size_t)first_byte & 15u);
    InfBits2 b;
    b.src = reinterpret_cast<const uint4 *>(first_byte - skip);
    b.start(ring, lane, skip);
    unsigned char *region = tokens + pc.tok_off;
    unsigned int *seq = reinterpret_cast<unsigned int *>(region);
    unsigned char *lit_end = region + pc.tok_len;
    // Every global store of the lane happens at a service call, next to the ring's loads: the records of the turns since the last call
    // wait in LDS (n_seq = stored + staged), a completed word of eight literals waits in a register (at most one completes in eight turns).
    // The wait for the ring's loads at the next call — eight turns later — is then a wait for operations that have long finished, stores
    // included (loads and stores share one counter: a store of the turn before would be waited for with them)
    unsigned int o = 0, n_stored = 0, n_staged = 0, n_lit = 0, run = 0;
    unsigned long long lw = 0, lw_done = 0;
    unsigned int lw_at = 0;                                           // 0: no completed word; else the literal count it ends at
    int err = 0, last = 0;
    unsigned int turn = 0;
    unsigned char lengths[INF_MAXL + INF_MAXD];
    // (n_stored is a multiple of 8 while the block lasts: record k sits in slot k mod 16, a flush takes the eight oldest — an aligned
    //  32-byte piece of the stream, written once; round 6 first stored whatever was staged every eight turns, eight dwords from the
    //  first staged record on: 1.6 x the bytes, every sector written twice)
    auto store8 = [&]() {
        const unsigned int s0 = n_stored & 8u;
        uint4 a, c;
        a.x = S.rec[s0 + 0][lane]; a.y = S.rec[s0 + 1][lane]; a.z = S.rec[s0 + 2][lane]; a.w = S.rec[s0 + 3][lane];
        c.x = S.rec[s0 + 4][lane]; c.y = S.rec[s0 + 5][lane]; c.z = S.rec[s0 + 6][lane]; c.w = S.rec[s0 + 7][lane];
        *reinterpret_cast<uint4 *>(seq + n_stored) = a;
        *reinterpret_cast<uint4 *>(seq + n_stored + 4) = c;
        n_stored += 8;
        n_staged -= 8;
    };
    auto record = [&](unsigned int rec) {
        if (n_staged == 16u) store8();                                // (never between two services of the symbol loop: stored-block and header paths)
        S.rec[(n_stored + n_staged) & 15u][lane] = rec;
        n_staged += 1;
    };
    auto flush = [&]() {
        if (n_staged >= 8u) store8();
        if (lw_at) { memcpy(lit_end - lw_at, &lw_done, 8); lw_at = 0; }
    };
    auto literal = [&](unsigned int byte) {
        lw |= (unsigned long long)byte << (8u * (7u - (n_lit & 7u)));
        n_lit += 1;
        if ((n_lit & 7u) == 0) {
            if (lw_at) memcpy(lit_end - lw_at, &lw_done, 8);            // (only where no service came in between: stored blocks)
            lw_done = lw; lw_at = n_lit; lw = 0;
        }
        o += 1;
        run += 1;
        if (run == 255u) { record(inf2_record(255u, true, 3u, 1u)); run = 0; }
    };
    while (!last && !err) {
        if (b.consumed_bytes(skip) > (long long)pc.in_len + 8) { err = 18; break; }      // ran past the block's stream
        last = (int)b.get(1, ring, lane);
        const int type = (int)b.get(2, ring, lane);
        if (type == 0) {                                         // stored
            b.buf >>= (b.cnt & 7);
            b.cnt -= (b.cnt & 7);
            const unsigned int len = b.get(16, ring, lane), nlen = b.get(16, ring, lane);
            if ((len ^ 0xFFFFu) != nlen || o + len > pc.out_len) { err = 2; break; }
            for (unsigned int k = 0; k < len && !err; ++k) {
                if ((k & 7u) == 0) { flush(); b.service(ring, lane); }
                if (4u * (n_stored + n_staged) + (n_lit | 7u) + 56u > pc.tok_len) err = (int)INF2_FULL;
                else literal(b.get(8, ring, lane));
            }
            continue;
        }
        if (type == 3) { err = 3; break; }
        InfCanon kl, kd;
        bool wide = false;
        if (type == 1) {                                         // fixed code
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            int used = 0;
            inf2_construct(T.ltab, T.lsym, kl, lengths, 288, lane, INF2_LSYM, &wide, &used);
            if (wide) { err = (int)INF2_WIDE; break; }                  // (never with a table of 288: the fixed code uses them all)
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            inf2_construct(T.dtab, T.dsym, kd, lengths, 30, lane, INF_MAXD, &wide, &used);
        } else {                                                 // dynamic code
            const int nlen = (int)b.get(5, ring, lane) + 257, ndist = (int)b.get(5, ring, lane) + 1, ncode = (int)b.get(4, ring, lane) + 4;
            if (nlen > 286 || ndist > 30) { err = 4; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[INF_CLORDER[idx]] = (unsigned char)b.get(3, ring, lane);
            for (; idx < 19; ++idx) lengths[INF_CLORDER[idx]] = 0;
            InfCanon kc;
            int used = 0;
            if (inf2_construct(T.ltab, T.lsym, kc, lengths, 19, lane, INF2_LSYM, &wide, &used) != 0) { err = 5; break; }
            idx = 0;
            unsigned int hs = 0;
            while (idx < nlen + ndist) {
                if ((hs++ & 7u) == 0) b.service(ring, lane);
                const int sym = inf2_decode(b, ring, kc, T.ltab, T.lsym, lane);
                if (sym < 0) { err = 6; break; }
                if (sym < 16) lengths[idx++] = (unsigned char)sym;
                else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { err = 7; break; }
                        len = lengths[idx - 1];
                        rep = 3 + (int)b.get(2, ring, lane);
                    } else if (sym == 17) rep = 3 + (int)b.get(3, ring, lane);
                    else rep = 11 + (int)b.get(7, ring, lane);
                    if (idx + rep > nlen + ndist) { err = 8; break; }
                    while (rep--) lengths[idx++] = (unsigned char)len;
                }
            }
            if (err) break;
            if (lengths[256] == 0) { err = 9; break; }
            int r = inf2_construct(T.ltab, T.lsym, kl, lengths, nlen, lane, INF2_LSYM, &wide, &used);
            if (wide) { err = (int)INF2_WIDE; break; }
            if (r < 0 || (r > 0 && used != 1)) { err = 10; break; }
            r = inf2_construct(T.dtab, T.dsym, kd, lengths + nlen, ndist, lane, INF_MAXD, &wide, &used);
            if (r < 0 || (r > 0 && used != 1)) { err = 11; break; }
        }
        // The block's symbols, one per turn of the wave.  Straight-line code: a literal / length symbol, then — for every lane, whether its
        // symbol was a length or not — the extra bits, the distance symbol and its extra bits, consuming no bits where there is no match;
        // all that can go wrong is collected in `bad` and looked at once at the end of the turn.  (64 lanes in lock-step execute both
        // sides of every branch anyway; without the branches the scalar unit has nothing to do and the turn is ~3 x shorter.)
        for (;;) {
            if ((turn++ & 7u) == 0) { flush(); b.service(ring, lane); }
            unsigned int bad = 0;
            // literal / length symbol (the reader holds >= 32 bits after the check: 15 + 5 fit)
            if (b.cnt < 32) b.refill(ring, lane);
            unsigned int sym;
            {
                const unsigned int x = __builtin_bitreverse32((unsigned int)b.buf) >> 17;
                unsigned int len = inf2_code_length(x, kl);
                bad |= len > 15u ? 12u : 0u;
                len = len > 15u ? 15u : len;
                const unsigned int t = T.ltab[len - 1][lane];
                unsigned int idx = (t >> 16) + ((x - (t & 0xFFFFu)) >> (15u - len));
                idx = idx < (unsigned int)INF2_LSYM ? idx : 0u;
                sym = (unsigned int)T.lsym.get((int)idx, lane);
                b.buf >>= len;
                b.cnt -= (int)len;
            }
            const bool is_lit = sym < 256u, is_end = sym == 256u, is_match = sym > 256u;
            unsigned int ms = is_match ? sym - 257u : 0u;
            bad |= ms >= 29u ? 14u : 0u;
            ms = ms >= 29u ? 0u : ms;
            const unsigned int lx = !is_match || ms < 8u || ms == 28u ? 0u : (ms >> 2) - 1u;
            const unsigned int lbase = ms < 8u ? 3u + ms : ms == 28u ? 258u : 3u + ((4u + (ms & 3u)) << lx);
            const unsigned int mlen = lbase + (unsigned int)(b.buf & ((1ull << lx) - 1));
            b.buf >>= lx;
            b.cnt -= (int)lx;
            // distance symbol (15 + 13 bits)
            if (b.cnt < 32) b.refill(ring, lane);
            unsigned int ds;
            {
                const unsigned int x = __builtin_bitreverse32((unsigned int)b.buf) >> 17;
                unsigned int len = inf2_code_length(x, kd);
                bad |= is_match && len > 15u ? 15u : 0u;
                len = len > 15u ? 15u : len;
                const unsigned int t = T.dtab[len - 1][lane];
                unsigned int idx = (t >> 16) + ((x - (t & 0xFFFFu)) >> (15u - len));
                idx = idx < (unsigned int)INF_MAXD ? idx : 0u;
                ds = (unsigned int)T.dsym.get((int)idx, lane);
                len = is_match ? len : 0u;
                b.buf >>= len;
                b.cnt -= (int)len;
            }
            bad |= is_match && ds >= 30u ? 15u : 0u;
            ds = ds >= 30u ? 0u : ds;
            const unsigned int dx = !is_match || ds < 4u ? 0u : (ds >> 1) - 1u;
            const unsigned int dbase = ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx);
            const unsigned int dist = dbase + (unsigned int)(b.buf & ((1ull << dx) - 1));
            b.buf >>= dx;
            b.cnt -= (int)dx;
            bad |= is_match && (dist > o || o + mlen > pc.out_len) ? 16u : 0u;
            bad |= is_lit && o >= pc.out_len ? 13u : 0u;
            // room for this turn's record / literal (a literal may complete a word of 8 and a run of 255 at once; a flush writes eight
            // dwords from the first staged record on: 32 bytes that must stay below the literals)
            if (!bad && 4u * (n_stored + n_staged) + (n_lit | 7u) + 56u > pc.tok_len) bad = INF2_FULL;
            if (is_lit && !bad) literal(sym);
            if (is_match && !bad) {
                record(inf2_record(run, false, mlen, dist));
                run = 0;
                o += mlen;
            }
            if (bad) err = (int)(bad & 31u);                           // (several checks may have failed at once: any of their codes)
            if (bad || is_end) break;
        }
    }
    if (!err && o != pc.out_len) err = 17;
    if (!err && b.consumed_bytes(skip) > (long long)pc.in_len + 8) err = 18;
    if (!err) {
        if (run) record(inf2_record(run, true, 3u, 1u));
        flush();
        if (n_staged >= 8u) store8();
        for (unsigned int k = 0; k < n_staged; ++k) seq[n_stored + k] = S.rec[(n_stored + k) & 15u][lane];      // the last, incomplete piece
        n_stored += n_staged;
        n_staged = 0;
        if (n_lit & 7u) memcpy(lit_end - ((n_lit + 7u) & ~7u), &lw, 8);
    }
    meta[i] = InfTokMeta{n_stored, n_lit, (unsigned int)err, 0u};
    if (err && err != (int)INF2_FULL && err != (int)INF2_WIDE) atomicCAS(status, 0u, (unsigned int)err | (i << 8));
}

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ unsigned int inf2_wave_scan(unsigned int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int u = (unsigned int)__shfl_up((int)v, d);
        if (lane >= d) v += u;
    }
    return v;
}

struct InfChunk2 {
    union {
        unsigned short dist[INF2_CAP];             // per byte of the chunk: 0 = literal, else the distance to a byte of equal value
        unsigned int dist_pair[INF2_CAP / 2];      // (two entries at a time)
    };
    unsigned char text[16 + INF2_CAP + 16];        // the chunk's bytes, shifted so that index = address mod 16 of where they go
};

// phase 2: one wave per block
__global__ __launch_bounds__(64) void bed_resolve_kernel(const InfPiece *__restrict__ pieces, unsigned int n_pieces, const unsigned char *__restrict__ tokens,
                                                         const InfTokMeta *__restrict__ meta, unsigned char *__restrict__ text, unsigned char *__restrict__ scratch,
                                                         unsigned int *__restrict__ status) {
    __shared__ InfChunk2 C;
    const int lane = threadIdx.x;
    const unsigned int i = blockIdx.x;
    if (i >= n_pieces) return;
    const InfTokMeta m = meta[i];
    if (m.err) return;                                               // (phase 1 has reported it)
    const InfPiece pc = pieces[i];
    const bool partial = pc.skip != 0 || pc.take != pc.out_len;
    unsigned char *dst = partial ? scratch + pc.full_off : text + pc.dst_off;
    const unsigned char *region = tokens + pc.tok_off;
    const unsigned int *seq = reinterpret_cast<const unsigned int *>(region);
    const unsigned char *lit_end = region + pc.tok_len;
    unsigned int first = 0, out = 0, lit_pos = 0;
    bool bad = false;
    // A chunk is a chain of dependent steps, and what a wave waits for between them is memory: the records, the literals, the gathers,
    // the stores before the next gathers.  The records of the NEXT chunk are asked for as soon as this chunk's size is known, the
    // literals before the distances are written, and the wait for the chunk's stores stands in front of the next chunk's gathers, not
    // behind the stores: one exposed round trip per chunk (the gathers) instead of four
    unsigned int rec_ahead = (unsigned int)lane < m.n_seq ? seq[lane] : 0u;
    while (first < m.n_seq) {
        // (a) up to 64 sequences, as many of them as fit the chunk
        const unsigned int j = first + (unsigned int)lane;
        const unsigned int rec = rec_ahead;
        unsigned int lit = rec & 255u;
        unsigned int mlen = (rec & 256u) ? 0u : ((rec >> 9) & 255u) + 3u;
        const unsigned int dist = (rec >> 17) + 1u;
        if (j >= m.n_seq) { lit = 0; mlen = 0; }
        const unsigned int packed = inf2_wave_scan(((lit + mlen) << 16) | lit, lane);      // bytes << 16 | literals, both below 2^15 over 64 lanes
        const unsigned int end_rel = packed >> 16;
        const unsigned long long fits = __ballot(j < m.n_seq && end_rel <= INF2_CAP);
        const int n_take = __popcll(fits);                               // (a prefix: end_rel grows with the lane; at least one: a sequence is <= 512 bytes)
        // (the same in every lane — said so, or the loops below are run with vector masks instead of scalar counters)
        const unsigned int n_bytes = (unsigned int)__builtin_amdgcn_readfirstlane(__shfl((int)end_rel, n_take - 1));
        const unsigned int n_lits = (unsigned int)__builtin_amdgcn_readfirstlane(__shfl((int)(packed & 0xFFFFu), n_take - 1));
        if (lane >= n_take) { lit = 0; mlen = 0; }
        if (out + n_bytes > pc.out_len || lit_pos + n_lits > m.n_lit) { bad = true; break; }      // (wave-uniform; cannot happen with phase 1's tokens)
        const unsigned int o_rel = end_rel - lit - mlen, lp = lit_pos + (packed & 0xFFFFu) - lit;
        const unsigned int a0 = (unsigned int)((size_t)(dst + out) & 15u);
        {
            const unsigned int jn = first + (unsigned int)n_take + (unsigned int)lane;
            rec_ahead = jn < m.n_seq ? seq[jn] : 0u;
        }
        // (b) literal runs: the literals of a run sit in descending addresses below lit_end, eight to an (unaligned) 8-byte load; the first
        //     eight of every run are asked for here and written behind (c)
        unsigned long long w0 = 0;
        if (lit) memcpy(&w0, lit_end - lp - 8, 8);
        for (unsigned int k8 = 8; __any(k8 < lit); k8 += 8) {
            if (k8 < lit) {
                unsigned long long w;
                memcpy(&w, lit_end - (lp + k8) - 8, 8);
#pragma unroll
                for (unsigned int k = 0; k < 8; ++k)
                    if (k8 + k < lit) {
                        C.text[a0 + o_rel + k8 + k] = (unsigned char)(w >> (8u * (7u - k)));
                        C.dist[o_rel + k8 + k] = 0;
                    }
            }
        }
        // (c) every match byte: the distance of its match; the unused bytes of the last 64-byte row: literals (the loops below run over
        //     whole rows and carry no bounds checks)
        {
            // (two entries per store where the pair is aligned: half the turns of the loop, which the longest match of the chunk sets)
            const unsigned int m0 = o_rel + lit, m1 = m0 + mlen;
            unsigned int at = m0;
            if ((at & 1u) && at < m1) { C.dist[at] = (unsigned short)dist; at += 1; }
            const unsigned int pair = dist | (dist << 16);
            while (__any(at + 2u <= m1))
                if (at + 2u <= m1) { C.dist_pair[at >> 1] = pair; at += 2; }
            if (at < m1) C.dist[at] = (unsigned short)dist;
        }
        const unsigned int rows = (n_bytes + 63u) >> 6;
        if (n_bytes + (unsigned int)lane < rows * 64u) C.dist[n_bytes + (unsigned int)lane] = 0;
#pragma unroll
        for (unsigned int k = 0; k < 8; ++k)
            if (k < lit) {
                C.text[a0 + o_rel + k] = (unsigned char)(w0 >> (8u * (7u - k)));
                C.dist[o_rel + k] = 0;
            }
        // (d) distances that land on another match byte of this chunk are added up (pointer doubling); a byte is settled once its
        //     distance lands on a literal of the chunk or in front of the chunk.  Reading a neighbour's distance while its owner
        //     updates it is harmless: the old and the new value both lead to a byte of equal value.  No branch inside a round: a byte
        //     that has nothing to add reads itself and writes back what it held
        //     A row (64 bytes) in which nothing was added in a round is settled for good — literals and bytes in front of the chunk never
        //     change — and is left out of the later rounds
        for (unsigned int live = rows >= 32u ? ~0u : (1u << rows) - 1u; live != 0;) {
            unsigned int next = 0;
            for (unsigned int todo = live; todo != 0; todo &= todo - 1u) {
                const unsigned int r = (unsigned int)__builtin_ctz(todo);
                const unsigned int bb = r * 64u + (unsigned int)lane;
                const unsigned int d = C.dist[bb];
                const bool inside = d - 1u < bb;                       // 1 <= d <= bb
                const unsigned int d2 = C.dist[bb - (inside ? d : 0u)];
                const unsigned int add = inside ? d2 : 0u;
                C.dist[bb] = (unsigned short)(d + add);
                next |= __ballot(add != 0) != 0ull ? 1u << r : 0u;
            }
            live = next;
        }
        // (e) the match bytes: from a literal of this chunk, or from text stored by an earlier chunk (eight rows' gathers asked for together).
        //     The gathers read what the chunk before stored: same wave, same L1 — the stores must have left first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (unsigned int r0 = 0; r0 < rows; r0 += 8) {
            unsigned int dd[8];
            unsigned char far[8];
#pragma unroll
            for (unsigned int q = 0; q < 8; ++q) {
                dd[q] = 0;
                far[q] = 0;
                if (r0 + q < rows) {                                 // (wave-uniform)
                    const unsigned int bb = (r0 + q) * 64u + (unsigned int)lane;
                    const unsigned int d = C.dist[bb];
                    dd[q] = d;
                    if (d > bb) far[q] = dst[out + bb - d];
                }
            }
#pragma unroll
            for (unsigned int q = 0; q < 8; ++q)
                if (r0 + q < rows) {
                    const unsigned int bb = (r0 + q) * 64u + (unsigned int)lane;
                    const unsigned int d = dd[q];
                    const unsigned char near = C.text[a0 + bb - (d <= bb ? d : 0u)];      // (d = 0: the byte itself)
                    C.text[a0 + bb] = d > bb ? far[q] : near;
                }
        }
        // (f) the chunk leaves in whole 16-byte groups (the bytes in front of the first and behind the last group one by one)
        {
            unsigned char *g0 = dst + out - a0;                       // 16-byte aligned
            const unsigned int end = a0 + n_bytes;
            for (unsigned int g = (unsigned int)lane; g * 16u < end; g += 64u) {
                const unsigned int lo = g * 16u, hi = lo + 16u;
                if (lo >= a0 && hi <= end) {
                    *reinterpret_cast<uint4 *>(g0 + lo) = *reinterpret_cast<const uint4 *>(&C.text[lo]);
                } else {
                    for (unsigned int k = lo < a0 ? a0 : lo; k < (hi < end ? hi : end); ++k) g0[k] = C.text[k];
                }
            }
        }
        out += n_bytes;
        lit_pos += n_lits;
        first += (unsigned int)n_take;
    }
    if (bad || out != pc.out_len || lit_pos != m.n_lit) {
        if (lane == 0) atomicCAS(status, 0u, 20u | (i << 8));
        return;
    }
    if (partial) {
        unsigned char *o2 = text + pc.dst_off;
        for (unsigned int k = (unsigned int)lane; k < pc.take; k += 64u) o2[k] = dst[pc.skip + k];
    }
}

// CRC-32 of every inflated block against its member's trailer (what Python's gzip and htslib check for the reference): one
// WAVE per block, the text streamed through in passes of 4 KiB — lane l takes 64 consecutive bytes of a pass, so a wave's loads
// cover whole cache lines once.  A CRC is linear over GF(2): lane l computes the register of the message with everything but
// ITS bytes zeroed (between two of its chunks lie 4032 zero bytes: one multiplication by the constant x^(8 * 4032) mod P, eight
// nibble look-ups), the lanes' registers are carried to the end of the message (x^(8 * 64 * (63 - l)), once per block) and
// XORed together.  Chunks are aligned to the END of the block, so only the first one can be short; the initial value
// 0xFFFFFFFF is the same as complementing the first four bytes of the message.
struct CrcConsts {
    unsigned int gap;                   // x^(8 * 4032) mod P
    unsigned int to_end[64];            // x^(8 * 64 * (63 - lane)) mod P
};

// product of two polynomials modulo the CRC-32 polynomial, bit-reflected (bit 31 = x^0) like the CRC register itself
__host__ __device__ inline unsigned int crc_mulmod(unsigned int a, unsigned int b) {
    unsigned int p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;        // b * x
    }
    return p;
}

inline unsigned int crc_xpow8(unsigned int n_bytes) {            // x^(8 n) mod P
    unsigned int r = 0x80000000u, q = 0x00800000u;              // 1, x^8
    for (; n_bytes; n_bytes >>= 1) {
        if (n_bytes & 1u) r = crc_mulmod(r, q);
        q = crc_mulmod(q, q);
    }
    return r;
}

inline CrcConsts crc_consts() {
    CrcConsts k;
    k.gap = crc_xpow8(4032);
    for (int l = 0; l < 64; ++l) k.to_end[l] = crc_xpow8(64u * (unsigned int)(63 - l));
    return k;
}

__global__ __launch_bounds__(256) void bed_crc_kernel(const InfPiece *__restrict__ pieces, unsigned int n_pieces, const unsigned char *__restrict__ text,
                                                      const unsigned char *__restrict__ scratch, CrcConsts k, unsigned int *__restrict__ status) {
    __shared__ unsigned int table[256], gap[8][16];
    {
        unsigned int c = threadIdx.x;
        for (int j = 0; j < 8; ++j) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        table[threadIdx.x] = c;
        if (threadIdx.x < 128) gap[threadIdx.x >> 4][threadIdx.x & 15] = crc_mulmod((threadIdx.x & 15u) << (4 * (threadIdx.x >> 4)), k.gap);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_pieces) return;                                  // (wave-uniform)
    const InfPiece pc = pieces[i];
    const bool partial = pc.skip != 0 || pc.take != pc.out_len;
    const unsigned char *src = partial ? scratch + pc.full_off : text + pc.dst_off;
    const long long n = pc.out_len;
    unsigned int c;
    if (n <= 64) {                                              // the whole message is lane 63's only chunk: the plain definition
        c = 0;
        if (lane == 63) {
            c = 0xFFFFFFFFu;
            for (long long at = 0; at < n; ++at) c = table[(c ^ src[at]) & 255u] ^ (c >> 8);
            c ^= 0xFFFFFFFFu;
        }
    } else {
        c = 0;
        const int first_pass = 15 - (int)((n - 1) >> 12);
        for (int pass = first_pass; pass < 16; ++pass) {        // (wave-uniform bounds)
            // the register moves over the 4032 bytes of the other lanes (zeros to this lane)
            c = gap[0][c & 15u] ^ gap[1][(c >> 4) & 15u] ^ gap[2][(c >> 8) & 15u] ^ gap[3][(c >> 12) & 15u] ^ gap[4][(c >> 16) & 15u] ^
                gap[5][(c >> 20) & 15u] ^ gap[6][(c >> 24) & 15u] ^ gap[7][c >> 28];
            const long long e = n - (long long)(15 - pass) * 4096 - (long long)(63 - lane) * 64, b = e - 64;
            if (e <= 0) continue;
            if (b < 4) {                                        // the chunk that holds the beginning of the message: byte by byte
                for (long long at = b > 0 ? b : 0; at < e; ++at) c = table[(c ^ src[at] ^ (at < 4 ? 255u : 0u)) & 255u] ^ (c >> 8);
            } else {
                unsigned long long w[8];
                memcpy(w, src + b, 64);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        c = table[(c ^ (unsigned int)w[q]) & 255u] ^ (c >> 8);
                        w[q] >>= 8;
                    }
                }
            }
        }
        c = crc_mulmod(c, k.to_end[lane]);
        for (int d = 32; d; d >>= 1) c ^= (unsigned int)__shfl_xor((int)c, d);
        c ^= 0xFFFFFFFFu;
    }
    if (lane == 63 && c != pc.crc) atomicCAS(status, 0u, 19u | (i << 8));
}

