// Internal to the translation units of libnmscan (nmscan.hip, nmingest.hip, nmwindows.hip): engine constants, the
// plane / context structures and the small host helpers they share.  Not part of the C ABI (include/nmscan.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/nmscan.h"
#include "nmspec.h"

// error sink shared by all translation units (defined in nmscan.hip; the text is what nm_last_error() returns)
int nm_set_error(int code, const char *fmt, ...);
#define fail nm_set_error

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(NM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

namespace nmdetail {

// Device memory comes from hipMalloc, or from the host application's pool when it registered one
// (nm_set_device_allocator: a process that already holds a caching pool — torch — hands out blocks it has paid for;
// fresh hipMallocs of memory other processes used before are scrubbed by the driver at 7-30 GB/s, tools/alloc_probe.py).
hipError_t device_alloc(void **p, size_t bytes);
hipError_t device_free(void *p);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return device_alloc(reinterpret_cast<void **>(p), bytes); }
inline hipError_t dev_free(void *p) { return device_free(p); }

constexpr int T_WORDS = 4;                              // 32-bit words per lane
constexpr int CHUNK_WORDS = 64 * T_WORDS;               // 256 words
constexpr int CHUNK_BP = CHUNK_WORDS * 32;              // 8192 positions per wave-chunk
constexpr int GAP_BP = 96;                              // invalid positions guaranteed after every contig (>= the widest offset a motif reaches: no match straddles contigs)
constexpr int SEG_CHUNKS = 16;                          // chunks per workgroup segment (128 Kbp)
constexpr int BMAX = 32;                                // candidates per LDS accumulation pass
// device-side programs are packed to the word-groups the launched kernel variant reads:
// narrow (offsets in [-32, 31]) = groups 1..2 -> 32 dwords (128 B), wide = all four -> 64 dwords
constexpr int NM_MAX_MOD_CODES = 8;                     // mod codes that can be given a slot / reported (ABI: slot_of_mod[8])
constexpr int NM_CODE_STRIDE = 128;                     // mod code ids the pre-filters tell apart (int8 ids; the reader numbers unknown codes 3, 4, ...)
constexpr int WIN_MAX_W = NM_WIN_MAX_WIDTH;   // window width limit (reference default 41)
constexpr int RANK_BLOCK_WORDS = 16;                          // 512 bp per rank entry
constexpr int RANK_PER_CHUNK = CHUNK_WORDS / RANK_BLOCK_WORDS;

struct Planes {
    const uint32_t *H, *L, *V;
    const uint8_t *needs_v;   // per chunk: 1 => V (and halo) must be consulted
};

struct StatePlanes {
    const uint32_t *M, *U;               // compact (strand implied by base)
    const uint32_t *MP, *UP, *MM, *UM;   // general
};

// One candidate record as the host stages it (sorted by mod-type slot, then bin).
struct CandRec {
    uint32_t mask_off;   // into the staged mask bytes
    uint32_t orig;       // caller's index of this candidate
    uint8_t len, modpos, slot, pad;
};

struct WinTask {
    uint64_t plane_off;     // into the plane pool (words): [col][5][nw]
    uint64_t alive_off;     // into the alive pool (words): [nw]
    uint32_t n, nw, width, pad;
};

struct ModSlot {
    bool present = false;
    uint8_t canonical = 0;          // 'A' or 'C'
    double low = 0.3, high = 0.7;
    uint32_t *planes[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // M U MP UP MM UM
    uint64_t n_rows = 0;
    // window extraction: rank tables over the methylated-row planes MP / MM and the per-contig row counts for the
    // edge padding `meth_pad` ({n_plus, n_minus, head_plus, head_minus} per contig); dropped when the planes change
    uint32_t *rank[2] = {nullptr, nullptr};
    uint64_t *rank_total[2] = {nullptr, nullptr};
    std::vector<uint64_t> meth_counts;
    uint32_t meth_pad = 0xFFFFFFFFu;
};

// Read statistics of one mod code for the per-contig read-methylation table (nmmeth.hip): presence planes of the records
// that passed the read filters, rank tables over them, and the records' counts in plane order.
struct ReadStats {
    bool present = false;
    uint32_t *planes = nullptr;                   // P+ | P-  (2 x plane words)
    uint32_t *rank[2] = {nullptr, nullptr};       // per 512-bp block: records of the contig before the block
    uint64_t *base[2] = {nullptr, nullptr};       // per contig: index of its first record in val[]
    uint2 *val[2] = {nullptr, nullptr};           // (n_valid_cov, n_modified) per record, position order
    uint64_t n_rows[2] = {0, 0};
};

}  // namespace nmdetail

// Staging pairs (device + pinned host buffer) and parts of the program table.  Scoring walks all of them: the host side,
// the upload and the compile of batch k+1 ... k+3 run while batch k is scored — on a small shard of a multi-GPU run that
// chain (~85 us) is as long as the scoring kernel itself, two pairs would hide only one kernel's worth of it.
#define NM_STAGE_RING 8
#define NM_FLIGHTS NM_SEARCH_MAX_FLIGHTS        /* batches of one kind that may be begun and not yet collected: the native search keeps several
                                                  groups of tasks in flight, some travelling while another is on the host (nmsearch.cpp);
                                                  each holds up to two pairs of the ring (its children's batch, its regular scoring batch) */
#define NM_STAGE_SLOTS (NM_STAGE_RING + NM_FLIGHTS)   /* + the pairs of the asynchronous window batches (nm_win_batch_w_begin) */

struct nm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;
    // second scoring lane (nm_set_score_lanes): asynchronous device-output batches alternate between the ctx's own
    // stream and this one, so that consecutive INDEPENDENT batches overlap on the device (the next launch fills the
    // compute units the previous one leaves idle while it drains, and nothing waits for a launch gap)
    hipStream_t lane_stream = nullptr;
    int score_lanes = 1;
    bool lane_pending = false;                        // work may be in flight on lane_stream
    hipStream_t last_score_stream = nullptr;          // where the most recent scoring launch went
    std::vector<uint32_t> bucket;                     // per-call host scratch, kept to avoid reallocation
    // window engine
    std::vector<nmdetail::WinTask> win_tasks;
    uint32_t *d_win_planes = nullptr, *d_win_alive = nullptr;
    uint64_t win_planes_cap = 0, win_alive_cap = 0, win_planes_used = 0, win_alive_used = 0;
    nmdetail::WinTask *d_win_tasks = nullptr;
    size_t d_win_tasks_cap = 0;
    bool win_tasks_dirty = false;
    // results of the last nm_ingest_pileup
    std::vector<uint32_t> ing_kept;                   // kept rows per (contig, mod code)
    // the confident rows of the last ingest are the set bits of the MP / MM planes of these slots (no list is built)
    int ing_slot_of_mod[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    uint64_t ing_nconf = 0, ing_total_kept = 0, ing_classified = 0;   // accumulated over the parts of one pileup
    uint32_t *d_programs = nullptr;                   // compiled constraint programs of the current batch
    size_t prog_cap_dw = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // per-launch event pairs since the last nm_timing_reset (bounded pool, summed lazily: no sync per launch)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    bool ev_collect = false;
    bool busy_open = false;
    bool ev_collect_all = false;                      // nm_timing_reset(ctx, 2): also the ingest / window / background phases
    // assembly
    uint32_t n_contigs = 0, n_bins = 0, n_chunks = 0;
    uint64_t total_bp = 0;
    std::vector<uint64_t> contig_len;
    std::vector<uint32_t> contig_chunk, contig_bin, contig_nchunks;
    std::vector<uint32_t> bin_chunk0, bin_nchunks, bin_ncontigs, contig_rank;   // contig_rank: position of a contig inside its bin
    uint32_t *d_chunk_rank = nullptr;                 // per chunk: contig_rank of its contig (per-contig counters)
    uint32_t *dH = nullptr, *dL = nullptr, *dV = nullptr;
    uint8_t *d_needs_v = nullptr;
    uint32_t *d_contig_chunk = nullptr;
    uint64_t *d_contig_len = nullptr;
    uint4 *d_segments = nullptr;
    uint32_t n_segments = 0;
    nmdetail::ModSlot slots[NM_MAX_MOD_SLOTS];
    nmdetail::ReadStats readstats[NM_MAX_MOD_SLOTS];   // nm_readstats_upload (per-contig read methylation, nmmeth.hip)
    uint32_t *d_chunk_contig = nullptr;                // per chunk: its contig, ~0 for pad chunks (built on first use)
    // per-call staging: ring of two (device, pinned host) buffer pairs so that compiling the next batch on the
    // host overlaps the previous launch; `busy` marks the last device work that read the pair.
    struct Stage {
        void *d = nullptr, *h = nullptr;
        size_t bytes = 0;
        hipEvent_t busy = nullptr;
        bool pending = false;
        const void *last_out = nullptr;        // count table the last scoring launch of this pair wrote, and its stream
        hipStream_t last_stream = nullptr;
    } stage[NM_STAGE_SLOTS];
    int stage_next = 0;                            // shallow users alternate between pairs 0 and 1
    int stage_next_deep = 0;                       // scoring walks the whole ring
    void *d_stage = nullptr, *h_stage = nullptr;   // the pair acquired by the current call
    Stage *cur_stage = nullptr;
    // a batch begun and not yet collected (nm_score_batch_begin / nm_win_batch_w_begin): where its results land in pinned memory
    struct Waiting {
        const void *h = nullptr;
        size_t bytes = 0;
        Stage *stage = nullptr;
        bool open = false;
    } score_wait[NM_FLIGHTS], win_wait[NM_FLIGHTS], spec_wait[NM_FLIGHTS];   // spec_wait: the speculative child scores riding on a window batch
    // The native search sends the batches of one flight from a thread of its own while the calling thread collects another flight's
    // (nmsearch.cpp): the *_begin halves run on the sending thread, the *_end halves on the collecting one.  What the two share are
    // these slots — ensure_stage looks at all of them to find a pair nobody holds — hence the mutex around every change of `open`;
    // a pair stays held until its results have been copied out of the pinned half.
    std::mutex wait_mu;
    hipStream_t flight_stream[NM_FLIGHTS - 1] = {};   // window batches of flight f > 0 (flight 0: copy_stream), made on first use
    // speculative child scoring of the search: the tasks' background PSSMs [task][4][W] (doubles, rows A T G C) and the count table
    double *d_spec_bg = nullptr;
    uint32_t spec_tasks = 0, spec_width = 0;
    unsigned long long *d_spec_counts[NM_FLIGHTS] = {}, *d_flight_counts[NM_FLIGHTS] = {};   // count tables of the deferred batches of a flight
    size_t spec_counts_cap[NM_FLIGHTS] = {}, flight_counts_cap[NM_FLIGHTS] = {};
    unsigned long long *d_counts = nullptr;
    size_t counts_cap = 0;
    unsigned int *d_err = nullptr;
    // window extraction: per-base rank tables over the sequence planes (built on first use), other-letter count
    uint32_t *d_rank[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_base_total[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long *d_other = nullptr;
    uint64_t other_letters = 0;
    uint64_t launches = 0, last_wgs = 0, last_compact = 0, last_general = 0;
    // multi-GPU exchange (nmcomm.cpp): RCCL communicator, its stream, one completion event per table buffer
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready = nullptr, comm_done[NM_COMM_SLOTS] = {};
    bool comm_pending[NM_COMM_SLOTS] = {};
    // kernel-variant switches for A/B runs (environment: NM_NO_LIT, NM_NO_CF, NM_PREFETCH), read at nm_ctx_create
    bool opt_no_lit = false, opt_no_cf = false;
    bool opt_stream_wait = false;
    bool opt_no_inline = false;         // NM_NO_INLINE_PREP: host-count batches prepare on the copy stream with two launches, as before round 3
    int opt_fine = -1;                  // NM_FINE: log2 of the extra cut of the last-dispatched pieces, -1 = 2 - split
    int opt_split = -1;                 // NM_SPLIT: force the segment split (log2), -1 = by size
    uint32_t n_cus = 256;
    uint32_t seg_chunks = nmdetail::SEG_CHUNKS;      // chunks per workgroup segment (NM_SEG_CHUNKS)
};

namespace nmdetail {

inline size_t plane_words(const nm_ctx *c) { return (size_t)c->n_chunks * CHUNK_WORDS; }

inline Planes seq_planes(const nm_ctx *c) {
    Planes p;
    p.H = c->dH;
    p.L = c->dL;
    p.V = c->dV;
    p.needs_v = c->d_needs_v;
    return p;
}

inline void drop_slot_ranks(ModSlot &ms) {
    for (int k = 0; k < 2; ++k) {
        if (ms.rank[k]) (void)nmdetail::dev_free(ms.rank[k]);
        if (ms.rank_total[k]) (void)nmdetail::dev_free(ms.rank_total[k]);
        ms.rank[k] = nullptr;
        ms.rank_total[k] = nullptr;
    }
    ms.meth_counts.clear();
    ms.meth_pad = 0xFFFFFFFFu;
}

// The six state planes of a slot come out of ONE allocation (planes[0] owns it): fewer, larger requests to the driver
// (a fresh hipMalloc of recycled memory is scrubbed first, tools/alloc_probe.py).
inline hipError_t alloc_slot_planes(ModSlot &ms, size_t words) {
    if (ms.planes[0]) return hipSuccess;
    uint32_t *base = nullptr;
    const hipError_t e = dev_malloc(&base, words * 4 * 6);
    if (e != hipSuccess) return e;
    for (int k = 0; k < 6; ++k) ms.planes[k] = base + (size_t)k * words;
    return hipSuccess;
}

inline void free_slot_planes(ModSlot &ms) {
    if (ms.planes[0]) (void)dev_free(ms.planes[0]);
    for (auto &p : ms.planes) p = nullptr;
}

inline void drop_ingest_rows(nm_ctx *c) {
    for (int &x : c->ing_slot_of_mod) x = -1;
    c->ing_nconf = c->ing_total_kept = c->ing_classified = 0;
}

// pinned staging ring of the ctx (nmscan.hip): acquire a (device, host) buffer pair of at least `bytes`, and mark it
// busy until the work enqueued so far on the ctx stream has run
inline void wait_set(nm_ctx *c, nm_ctx::Waiting &slot, const nm_ctx::Waiting &v) {
    std::lock_guard<std::mutex> lk(c->wait_mu);
    slot = v;
}
inline void wait_close(nm_ctx *c, nm_ctx::Waiting &slot) {
    std::lock_guard<std::mutex> lk(c->wait_mu);
    slot.open = false;
}
struct WaitCloser {                                     // closes the slots of a collected batch when the *_end half returns, however it returns
    nm_ctx *c;
    nm_ctx::Waiting *a, *b;
    ~WaitCloser() {
        std::lock_guard<std::mutex> lk(c->wait_mu);
        if (a) a->open = false;
        if (b) b->open = false;
    }
};
inline hipError_t sync_flight_streams(nm_ctx *c) {     // window batches of the search's further flights (nmwindows.hip: batch_stream)
    for (hipStream_t s : c->flight_stream)
        if (s) { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) return e; }
    return hipSuccess;
}
int ensure_stage(nm_ctx *c, size_t bytes, int mode = 0);   // 0: pairs 0 / 1 in turn; 1: scoring, all NM_STAGE_RING pairs; 2 + f: the asynchronous window batch's own pair
int release_stage(nm_ctx *c, hipStream_t s = nullptr);   // s: the stream that read the pair (default: c->stream)
int join_lanes(nm_ctx *c);                               // host-side: the second scoring lane has drained
// The tables of a lock-step round are a few KB to a few hundred KB.  As hipMemcpyAsync they go through the SDMA engines: every
// hand-over between a compute queue and an SDMA queue costs ~10 us, and two chains on two streams block each other at the head of
// a shared SDMA queue (round 5: the search's scoring batches waited 100 us for the other chain's copy-back).  These two are plain
// KERNELS on the batch's own stream instead: stage_in copies pinned host memory to the device and clears the outputs in the same
// launch, stage_out writes results to pinned host memory.  bytes are rounded up to 16 (the staging pairs have the slack).
constexpr size_t STAGE_KERNEL_MAX = (size_t)4 << 20;     // larger transfers keep the copy engines
int stage_in(hipStream_t st, const void *h_src, void *d_dst, size_t bytes, void *d_zero, size_t zero_bytes);
int stage_out(hipStream_t st, const void *d_src, void *h_dst, size_t bytes);
// A scoring batch whose candidates are WRITTEN ON THE DEVICE (speculative children of the search, nmwindows.hip): the host knows
// every candidate's (bin, slot) — grouping, ranges, segment table are built as for any batch — but not its motif.  `fill` is
// called with the tables staged and copied to the device, on the batch's stream, before the programs are compiled: it must set
// len / modpos of every record (d_rec[pos_of[k]] is caller's candidate k), write its mask bytes (mask_stride reserved each, at
// d_masks + rec.mask_off) and may lower the candidate count range[entry].y of a group (entry = active index of the slot * n_bins +
// bin).  The candidates must be literal, compact (the canonical base at the modified position) and reach at most 31 positions.
// The writer ALSO compiles the programs (what compile_common_kernel does for a host-written batch; compile_one / common_one of
// nmscan_device.h) and rides in the batch's own staging pair: `extra_in` bytes behind the staged tables (fill_host writes them,
// they travel in the batch's one host-to-device copy), `extra_out` bytes behind the count table (cleared with it by one memset,
// copied back with it by one copy; *h_extra_out says where they land in pinned memory).  `launch` is called on the batch's stream
// with the tables on the device and the outputs cleared, before the scoring kernel.  (The window batch of the search travels
// this way: six commands for window counts + children's counts.)
struct SpecCompile {
    CandRec *rec;
    const uint32_t *pos_of;
    uint8_t *masks;
    uint4 *range;
    uint32_t *programs;
    uint32_t pdw, n_prog;
    int np, wide, fold_modpos, common;        // common: factor the siblings' shared constraints out (common_one), like the batch's kernel expects
};
struct SpecSource {
    uint32_t mask_stride;
    size_t extra_in, extra_out;
    std::function<void(uint8_t *h_in)> fill_host;
    std::function<int(hipStream_t st, const uint8_t *d_in, uint8_t *d_extra_out, const SpecCompile &cc)> launch;
    const void **h_extra_out;
};
// begin: everything enqueued on `st`, counts land in the pinned half of a staging pair noted in c->spec_wait[flight] (and stay in c->d_spec_counts[flight])
int score_batch_spec_begin(nm_ctx *c, int flight, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const SpecSource &spec,
                           hipStream_t st);
// nm_upload_contigs_device with one source offset per contig (nmfasta.hip: records of a device-parsed FASTA)
int upload_contigs_gather(nm_ctx *c, uint32_t n_contigs, const uint64_t *offsets, const uint64_t *src_off, const uint32_t *bin_id,
                          uint32_t n_bins, const uint8_t *d_seq_ascii);
void free_readstats(nm_ctx *c);
// nm_timing_reset(ctx, 2): one event pair around a device phase of the library on the ctx stream (no-ops otherwise); the
// pairs are summed by nm_timing_total_ms together with the scoring launches
void busy_begin(nm_ctx *c);
void busy_end(nm_ctx *c);                          // nmmeth.hip: read statistics are tied to the resident assembly

}  // namespace nmdetail
