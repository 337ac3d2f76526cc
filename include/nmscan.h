/*
 * nmscan.h — C ABI of libnmscan.so, the MI355X (gfx950) motif-scan / methylation-count engine.
 *
 * Drop-in boundary.  The reference (MicrobialDarkMatter/nanomotif, pure Python) has no FFI; the seam this
 * library replaces is the Python call
 *
 *     motif_model_bin(pileup, contigs, motif, model, low, high) -> BetaBernoulliModel
 *         nanomotif/find_motifs_bin.py:1265-1283  (per contig: motif_model_contig, :1285-1331;
 *         scan = utils.subseq_indices, utils.py:44-67; count = methylated_motif_occourances, :1234-1263)
 *
 * which is called once per candidate from MotifSearcher.run (:1035, :1128), get_parent_scores (:1400, :1418)
 * and merge_motifs_in_df (:1462, :1473, :1503).  Each entry point below names the reference lines whose
 * work it takes over.  The Python host (nanomotif_amd/engine.py) binds these with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative nm_status; nm_last_error() gives the text
 * for the calling thread.  The caller owns all host buffers (they may be freed when the call returns);
 * the library owns all device memory inside the opaque nm_ctx.  One ctx per GPU / process; calls on one ctx
 * are not re-entrant.  No torch / C++ types cross this boundary.
 */
#ifndef NMSCAN_H
#define NMSCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nm_ctx nm_ctx;

typedef enum nm_status {
    NM_OK = 0,
    NM_EINVAL = -1,      /* bad argument (message says which) */
    NM_EHIP = -2,        /* HIP runtime error */
    NM_ESTATE = -3,      /* call order (e.g. scoring before contigs / pileup were uploaded) */
    NM_ENOMEM = -4,
    NM_ERANGE = -5,      /* motif longer / offset wider than the engine supports (NM_MAX_MOTIF_LEN) */
    NM_EINDEX = -6,      /* nm_bed_*_indexed: the tabix index cannot be used with this pileup (not an index, stale):
                            the caller reads the whole file instead; every other failure is final */
    NM_EDECLINED = -7,   /* a DEVICE parser declines an input its host twin reads (gzip that is not bgzip, rows not grouped by contig, too many
                            rows in unusual number formats): call nm_bed_open / nm_fasta_open instead — not an error of the file */
    NM_ESEQUENCE = -8    /* a FASTA record is empty or holds a letter outside ATGCRYSWKMBDHVN: what DNAsequence asserts (seq.py:53-71) */
} nm_status;

#define NM_MAX_MOTIF_LEN 191  /* stripped motif length; every position within 95 of the modified base */
#define NM_MAX_WIDE_MOTIF_LEN 4095   /* nm_score_batch_wide: candidates of search frames above 191 */
#define NM_MAX_MOD_SLOTS 8    /* pileup classifications resident at once: the reference's 3 mod types (m, a, 21839 — constants.py:29-33), each possibly under two threshold pairs */

/* motif position sets are 4-bit masks: bit0 = A, bit1 = C, bit2 = G, bit3 = T; 15 = '.'/N (any character,
 * including non-ACGT assembly letters — regex '.' semantics of utils.py:61-66). */
#define NM_BASE_A 1
#define NM_BASE_C 2
#define NM_BASE_G 4
#define NM_BASE_T 8

int nm_abi_version(void);
const char *nm_last_error(void);

/* Optional, process-wide: take device memory from the host application's pool instead of hipMalloc / hipFree
 * (alloc returns 0 and sets *ptr; both NULL restores the default).  Set it before the first nm_ctx_create and keep it
 * until the last ctx is destroyed: a block must be freed by the allocator that made it.  A process that already holds
 * a caching pool (PyTorch) thereby spares the library the driver's scrubbing of recycled memory. */
typedef int (*nm_alloc_fn)(void *user, void **ptr, size_t bytes);
typedef int (*nm_free_fn)(void *user, void *ptr);
int nm_set_device_allocator(nm_alloc_fn alloc, nm_free_fn free_fn, void *user);

/* The library's own allocator for that slot, for a process that brings no pool (the command line): blocks of 32 MiB and more stay with
 * the process when they are freed (at most max_idle_bytes of them) and serve the next request they fit — memory another process used
 * before is scrubbed by the driver when it is handed out again, and hipFree synchronises the device.  enable 1: install (as
 * nm_set_device_allocator: while no nm_ctx is alive) or change the limit; 0: uninstall and release the idle blocks (NM_ESTATE while
 * blocks are in use); -1: statistics only.  stats (may be NULL): requests served from the cache, requests that went to hipMalloc,
 * idle bytes, blocks in use.  A block serves requests made with the same current HIP device only; max_idle_bytes is capped at a quarter
 * of the device's memory (idle blocks are invisible to other users of hipMalloc in the process: torch, RCCL) and every idle block goes
 * back to the driver when a request of the cache fails; a block that came back is reused only behind a hipDeviceSynchronize — the
 * implicit synchronisation hipFree would have made — so the caller's contract is hipFree's: no work may still WRITE a block it frees. */
int nm_block_cache(int enable, uint64_t max_idle_bytes, uint64_t stats[4]);
/* The file parsers (nm_fasta_parse_device, nm_bed_parse_device*) move a file through pinned host buffers on the ctx's copy stream; the
 * buffers are kept between calls (at most 160 MB idle; nm_block_cache(0, ...) releases them).  Pinning costs 0.17 ms per MB and the first
 * transfer of a stream 7 - 17 ms (tools/alloc_costs_probe.hip): this call pays both ahead of time — `count` buffers of `bytes_each` pinned
 * and parked in the cache, one small transfer each way on the copy stream — e.g. on the thread that created the ctx, while the caller is
 * still busy elsewhere.  ctx NULL: the buffers only (any thread, any time after the library is loaded: the command line pins them on a
 * second thread while the first one creates the ctx, beside the interpreter's imports — for device 0: the buffers are pinned for the calling
 * thread's current device).  Purely a warm-up: the parsers work without it. */
int nm_warm_file_parsers(nm_ctx *ctx, uint64_t bytes_each, uint32_t count);

/* Create / destroy an engine bound to HIP device `device`. */
int nm_ctx_create(int device, nm_ctx **out);
int nm_ctx_destroy(nm_ctx *ctx);

/* Run all subsequent work of this ctx on `hip_stream` (a hipStream_t, e.g. torch's current stream);
 * NULL restores the ctx's own stream. */
int nm_set_stream(nm_ctx *ctx, void *hip_stream);
/* Scoring lanes.  lanes = 2 (ctx on its own stream only): consecutive nm_score_batch_device calls alternate between two
 * streams, so that INDEPENDENT batches submitted back to back overlap on the device — the next launch fills the compute
 * units the previous one leaves idle while its last workgroups drain, and no launch gap separates them (a shard of a
 * multi-GPU run scores a 10 000-candidate table in ~0.08 ms: the ~20 us between two dependent launches are a quarter
 * of that).  The rounds of ONE search depend on each other (find_motifs_bin.py:957-1182) and gain nothing; whole
 * candidate tables scored step after step, or the tables of different searches, do.  A table is complete after
 * nm_sync, or — for the all-reduce — nm_allreduce_counts_async orders itself after the launch that produced it.
 * Every other entry point waits for the second lane first.  lanes = 1 (default) restores strict stream order. */
int nm_set_score_lanes(nm_ctx *ctx, int lanes);
/* Wait until all scoring work queued on the ctx (both lanes) has finished. */
int nm_sync(nm_ctx *ctx);

/*
 * Assembly upload — replaces the per-call `contig_sequence.sequence` strings handed to motif_model_contig
 * (find_motifs_bin.py:1273-1277; loaded and upper-cased by fasta.py:35-49 / seq.py:53-56).
 *   offsets[n_contigs+1]  byte offsets of each contig inside seq_ascii (offsets[0] = 0)
 *   bin_id[n_contigs]     bin of each contig, 0 <= bin_id < n_bins
 *   seq_ascii             concatenated sequences, any case; non-ACGT letters are kept as "matches only '.'"
 * Packs on the device into bit-sliced planes (2 bits/bp + validity plane), contigs grouped by bin.
 */
int nm_upload_contigs(nm_ctx *ctx, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id,
                      uint32_t n_bins, const uint8_t *seq_ascii);

/* Same, with the concatenated sequences already in device memory (offsets / bin_id stay host arrays). */
int nm_upload_contigs_device(nm_ctx *ctx, uint32_t n_contigs, const uint64_t *offsets, const uint32_t *bin_id,
                             uint32_t n_bins, const uint8_t *d_seq_ascii);

/*
 * Pileup upload for one modification type — replaces the six polars filters executed per candidate per contig
 * (find_motifs_bin.py:1274, 1308-1314): rows are classified ONCE, methylated <=> fraction_mod >= high,
 * unmethylated <=> fraction_mod <= low (float64 compares, same expression as the reference).
 *   mod_slot        0..NM_MAX_MOD_SLOTS-1, chosen by the caller
 *   canonical_base  'A' or 'C' (constants.py MOD_TYPE_TO_CANONICAL)
 *   rows (SoA, post pre-filter): contig_id (index into the uploaded contigs), position (0-based), strand
 *   ('+' / '-'), fraction_mod (= percent / 100, dataload.py:85).  (contig, position, strand) must be unique.
 *   append != 0 adds rows to the slot (streaming in chunks); append == 0 clears the slot first.
 */
int nm_upload_pileup(nm_ctx *ctx, uint32_t mod_slot, uint8_t canonical_base, double low, double high,
                     uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                     const uint8_t *strand, const double *fraction_mod, int append);

/* Same, with the four columns already in device memory (e.g. written by a GPU-side parser or generator). */
int nm_upload_pileup_device(nm_ctx *ctx, uint32_t mod_slot, uint8_t canonical_base, double low, double high,
                            uint64_t n_rows, const uint32_t *d_contig_id, const uint32_t *d_position,
                            const uint8_t *d_strand, const double *d_fraction_mod, int append);

/*
 * Raw pileup ingestion with the reference's three pre-filters evaluated ON THE DEVICE, in the reference's order
 * (find_motifs_bin.py:399-414): Nvalid_cov > 5 (dataload.py:191-200); per (contig, mod code) #(frac > 0.7) / #rows
 * > 1e-4 and #(frac > 0.7) > 50 (:202-226); adjacency: per (contig, strand), mod codes mixed, keep a row iff its
 * fraction equals the maximum over positions p-8..p+8 or is below 0.7 (:228-247).  Surviving rows of the mod codes
 * with slot_of_mod[code] >= 0 are classified into that slot's state planes exactly as nm_upload_pileup does (the slots
 * are cleared first).  The surviving rows with fraction_mod >= high — the input of the window extraction
 * (find_motifs_bin.py:625-661) — are exactly the set bits of the slots' per-strand methylated planes; no list is built
 * (*n_confident is their number).  nm_ingest_results returns the number of surviving rows per (contig, mod code) and,
 * when the four conf_* pointers are given, enumerates those rows from the planes (host-side, by mod code, strand,
 * contig, position; valid until a slot is uploaded again).  The device-side window extraction
 * (nm_win_add_task_contigs) never needs the list.
 *   rows: contig_id = engine contig index or 0xFFFFFFFF for contigs this device does not hold (ignored);
 *   mod_code 0..127 (0 = m, 1 = a, 2 = 21839, others as numbered by the reader: they form their own frequency-filter
 *   groups and take part in the adjacency maximum like in the reference, but only codes 0..7 can be given a slot or are
 *   reported by nm_ingest_results); nvalid_cov as read (col 10); fraction_mod < 0 = null percentage (such a row counts
 *   as a position of its group, dataload.py:216, and is dropped by the adjacency filter).
 *   rows_on_device != 0: the six columns are device pointers.
 */
int nm_ingest_pileup(nm_ctx *ctx, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                     const int8_t *mod_code, const uint8_t *strand, const double *fraction_mod,
                     const int32_t *nvalid_cov, const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8],
                     double low, double high, int rows_on_device, uint64_t *n_kept, uint64_t *n_confident);
/* The same for a pileup that arrives in PARTS (a file too large to hold at once): every contig's rows must lie within
 * one part — the three filters are per contig — and part_contigs[n_part_contigs] lists the engine contig ids whose rows
 * this part holds; the dense adjacency arrays (16 B per bp) then cover those contigs only.  first != 0 clears the slots
 * and the accumulated tables; later parts must repeat the slots / thresholds.  *n_kept and *n_confident are cumulative;
 * nm_ingest_results reports the union. */
int nm_ingest_pileup_part(nm_ctx *ctx, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                          const int8_t *mod_code, const uint8_t *strand, const double *fraction_mod,
                          const int32_t *nvalid_cov, const int32_t slot_of_mod[8], const uint8_t canonical_of_mod[8],
                          double low, double high, int rows_on_device, int first, uint32_t n_part_contigs,
                          const uint32_t *part_contigs, uint64_t *n_kept, uint64_t *n_confident);
int nm_ingest_results(nm_ctx *ctx, uint32_t *conf_contig, uint32_t *conf_position, uint8_t *conf_strand,
                      int8_t *conf_mod, uint64_t capacity, uint32_t *kept_per_contig_mod);

/*
 * Score a batch of candidate motifs — replaces n_cand calls of motif_model_bin(..., BetaBernoulliModel())
 * (find_motifs_bin.py:1265-1283).  Candidate k is the STRIPPED motif (motif.py:213-224) given as
 * cand_len[k] position masks starting at cand_masks[cand_mask_offset[k]], with the modified base at index
 * cand_modpos[k]; it is scored against every contig of bin cand_bin[k] on both strands (forward motif vs
 * '+' rows, reverse complement vs '-' rows, find_motifs_bin.py:1316-1317).
 *   out_counts[2k] = n_mod, out_counts[2k+1] = n_nomod  (what model.update receives, :1320)
 * nm_score_batch writes host memory and returns when done; nm_score_batch_device writes device memory
 * (e.g. a torch int64 tensor's data_ptr) asynchronously on the ctx stream so the caller can all-reduce it
 * with RCCL before reading.  (The call returns once its own batch is uploaded and compiled — tens of microseconds on
 * a side stream while earlier batches are scored; up to four batches are in flight, nm_set_score_lanes lets
 * consecutive ones overlap.)
 */
int nm_score_batch(nm_ctx *ctx, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                   const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                   const uint8_t *cand_masks, int64_t *out_counts);
int nm_score_batch_device(nm_ctx *ctx, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                          const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                          const uint8_t *cand_masks, int64_t *d_out_counts);
/* Candidates of a search frame above 191 (find_motifs_bin.py:110-130 takes any --search_frame_size): the same counts as
 * nm_score_batch for motifs up to NM_MAX_WIDE_MOTIF_LEN positions, any distance from the modified base, lengths and mod
 * positions as uint16.  A site counts only while every specified position lies inside the site's own contig (a regex match
 * never leaves the string, utils.py:44-67).  Plain kernel (plane words fetched per specified position), synchronous,
 * host counts; whole bins on this device only (no per-contig rows).  Not a hot path: nobody runs such frames. */
int nm_score_batch_wide(nm_ctx *ctx, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                        const uint16_t *cand_len, const uint16_t *cand_modpos, const uint32_t *cand_mask_offset,
                        const uint8_t *cand_masks, int64_t *out_counts);
/* nm_score_batch in two halves, for a caller with host work to do while the batch runs (the native search resumes the
 * tasks of a round's window batch under the scoring kernel): _begin returns with the upload, the program compile and
 * the scoring launch enqueued; _end waits for the counts (pinned staging, then out_counts) and must be called before the
 * next _begin (out_counts NULL: wait and drop the batch).  Other calls on the ctx may come in between. */
int nm_score_batch_begin(nm_ctx *ctx, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                         const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                         const uint8_t *cand_masks);
int nm_score_batch_end(nm_ctx *ctx, int64_t *out_counts);

/*
 * Parse n regex-style motif strings (characters A C G T . and [..] sets, the form the reference keeps in
 * Motif.string) into the candidate arrays nm_score_batch takes: strips flanking dots and shifts mod_position
 * exactly like Motif.new_stripped_motif (motif.py:213-224).  text holds the strings back to back,
 * text_offset[n+1] their byte offsets.  No GPU involved.
 */
int nm_parse_motifs(uint32_t n, const char *text, const uint32_t *text_offset, const int32_t *mod_position,
                    uint8_t *out_len, uint8_t *out_modpos, uint32_t *out_mask_offset, uint8_t *out_masks,
                    uint64_t masks_capacity, uint64_t *masks_used);

/*
 * Window engine — the per-expansion work of the greedy search on the device.  The windows around the confidently
 * methylated sites of one (bin, mod type) search (find_motifs_bin.py:635-677) are kept as bit planes over windows;
 * a request then replaces DNAarray.filter_sequence_matches (seq.py:499-524) + DNAarray.pssm (seq.py:526-537):
 *   nm_win_add_task  windows as base-set bytes [n][width] (bit0 A, bit1 C, bit2 G, bit3 T, 15 = N), width <= NM_WIN_MAX_WIDTH
 *                    (the reference accepts any --search_frame_size; here 2 * (frame // 2) + 1 <= 191);
 *                    returns the task id.  nm_win_clear drops all tasks.
 *   nm_win_batch     n_req requests (task, kind, motif as one base-set byte per column in a slot of NM_WIN_MAX_WIDTH bytes):
 *                    kind 0: out = { n_active, 0, counts[4][64] } — number of not-yet-removed windows that match the
 *                            motif and, per column, how many of them carry A / T / G / C (rows in the reference's
 *                            A, T, G, C order; an N window counts for all four) => pssm = counts / n_active;
 *                    kind 1: remove the matching windows (filter_sequence_matches(keep_matches=False),
 *                            find_motifs_bin.py:803); out = { alive before, alive after, ... }.
 *                    out is int32[n_req][2 + 4*64].  Requests of one batch must not mix kinds on one task.
 */
#define NM_WIN_MAX_WIDTH 192
#define NM_WIN_OUT_STRIDE (2 + 4 * NM_WIN_MAX_WIDTH)
int nm_win_clear(nm_ctx *ctx);
int nm_win_add_task(nm_ctx *ctx, uint32_t n_windows, uint32_t width, const uint8_t *sets, uint32_t *task_id);
int nm_win_batch(nm_ctx *ctx, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind,
                 const uint8_t *req_sets, int32_t *out);
/* The same with a caller-chosen width stride ws (>= the width of every task asked about, <= NM_WIN_MAX_WIDTH): req_sets is
 * uint8[n_req][ws], out is int32[n_req][2 + 4 * ws] (counts row r, column j at 2 + r * ws + j) — the default 41-column
 * search moves 258 ints per request instead of 770. */
int nm_win_batch_w(nm_ctx *ctx, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind,
                   const uint8_t *req_sets, uint32_t ws, int32_t *out);
/* the same in two halves (see nm_score_batch_begin): _begin enqueues the batch on its own staging pair, _end collects */
int nm_win_batch_w_begin(nm_ctx *ctx, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets,
                         uint32_t width_stride);
int nm_win_batch_w_end(nm_ctx *ctx, int32_t *out);

/*
 * Window extraction on the device (find_motifs_bin.py:625-686) — the windows never exist as bytes on the host.
 *   nm_win_add_task_rows   the methylation windows of one search, gathered from the resident sequence planes:
 *                          row i = (contig id as uploaded, position, minus strand flag); the window is
 *                          seq[pos-pad : pos+pad+1] (seq.py:170-189), reverse-complemented for minus rows
 *                          (find_motifs_bin.py:655-659).  Every row must satisfy pad < pos < len - pad (the
 *                          reference's edge filter is the caller's job); width = 2*pad+1 <= 64.  A base that is not
 *                          A/C/G/T becomes N — callers must keep contigs with other IUPAC letters on the host path
 *                          (the reference raises KeyError there; nm_assembly_other_letters tells).
 *   nm_methylated_row_counts   out[contig][2] = confidently methylated rows of the slot's pileup (fraction >= high,
 *                          after the pre-filters) on the plus / minus strand with pad < pos < len - pad — the rows
 *                          window extraction uses (find_motifs_bin.py:635-661); they are the bits of the slot's
 *                          methylated-state planes, no row list is kept.
 *   nm_win_add_task_contigs    the methylation windows of one search = all those rows of the listed contigs (per
 *                          contig plus rows then minus rows, ascending); *n_windows = how many.
 *   nm_contig_base_counts  out[contig] = number of positions p in [pad, len-pad) whose base is `base`
 *                          ('A','C','G','T') = len(valid starts) of sample_n_subsequences (seq.py:202-225).
 *   nm_bg_counts           background letter counts of the sampled sub-sequences: sample j = (contig, rank k) stands
 *                          for the k-th (ascending) valid start of that contig; samples are grouped by task,
 *                          task_begin[n_tasks+1] delimits them.  out = int64[n_tasks][4][width], rows A, T, G, C
 *                          (DNAarray.pssm numerators, seq.py:391-422: exact letters only).
 */
int nm_win_add_task_rows(nm_ctx *ctx, uint32_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                         const uint8_t *minus, uint32_t pad, uint32_t *task_id);
int nm_methylated_row_counts(nm_ctx *ctx, uint32_t mod_slot, uint32_t pad, uint64_t *out);
int nm_win_add_task_contigs(nm_ctx *ctx, uint32_t mod_slot, uint32_t n_contigs, const uint32_t *contig_id, uint32_t pad,
                            uint32_t *task_id, uint64_t *n_windows);
int nm_contig_base_counts(nm_ctx *ctx, uint8_t base, uint32_t pad, uint64_t *out);
int nm_bg_counts(nm_ctx *ctx, uint8_t base, uint32_t pad, uint64_t n_samples, const uint32_t *sample_contig,
                 const uint32_t *sample_rank, uint32_t n_tasks, const uint64_t *task_begin, int64_t *out);
/* The same with the samples given as RUNS: run r holds run_count[r] consecutive entries of sample_rank, all on contig
 * run_contig[r] (sample_n_subsequences draws contig after contig, find_motifs_bin.py:640-661); task t owns the runs
 * [task_run_begin[t], task_run_begin[t + 1]).  The per-sample contig column is written on the device. */
int nm_bg_counts_runs(nm_ctx *ctx, uint8_t base, uint32_t pad, uint32_t n_runs, const uint32_t *run_contig, const uint32_t *run_count,
                      const uint32_t *sample_rank, uint32_t n_tasks, const uint32_t *task_run_begin, int64_t *out);
/* Window extraction of ALL (bin, mod type) tasks in one call — find_motifs_bin.py:625-686 for every task.  Task t walks the
 * resident contigs contig_id[task_contig_begin[t] .. task_contig_begin[t + 1]) in the order given (the reference's order:
 * sorted contig name); per contig the background sample is drawn first (n = max(ceil(len * freq), 50) of the valid
 * starts with the canonical base task_base[t] in the middle, CPython's random.sample bit for bit) and then the
 * methylation windows are taken around the confident rows of classification task_slot[t]; the task ends as "None" at the
 * first contig without a window (the contigs up to and including that one have consumed random numbers).  Tasks with the
 * same task_group[t] (non-decreasing) share one generator stream in task order; group g starts from
 * group_init_state[g][625] (shared_init != 0: every group from group_init_state[0] — the reference reseeds every task of
 * a plain pileup alike, find_motifs_bin.py:152-171); final_state = where the last group's stream ends.
 * Outputs per task: task_status 0 = windows made / 1 = None; task_window = window-engine task id; task_n_windows;
 * task_n_bg = background samples; bg_counts int64[n_tasks][4][2*pad+1], rows A, T, G, C (pssm = counts / task_n_bg).
 * One gather launch serves the windows of every task.  With shared_init and at least 32 streams the draws are consumed
 * on the device (one wave per stream over one shared sequence of MT19937 outputs, bit-identical to random.sample's set
 * and pool branches); otherwise on host threads while the gather kernel runs (NM_HOST_DRAWS=1 forces that).  The reference's two
 * ValueErrors come back as NM_EINVAL with their text ("Too many samples requested ...", "Not enough subsequences ..."). */
int nm_plan_windows(nm_ctx *ctx, uint32_t n_tasks, const uint32_t *task_slot, const uint8_t *task_base, const uint32_t *task_group,
                    const uint32_t *task_contig_begin, const uint32_t *contig_id, uint32_t pad, double freq, uint32_t n_groups,
                    const uint32_t *group_init_state, int shared_init, uint8_t *task_status, uint32_t *task_window,
                    uint64_t *task_n_windows, uint64_t *task_n_bg, int64_t *bg_counts, uint32_t final_state[625]);
/* Number of assembly letters that are none of A C G T N (any case) seen by the last nm_upload_contigs. */
int nm_assembly_other_letters(nm_ctx *ctx, uint64_t *n);

/*
 * Hit positions of one candidate on one contig — the four arrays motif_model_contig returns with
 * save_motif_positions=True (find_motifs_bin.py:1322-1329), ascending.  which: 0 = index_meth_fwd,
 * 1 = index_nonmeth_fwd, 2 = index_meth_rev, 3 = index_nonmeth_rev.  Writes at most `capacity` positions to
 * `out`, always stores the full count in *n_out.
 */
int nm_hit_positions(nm_ctx *ctx, uint32_t contig_id, uint32_t mod_slot, uint8_t len, uint8_t modpos,
                     const uint8_t *masks, int which, int64_t *out, uint64_t capacity, uint64_t *n_out);

/* Engine facts for measurement: what[0] = total bp uploaded, [1] = padded bp resident, [2] = bytes of the
 * sequence planes, [3] = bytes of one mod slot's compact state planes, [4] = kernel launches so far,
 * [5] = workgroups of the last scoring launch, [6] = candidates of the last launch scored on the
 * strand-implied ("compact") state, [7] = on the general 4-plane state. */
int nm_stats(nm_ctx *ctx, uint64_t what[8]);

/* Per-CONTIG counters: what nm_score_batch sums over a bin, kept apart per contig — motif_model_contig
 * (find_motifs_bin.py:1285-1331) for every contig of the candidate's bin in one launch; the per-contig motif
 * methylation table binnary builds its contamination / inclusion calls on (main.py:167-178 consumes one row per
 * (contig, motif); there the numbers come from the epymetheus crate).  Candidate k gets
 * row_offset[k + 1] - row_offset[k] rows = the resident contigs of its bin in nm_bin_contigs order; out_counts is
 * int64[row_offset[n_cand]][2] = (n_mod, n_nomod) per (candidate, contig), fwd + rev summed, host memory. */
int nm_score_batch_per_contig(nm_ctx *ctx, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot,
                              const uint8_t *cand_len, const uint8_t *cand_modpos, const uint32_t *cand_mask_offset,
                              const uint8_t *cand_masks, const uint64_t *row_offset, int64_t *out_counts);
/* resident contigs of a bin (nm_upload_contigs indices) in the order of the per-contig rows */
int nm_bin_contigs(nm_ctx *ctx, uint32_t bin, uint32_t *contig_ids, uint32_t capacity, uint32_t *n_contigs);

/* ---- per-contig READ methylation of motifs: the table binnary starts from -------------------------
 * Reference: nanomotif/main.py:142-193 — `contig_methylation = methylation_pattern(pileup, assembly, motifs,
 * min_valid_read_coverage, min_valid_cov_to_diff_fraction = 0.8, output_type = Median | WeightedMean)` from the
 * third-party crate epimetheus-py 0.7.5 (setup.py:38; source not in the reference tree: its published behaviour is
 * restated in oracle/contig_methylation.py, parity UNPINNED), then main.py:193 keeps rows with
 * n_motif_obs * mean_read_cov >= --methylation_threshold.
 *
 * nm_readstats_upload: the pileup records of ONE mod code (bedMethyl columns 1 contig, 2 start, 6 strand, 10 N_valid_cov,
 *   12 N_mod, 17 N_diff; nm_bed_open_counts reads them) -> read-statistics slot `slot`.  Records with
 *   n_valid_cov < min_valid_read_coverage or n_valid_cov / (n_valid_cov + n_diff) < min_valid_cov_to_diff_fraction are
 *   dropped (n_diff NULL: no second filter).  contig_id 0xFFFFFFFF = contig not held by this ctx (ignored).  One call per
 *   slot replaces its contents; *n_kept (may be NULL) = records kept.  Independent of nm_upload_pileup / nm_ingest_pileup.
 * nm_contig_methylation: motif m (stripped, masks as for nm_score_batch; motif_slot[m] = read-statistics slot of its mod
 *   code) against EVERY resident contig i, both strands (forward motif on '+' records, reverse complement on '-'
 *   records, like motif_model_contig): out[m * n_contigs + i] =
 *     n_obs          motif sites that carry a kept record (the row exists in the reference's table iff n_obs > 0),
 *     mean_cov       mean n_valid_cov over those sites,
 *     median         median of the per-site read fractions n_modified / n_valid_cov (mean of the two middle ones for an
 *                    even count) — MethylationOutput.Median,
 *     weighted_mean  sum(n_modified) / sum(n_valid_cov) — MethylationOutput.WeightedMean.
 *   Host arrays of n_motifs * n_contigs entries each. */
int nm_readstats_upload(nm_ctx *ctx, uint32_t slot, uint64_t n_rows, const uint32_t *contig_id, const uint32_t *position,
                        const uint8_t *strand, const int32_t *n_valid_cov, const int32_t *n_modified, const int32_t *n_diff,
                        int32_t min_valid_read_coverage, double min_valid_cov_to_diff_fraction, int rows_on_device, uint64_t *n_kept);
int nm_contig_methylation(nm_ctx *ctx, uint32_t n_motifs, const uint8_t *motif_slot, const uint8_t *motif_len, const uint8_t *motif_modpos,
                          const uint32_t *motif_mask_offset, const uint8_t *motif_masks, uint32_t *out_n_obs, double *out_mean_cov,
                          double *out_median, double *out_weighted_mean);

/* ---- the greedy candidate search of all (bin, mod type) tasks, in lock-step ---------------------------------------
 * find_best_candidates (find_motifs_bin.py:688-839) with MotifSearcher.run (:1026-1182), the KL child generation
 * (:957-1023), get_parent_scores pruning (:1382-1433), the dead-end / remaining-windows stops and the missed-candidate
 * rescue (motif.py:594-607), for EVERY task at once: each round, the open requests of all tasks become one window
 * batch (nm_win_batch: filter_sequence_matches + pssm, or window removal) and one scoring batch (nm_score_batch).
 * Scores are float64 in the reference's operation order (model.py:78-92, :1371-1379); visit order, heap ties and
 * thresholds are the reference's, so graphs and best candidates are identical to the Python path.
 *
 * Task i: windows of width W = 2 * padding + 1 live in window-engine task task_window[i] (nm_win_add_task*), its
 * candidates are scored on bin task_bin[i] with classification task_slot[i]; bg_pssm[i] = float64[4][W] background
 * (rows A, T, G, C, seq.py:391-422); total_windows[i] = windows of the task over all ranks; canonical[i] = 'A' / 'C'.
 * reduce (may be NULL): sums an int64 array over the ranks of a contig-sharded run, called once per batch.
 * nm_search_run_custom runs the same state machine on caller-supplied back ends (the CPU tests drive it with the
 * oracle): score_fn gets n motifs of W characters (A C G T .) with their task index and fills int64[n][2];
 * window_fn gets n requests (kind 0 = pssm, 1 = remove) and fills int32[n][2 + 4 * ws] like nm_win_batch_w, with
 * ws = W rounded up to a multiple of 64 (64 for the default frame).
 *
 * Results (nm_search_result_sizes / _export, then _free): per task either "none" (no graph) or the graph in insertion
 * order — motif characters, raw counts, score, priority, depth, visited, edges as task-local node index pairs — and
 * the best candidates as node indices (the reference's list order, missed candidates appended in string order). */
typedef struct nm_search_params {
    uint32_t padding;                    /* search_frame_size // 2 */
    uint32_t max_dead_ends;              /* 25 */
    uint32_t max_rounds_since_new_best;  /* 30 */
    uint32_t max_motif_length;           /* 25 */
    double min_kl;                       /* --minimum_kl_divergence */
    double score_threshold;              /* --min_motif_score */
    double remaining_threshold;          /* 0.001 */
    double freq_threshold;               /* 0.15 */
} nm_search_params;
typedef struct nm_search_result nm_search_result;
typedef int (*nm_search_score_fn)(void *user, uint32_t n, const uint32_t *task, const char *motifs, int64_t *out_counts);
typedef int (*nm_search_window_fn)(void *user, uint32_t n, const uint32_t *task, const uint8_t *kind, const char *motifs, int32_t *out);
typedef int (*nm_search_reduce_fn)(void *user, int64_t *values, uint64_t n);
int nm_search_run(nm_ctx *ctx, uint32_t n_tasks, const uint32_t *task_bin, const uint32_t *task_slot, const uint32_t *task_window,
                  const nm_search_params *params, const double *bg_pssm, const uint64_t *total_windows, const uint8_t *canonical,
                  nm_search_reduce_fn reduce, void *reduce_user, nm_search_result **out);
int nm_search_run_custom(uint32_t n_tasks, const nm_search_params *params, const double *bg_pssm, const uint64_t *total_windows,
                         const uint8_t *canonical, nm_search_score_fn score_fn, nm_search_window_fn window_fn, void *user,
                         nm_search_result **out);
/* stats[3] = scoring batches (lock-step rounds), candidates scored, window requests */
int nm_search_result_sizes(const nm_search_result *res, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_best, uint64_t stats[3]);
/* Speculative children (round 5, nm_search_run on one GPU with windows of at most 63 columns; NM_SEARCH_NO_SPEC=1 turns it off): the
 * window batch of a round also picks, ON THE DEVICE, the column of the largest KL divergence of every PSSM request and scores the
 * children there in the same chain of launches (find_motifs_bin.py:957-1023, :1116-1135: what the host would ask for in its next
 * round).  The host still makes the reference's pick itself, in float64 the way scipy does, and takes a speculative count only
 * for the very motif it was computed for — results are identical with and without, a differing pick costs the round it would
 * have cost anyway.  stats[3] = lock-step iterations of the run, children answered by the speculation, children asked for after
 * all. */
int nm_search_result_speculation(const nm_search_result *res, uint64_t stats[3]);
/* The search graph of every task as GML text — what the command line leaves as temp/<bin>/motif_graph_<mod>.gml (find_motifs_bin.py:521-535:
 * nx.write_gml of the MotifTree): nodes in insertion order (id, label = the motif without its padding, score, priority, depth, visited),
 * a node's edges in the order they were made, floats as Python's repr.  Task t = (*text)[(*off)[t], (*off)[t + 1]) — empty for a task
 * without a result; *text and *off belong to `res` until nm_search_result_free. */
int nm_search_result_gml(nm_search_result *res, const char **text, const uint64_t **off, uint64_t *n_off);
/* offsets are [n_tasks + 1]; any of the column pointers may be NULL */
int nm_search_result_export(const nm_search_result *res, uint64_t *node_off, uint64_t *edge_off, uint64_t *best_off, uint8_t *task_none,
                            char *node_motif, int64_t *node_counts, double *node_score, double *node_priority, int32_t *node_depth,
                            uint8_t *node_visited, int32_t *edges, int32_t *best);
int nm_search_result_free(nm_search_result *res);
/* ---- post-processing of the searches' best candidates, natively -----------------------------------------------------
 * process_subpileup after the search (find_motifs_bin.py:537-596) for EVERY task of a finished nm_search_run at once:
 * graph nodes that are best candidates (score-descending) -> remove_noisy_motifs (postprocess.py:7-25) -> the clique
 * merge of merge_motifs_in_df (find_motifs_bin.py:1436-1537, motif.py:484-560) -> remove_sub_motifs (postprocess.py:
 * 41-82) -> join_motif_complements (:85-109), with the reference's de-duplication after each of the last three.  The
 * merge stage scores, for all tasks together, (1) every merged motif with its exploded pre-merge variants and (2) every
 * accepted merged motif with its parents: two scoring batches on classification task_merge_slot[i] of bin task_bin[i]
 * (the reference evaluates this stage at 0.3 / 0.7 whatever the CLI thresholds are).  reduce: as in nm_search_run.
 * nm_post_run_custom: the same on a caller-supplied scorer (CPU tests); it gets n motifs as regex-style text
 * (letters, '.', sorted "[..]" groups; text_off[n + 1]) with their mod positions and task indices.
 *
 * Export: one record per (task, stage, row), tasks ascending, stages 0..4 = motifs, -noise, -merge, -sub, -complement
 * (the precleanup tables; a task ends at the first stage that leaves no row), rows in the reference's order.  text holds
 * for record i the motif at [text_off[2i], text_off[2i+1]) and its IUPAC form at [text_off[2i+1], text_off[2i+2]);
 * counts = int64[n][2] (n_mod, n_nomod); complement = record index of the complement row (a stage-3 record of the same
 * task) or -1.  stats[2] = scoring batches, candidates scored. */
typedef struct nm_post_result nm_post_result;
typedef int (*nm_post_score_fn)(void *user, uint32_t n, const uint32_t *task, const char *text, const uint32_t *text_off,
                                const int32_t *mod_position, int64_t *out_counts);
int nm_post_run(nm_ctx *ctx, const nm_search_result *res, const uint32_t *task_bin, const uint32_t *task_merge_slot,
                nm_search_reduce_fn reduce, void *reduce_user, nm_post_result **out);
int nm_post_run_custom(const nm_search_result *res, nm_post_score_fn score_fn, void *user, nm_post_result **out);
/* the same from explicit rows instead of a search result (tests): task t holds rows [row_off[t], row_off[t + 1]) in graph
 * node order — motifs = width characters each (A C G T ., modified base in the middle), counts = int64[n][2], score */
int nm_post_run_rows_custom(uint32_t n_tasks, uint32_t width, const uint64_t *row_off, const char *motifs, const int64_t *counts,
                            const double *score, nm_post_score_fn score_fn, void *user, nm_post_result **out);
int nm_post_sizes(const nm_post_result *post, uint64_t *n_rows, uint64_t *text_bytes, uint64_t stats[2]);
int nm_post_export(const nm_post_result *post, uint32_t *row_task, uint8_t *row_stage, uint64_t *text_off, char *text,
                   int32_t *mod_position, int32_t *mod_position_iupac, int64_t *counts, double *score, int64_t *complement);
/* The per-stage tables of every task as TEXT (the reference writes them with motifs.write_csv, motif.py:891-897 / find_motifs_bin.py:537-596:
 * precleanup-motifs/<bin>-<mod>/motifs[-noise[-merge[-sub[-complement]]]].tsv): columns reference, motif, mod_type, mod_position, score,
 * n_mod, n_nomod, motif_iupac, mod_position_iupac (+ the seven *_complement columns in the last stage), rows sorted by motif, floats as
 * Python's repr — byte for byte what nanomotif_amd.postprocess.format_motifs makes of nm_post_export's rows.  task_reference / task_mod_type:
 * the two constant columns of task t.  Table (t, s) = (*text)[(*off)[5 t + s], (*off)[5 t + s + 1]); a stage without rows is its header.
 * *text and *off belong to `post` until nm_post_free. */
int nm_post_tables(nm_post_result *post, const char *const *task_reference, const char *const *task_mod_type, const char **text,
                   const uint64_t **off, uint64_t *n_off);
int nm_post_free(nm_post_result *post);
/* digamma of a positive integer, the value scipy.special.psi returns bit for bit (Cephes psi; model.py:82-83 only ever
 * evaluates it at alpha, beta, alpha + beta = 5 + counts) — what the native search uses, for hosts that score without SciPy. */
int nm_psi_posint(int64_t n, double *out);

/* ---- multi-GPU exchange: sum of the per-rank count tables ------------------------------------------------------
 * One process per GPU; contigs are sharded over the ranks and counts are sums over contigs
 * (motif_model_bin, find_motifs_bin.py:1273-1283), so a scoring step on N GPUs ends with ONE all-reduce
 * (sum, int64) of the table nm_score_batch[_device] produced — RCCL over xGMI, loaded with dlopen("librccl.so.1").
 *
 * nm_comm_unique_id: rank 0 makes the 128-byte id; the HOST carries it to the other ranks by whatever it has
 *   (MPI, a TCP store, a file, torch.distributed.broadcast_object_list — the library does no rendezvous).
 * nm_comm_init: collective over all `world` ranks; every rank passes the same id and its own rank.  Two ranks on one
 *   device are refused by RCCL and come back as NM_ESTATE with its message.
 * nm_allreduce_counts: d_counts (device, int64[n], e.g. the out_counts of nm_score_batch_device) is summed in place
 *   over all ranks; ordered after the work queued so far on the ctx stream, and later ctx-stream work is ordered
 *   after it.  Asynchronous for the host: synchronise the ctx stream (or copy on it) before reading on the host.
 * nm_allreduce_counts_async + nm_comm_wait: the same collective on the ctx's communication stream WITHOUT blocking
 *   the ctx stream: start the all-reduce of table k (buffer_slot b), queue the scoring of table k+1, and call
 *   nm_comm_wait(ctx, b) before the ctx stream touches table k's buffer again (device-side wait, returns at once).
 * nm_allreduce_counts_host: host int64[n] in, summed host int64[n] out (staged through the ctx; blocking) — for
 *   hosts that keep the tables in host memory (the Python lock-step scheduler).
 * nm_comm_sync: host waits for every outstanding all-reduce.  nm_comm_destroy: also done by nm_ctx_destroy.
 * With two scoring lanes a count table may be handed to the next nm_score_batch_device only after nm_comm_wait on the
 * buffer_slot of its last all-reduce (or after nm_sync when it was not all-reduced): the library orders a launch after
 * the previous launch that wrote the SAME d_out_counts, but it cannot see a collective still reading it. */
#define NM_COMM_ID_BYTES 128
#define NM_COMM_SLOTS 4
int nm_comm_unique_id(uint8_t id[NM_COMM_ID_BYTES]);
int nm_comm_init(nm_ctx *ctx, int rank, int world, const uint8_t id[NM_COMM_ID_BYTES]);
int nm_allreduce_counts(nm_ctx *ctx, int64_t *d_counts, uint64_t n);
int nm_allreduce_counts_async(nm_ctx *ctx, int64_t *d_counts, uint64_t n, int buffer_slot);
int nm_comm_wait(nm_ctx *ctx, int buffer_slot);
int nm_allreduce_counts_host(nm_ctx *ctx, int64_t *counts, uint64_t n);
int nm_comm_sync(nm_ctx *ctx);
/* What the communicator of this ctx reports about itself: info[0] = number of ranks (ncclCommCount; 0 = no communicator),
 * info[1] = this rank (ncclCommUserRank), info[2] = its HIP device (ncclCommCuDevice), info[3] = RCCL version code
 * (ncclGetVersion, 0 = no communicator).  A run can thereby state the world its count tables were REALLY summed over. */
int nm_comm_info(nm_ctx *ctx, int32_t info[4]);
int nm_comm_destroy(nm_ctx *ctx);

/* Device time of the last scoring launch(es) in milliseconds, measured with HIP events on the ctx stream
 * (kernels only, no copies).  Blocks until the launch finished. */
int nm_last_kernel_ms(nm_ctx *ctx, float *ms);

/*
 * Native modkit bedMethyl reader — replaces polars' scan_csv of the 18-column pileup (dataload.py:15-34, 72-100)
 * and the tabix reader of the bgzip path (dataload.py:102-152).  Accepts plain text, gzip and bgzip (BGZF blocks are
 * inflated in parallel, size and CRC-32 of every member checked; nm_bed_open reads everything, nm_bed_open_indexed only the contigs asked for).  Columns kept, struct-of-arrays, in file order: contig id (first-appearance
 * order, names via nm_bed_contig_name), start (col 2), mod code id (col 4: 0 = m, 1 = a, 2 = 21839, other codes
 * numbered 3, 4, ... in first-appearance order, names via nm_bed_mod_code),
 * strand (col 6), fraction_mod = col 11 / 100 (-1 for the null markers "NA" / "null"), Nvalid_cov (col 10, -1 for
 * null).  The column pointers stay valid until nm_bed_close.  threads = 0: one per core, at most 32.
 * STRICT like the fixed 18-column schema the reference reads the file against (PILEUP_SCHEMA, dataload.py:15-34): a line that does
 * not have exactly 18 tab-separated columns, a start that is not a non-negative integer, a strand other than "+" / "-" (the only
 * values the scoring path compares with, find_motifs_bin.py:1308-1314), a coverage that is not an integer or a percentage that
 * is not a number is NM_EINVAL naming the column — a damaged file is refused, never half-read.  The device parser below applies
 * the same rules and the same messages.
 */
typedef struct nm_bed nm_bed;
int nm_bed_open(const char *path, uint32_t threads, nm_bed **out);
/* The same, also keeping N_mod (col 12) and N_diff (col 17) as int32 columns (nm_bed_count_columns): the inputs of the
 * read-methylation table (nm_readstats_upload). */
int nm_bed_open_counts(const char *path, uint32_t threads, nm_bed **out);
int nm_bed_count_columns(nm_bed *bed, const int32_t **n_modified, const int32_t **n_diff);
/* The tabix path of the reference (dataload.py:102-152, find_motifs_bin.py:233-246: the records of a bin's contigs are
 * fetched through the .tbi index): only the BGZF blocks holding the n_contigs wanted contigs (names back to back,
 * name_offset[n_contigs + 1]) are inflated and parsed; contigs absent from the index are skipped and counted.  stats
 * (may be NULL): {bytes inflated, bytes of the compressed file, wanted contigs without an index entry, 0}.  NM_EINDEX
 * "not a tabix index" when tbi_path is not one; NM_EINDEX "does not match the pileup" / "index points ..." / "does not
 * parse" when the regions the index names hold rows of other contigs, do not start at BGZF blocks or start inside a
 * line (a stale index): the caller should then read the whole file (nm_bed_open).  A damaged pileup (NM_EINVAL "corrupt
 * BGZF block", a CRC-32 that is off), NM_EHIP and NM_ENOMEM are NOT index problems and come back as themselves. */
int nm_bed_open_indexed(const char *path, const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset,
                        uint32_t threads, nm_bed **out, uint64_t stats[4]);
/* Host only, no pileup needed: the [begin, end) VIRTUAL offsets (file offset of a BGZF block << 16 | offset inside its text) a
 * tabix index holds for each of the n_contigs names — the metadata pseudo-bin 37450 when the index carries it (htslib's do), else
 * the hull of the sequence's chunks: what pysam's TabixFile.fetch(contig) starts from in the reference (dataload.py:102-152).
 * present[i] = 0 for a name the index does not know (begin = end = 0).  NM_EINDEX when tbi_path is not a tabix index. */
int nm_tabix_regions(const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset, uint64_t *begin,
                     uint64_t *end, uint8_t *present);
int nm_bed_shape(nm_bed *bed, uint64_t *n_rows, uint32_t *n_contigs);
int nm_bed_contig_name(nm_bed *bed, uint32_t i, const char **name);
int nm_bed_mod_code(nm_bed *bed, uint32_t id, const char **code);
int nm_bed_columns(nm_bed *bed, const uint32_t **contig_id, const int64_t **position, const int8_t **mod_type,
                   const uint8_t **strand, const double **fraction_mod, const int64_t **nvalid_cov);
/* The columns in the exact types nm_ingest_pileup[_part] takes, without further copies: contig ids mapped through
 * contig_lut[n_contigs of the file] (engine contig id, or 0xFFFFFFFF for contigs the engine does not hold), positions
 * as uint32, Nvalid_cov as int32 with -1 for rows whose coverage is null (they fall to the coverage filter like in the
 * reference; a null percentage stays fraction_mod = -1).  Releases the 64-bit originals: nm_bed_columns must not be used afterwards. */
int nm_bed_ingest_columns(nm_bed *bed, const uint32_t *contig_lut, uint32_t n_lut, const uint32_t **contig_id,
                          const uint32_t **position, const int8_t **mod_type, const uint8_t **strand,
                          const double **fraction_mod, const int32_t **nvalid_cov);
int nm_bed_close(nm_bed *bed);

/*
 * Device-side bedMethyl parser — the same six columns as nm_bed_open, parsed ON THE GPU for plain-text pileups
 * (dataload.py:15-34, 72-100): the host moves the file through pinned slabs into HBM; kernels find the lines, split the
 * fields and convert the numbers with the host parser's own rules (integers digit by digit; the percentage through the
 * correctly rounded fast path, anything else is handed to the host parser's routines and patched in), contig names
 * become ids through a per-row hash and the runs of equal names (first-appearance order, like nm_bed_contig_name).
 * Every row equals nm_bed_open's bit for bit.  The columns stay in device memory in the types nm_ingest_pileup takes with
 * rows_on_device = 1.  A bgzip file (what the reference recommends, docs/source/required_files.md:21) takes the same path:
 * only the compressed bytes cross PCIe, its BGZF blocks are inflated ON THE DEVICE (one lane per block; NM_BED_HOST_INFLATE=1:
 * by the copy threads, into the pinned slabs) and every block's text is checked against the size and the CRC-32 of its gzip
 * member (NM_EINVAL "corrupt BGZF block ..." — the reference's readers, Python's gzip and htslib, refuse such a file too; so
 * does nm_bed_open).  nm_bed_parse_device_indexed reads only the
 * blocks a tabix index names for the wanted contigs (dataload.py:102-152, find_motifs_bin.py:192-312: the reference fetches a
 * bin's contigs through the .tbi) — arguments, stats and errors as nm_bed_open_indexed; rows equal to it bit for bit.
 * Any other gzip stream is refused (NM_EDECLINED "compressed input ...": use nm_bed_open; so are a pileup whose rows are not grouped by contig and one with more than a million rows in unusual number formats).
 *   nm_bedcols_shape           rows, contigs, runs of equal contig names; times = {seconds in total, seconds copying the file}
 *   nm_bedcols_phase_seconds   {total, moving the file (pread / memcpy into pinned slabs, H2D issue), waiting for the device inflate of
 *                              a bgzip file's slabs (0 for plain text), the rest: line / field kernels, contig tables}; for a bgzip file
 *                              inflated on the device the file is moved by a staging thread BESIDE the rest (round 6): its seconds
 *                              are reported but not part of the sum
 *   nm_bedcols_runs            run_row[n_runs + 1] (first row of each run, then n_rows), run_contig[n_runs] (file contig id)
 *   nm_bedcols_map_contigs     contig column = contig_lut[file contig id] (engine contig id or 0xFFFFFFFF)
 *   nm_bedcols_device_columns  DEVICE pointers; contig_id is valid after nm_bedcols_map_contigs
 */
typedef struct nm_bedcols nm_bedcols;
int nm_bed_parse_device(nm_ctx *ctx, const char *path, uint32_t threads, nm_bedcols **out);
int nm_bed_parse_device_indexed(nm_ctx *ctx, const char *path, const char *tbi_path, uint32_t n_contigs, const char *names,
                                const uint32_t *name_offset, uint32_t threads, nm_bedcols **out, uint64_t stats[4]);
int nm_bedcols_shape(nm_bedcols *cols, uint64_t *n_rows, uint32_t *n_contigs, uint32_t *n_runs, double times[2]);
int nm_bedcols_contig_name(nm_bedcols *cols, uint32_t i, const char **name);
int nm_bedcols_mod_code(nm_bedcols *cols, uint32_t id, const char **code);
int nm_bedcols_runs(nm_bedcols *cols, uint64_t *run_row, uint32_t *run_contig);
int nm_bedcols_map_contigs(nm_bedcols *cols, const uint32_t *contig_lut, uint32_t n_lut);
int nm_bedcols_device_columns(nm_bedcols *cols, const uint32_t **contig_id, const uint32_t **file_contig_id, const uint32_t **position,
                              const int8_t **mod_type, const uint8_t **strand, const double **fraction_mod, const int32_t **nvalid_cov);
int nm_bedcols_phase_seconds(nm_bedcols *cols, double out[4]);
/* nm_bed_parse_device_indexed in two halves (round 5).  nm_bed_plan_indexed is its HOST-ONLY half — the tabix index read, the wanted
 * contigs' regions, the walk over their BGZF blocks (half a second at 1 Gbp) — with no GPU involved: a caller runs it on a thread
 * while the HIP runtime comes up and the assembly is parsed (python -m nanomotif_amd does; find_motifs_bin.py:382-396 needs the .tbi
 * before anything else too).  Arguments, stats and errors as nm_bed_open_indexed (stats[3] = microseconds spent).
 * nm_bed_parse_device_planned does the rest on the device; the plan can be used once or several times and is closed by the caller. */
typedef struct nm_bedplan nm_bedplan;
int nm_bed_plan_indexed(const char *path, const char *tbi_path, uint32_t n_contigs, const char *names, const uint32_t *name_offset, uint32_t threads,
                        nm_bedplan **out, uint64_t stats[4]);
int nm_bed_parse_device_planned(nm_ctx *ctx, nm_bedplan *plan, uint32_t threads, nm_bedcols **out);
int nm_bedplan_close(nm_bedplan *plan);
int nm_bedcols_close(nm_bedcols *cols);
/* Copy `bytes` from device memory the library handed out (e.g. the columns above) to host memory, after the work queued on
 * the ctx stream. */
int nm_device_read(nm_ctx *ctx, void *host_dst, const void *device_src, uint64_t bytes);

/*
 * Native FASTA reader — replaces pyfastx / the line loop of fasta.py:35-49 plus DNAsequence's checks (seq.py:53-71):
 * plain or gzip / bgzip files; record name = first whitespace-delimited token of the header; sequences upper-cased,
 * all records back to back in nm_fasta_sequence; an empty record or a letter outside ATGCRYSWKMBDHVN is an error (the
 * reference asserts).  Records are parsed in parallel.
 */
typedef struct nm_fasta nm_fasta;
int nm_fasta_open(const char *path, uint32_t threads, nm_fasta **out);
int nm_fasta_shape(nm_fasta *fa, uint32_t *n_records, uint64_t *total_bp);
int nm_fasta_record(nm_fasta *fa, uint32_t i, const char **name, uint64_t *offset, uint64_t *length);
int nm_fasta_sequence(nm_fasta *fa, const uint8_t **seq_upper);
int nm_fasta_close(nm_fasta *fa);

/*
 * Device-side FASTA parser (round 5) — the same records as nm_fasta_open, found ON THE GPU for plain-text assemblies
 * (fasta.py:35-49: every record of the file; seq.py:53-71: upper-cased, non-empty, letters of ATGCRYSWKMBDHVN only): the host
 * moves the file through pinned slabs into HBM, kernels find the header lines ('>' at the start of a line), measure every
 * record (bytes that are neither '\n' nor '\r' up to the next header: CRLF files, a last line without newline and lines of
 * any width are the same thing) and write the bases back to back in device memory — byte for byte the array nm_fasta_sequence
 * returns.  Only the header lines come back to the host (record name = first whitespace-delimited token).  Errors as
 * nm_fasta_open: the first record in file order that is empty or holds another letter (NM_ESEQUENCE "DNA sequence must ...").
 * A gzip file is refused (NM_EDECLINED "compressed input ...": use nm_fasta_open).
 *   nm_fastadev_shape            records, bases; times = {seconds in total, seconds a reader thread spent in pread}
 *   nm_fastadev_record           name, offset into the packed sequence, length
 *   nm_fastadev_table            all records at once: the names back to back, each followed by a NUL, and offsets[n_records + 1]
 *   nm_fastadev_sequence_device  DEVICE pointer to the packed upper-case sequence (nm_device_read copies from it)
 *   nm_upload_contigs_fasta      nm_upload_contigs_device for contig i = record[i] of the parsed file: any subset, any order,
 *                                a record more than once (a contig listed under several bins); no host copy of the sequence
 */
typedef struct nm_fastadev nm_fastadev;
int nm_fasta_parse_device(nm_ctx *ctx, const char *path, uint32_t threads, nm_fastadev **out);
int nm_fastadev_shape(nm_fastadev *fa, uint32_t *n_records, uint64_t *total_bp, double times[2]);
int nm_fastadev_record(nm_fastadev *fa, uint32_t i, const char **name, uint64_t *offset, uint64_t *length);
int nm_fastadev_table(nm_fastadev *fa, const char **names, uint64_t *names_bytes, const uint64_t **offsets);
int nm_fastadev_sequence_device(nm_fastadev *fa, const uint8_t **d_seq_upper);
int nm_upload_contigs_fasta(nm_ctx *ctx, nm_fastadev *fa, uint32_t n_contigs, const uint32_t *record, const uint32_t *bin_id, uint32_t n_bins);
int nm_fastadev_close(nm_fastadev *fa);

/*
 * Host helpers of the window-extraction step (no GPU involved).
 * nm_py_random_sample: the indices CPython's random.sample(range(n), k) would return from the MT19937 state
 *   mt_state[0..623] + position mt_state[624] (random.getstate()[1]); the state is advanced in place, so that
 *   random.setstate() afterwards leaves the interpreter's generator where the reference's call would have left it
 *   (the reference samples its background windows this way: seq.py:202-225, find_motifs_bin.py:640-642).
 * nm_window_letter_counts: counts[4][width] (rows A, T, G, C) of exact letters per column over the windows
 *   seq[starts[i] : starts[i] + width] — EqualLengthDNASet.pssm before the division (seq.py:391-422).
 */
int nm_py_random_sample(uint32_t mt_state[625], uint64_t n, uint64_t k, uint32_t *out_indices);
/* m consecutive calls on one generator (the background samples of the contigs of one task), results back to back */
int nm_py_random_sample_many(uint32_t mt_state[625], uint32_t m, const uint64_t *n, const uint64_t *k, uint32_t *out_indices);
/* n_groups independent generator streams (group g starts from init_state[g] and makes the calls group_off[g] ..
 * group_off[g + 1]), drawn on several host threads; results back to back in call order; final_state = the state the last
 * group ends in (where the interpreter's generator stands after a sequential run of the same calls) */
int nm_py_random_sample_groups(uint32_t n_groups, const uint32_t *init_state, const uint64_t *group_off, const uint64_t *n,
                               const uint64_t *k, uint32_t *out_indices, uint32_t final_state[625]);
int nm_window_letter_counts(const uint8_t *seq, uint64_t seq_len, const int64_t *starts, uint64_t n_windows,
                            uint32_t width, int64_t *counts);

/* Per-launch timing over a region: nm_timing_reset(ctx, 1) starts collecting one HIP event pair per scoring
 * launch (no synchronisation per launch); nm_timing_total_ms sums the kernel durations recorded since then and
 * reports how many launches they cover; nm_timing_reset(ctx, 0) stops collecting.  enable = 2 also brackets the other
 * device phases of the library on the ctx stream (the pre-filter kernels of nm_ingest_pileup, the launches of
 * nm_plan_windows, every nm_win_batch): the sum is then the time the GPU was busy for this ctx. */
int nm_timing_reset(nm_ctx *ctx, int enable);
int nm_timing_total_ms(nm_ctx *ctx, double *total_ms, uint64_t *n_launches);
/* The same phases on the device's clock: begin_ms[i] / end_ms[i] of phase i in ms after the FIRST phase `epoch` recorded (ctx itself,
 * or another ctx on the same device: engines that work side by side share one time line that way).  Phases on different streams
 * may overlap — the union of the intervals is the time the device was busy.  capacity 0: only *n (how many there are). */
int nm_timing_intervals(nm_ctx *ctx, nm_ctx *epoch, uint64_t capacity, double *begin_ms, double *end_ms, uint64_t *n);

#ifdef __cplusplus
}
#endif
#endif /* NMSCAN_H */
