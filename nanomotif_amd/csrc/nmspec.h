// nmspec.h — the search's window batch with SPECULATIVE CHILDREN (round 5; internal, not part of the C ABI): what nmsearch.cpp (host
// state machines, no HIP headers) calls and nmwindows.hip implements.  See spec_children_kernel in nmwindows.hip.
#pragma once
#include <cstdint>

#include "../../include/nmscan.h"

#define NM_SEARCH_MAX_FLIGHTS 4     /* groups of tasks the native search may keep in flight (NM_SEARCH_FLIGHTS picks how many) */
#define NM_SEARCH_DEFAULT_FLIGHTS 2

namespace nmdetail {

// per request of the window batch: its (bin, mod slot) on the engine and its search task (row of the background table of spec_setup)
struct WinSpec {
    const uint32_t *req_bin;
    const uint8_t *req_slot;
    const uint32_t *req_search_task;
    uint32_t width, pad;
    double min_kl, freq_threshold;
};
// the tasks' background PSSMs [task][4][width] (float64, rows A T G C) go to the device once per search; n_tasks = 0 drops them
int spec_setup(nm_ctx *c, uint32_t n_tasks, uint32_t width, const double *bg_pssm);
// = nm_win_batch_w_begin + the children's scoring batch behind it on the same stream
// flight (0 or 1): two batches of a kind may be begun and not yet collected, each on a stream of its own — the search keeps two groups
// of tasks going, one on the device while the other is on the host.  spec = NULL: the plain window batch (nm_win_batch_w_begin).
int win_batch_spec_begin(nm_ctx *c, int flight, uint32_t n_req, const uint32_t *req_task, const uint8_t *req_kind, const uint8_t *req_sets, uint32_t ws,
                         const WinSpec *spec);
// = nm_win_batch_w_end + per request {column or -1, base rows A T G C as bits} and four (n_mod, n_nomod) pairs: the children in the
// order of the set bits
int win_batch_spec_end(nm_ctx *c, int flight, uint32_t n_req, int32_t *out, int32_t *spec_info, int64_t *spec_counts);
// nm_score_batch_begin / _end of a flight (its own count table and wait slot)
int score_batch_flight_begin(nm_ctx *c, int flight, uint32_t n_cand, const uint32_t *cand_bin, const uint8_t *cand_mod_slot, const uint8_t *cand_len,
                             const uint8_t *cand_modpos, const uint32_t *cand_mask_offset, const uint8_t *cand_masks);
int score_batch_flight_end(nm_ctx *c, int flight, int64_t *out_counts);

}  // namespace nmdetail
