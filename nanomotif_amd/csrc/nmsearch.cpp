// nmsearch — the greedy candidate search of ALL (bin, mod type) tasks advanced in lock-step, natively.
//
// Restates, decision for decision, nanomotif/find_motifs_bin.py:688-839 (find_best_candidates: outer loop, pruning by
// parent scores, dead ends, missed candidates), :843-1182 (MotifSearcher: best-first search, KL child generation,
// priority, stale-round stop), :1360-1433 (predictive_evaluation_score, get_parent_scores), model.py:11-92 and
// motif.py:98-125, 160-194, 594-607 — the same algorithm nanomotif_amd/search.py runs as Python coroutines.  With a
// thousand searches open, those coroutines cost about a second of interpreter time per 1 Gbp metagenome while the GPU
// needs milliseconds; here a round is: collect every task's request, ONE window batch (nm_win_batch) and ONE scoring
// batch (nm_score_batch) for all of them, hand the replies back.
//
// Arithmetic is float64 in the reference's own operation order: digamma = scipy.special.psi restated for the integer
// arguments it receives (harmonic sum up to 10, the Cephes asymptotic series above; bit-equal to scipy on this host,
// tests/test_native_search.py), column KL = scipy.stats.entropy's normalise / rel_entr / sum with numpy's axis-0
// order, np.mean with numpy's pairwise blocks.  Heap ties compare (priority, depth, motif string) like Python tuples.
//
// The scoring and window back ends are callbacks, so the same state machine runs against the HIP engine
// (nm_search_run) and, in the CPU tests, against the oracle's scan (nm_search_run_custom).
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/nmscan.h"
#include "nmsearch_internal.h"
#include "nmspec.h"

namespace {

using namespace nmsearch;

constexpr int MAXW = NM_WIN_MAX_WIDTH;
inline uint32_t width_stride(uint32_t W) { return (W + 63u) / 64u * 64u; }      // columns per row of window sets / counts

size_t stripped_length(const std::string &s) {
    size_t lo = 0, hi = s.size();
    while (lo < hi && s[lo] == '.') ++lo;
    while (hi > lo && s[hi - 1] == '.') --hi;
    return hi - lo;
}

// Motif.sub_motif_of (motif.py:57-96) for search-window motifs (letters and dots, same mod position in both)
bool sub_motif_of(const std::string &self, const std::string &other, int modpos) {
    if (self == other) return false;
    auto bounds = [](const std::string &s, size_t &lo, size_t &hi) {
        lo = 0; hi = s.size();
        while (lo < hi && s[lo] == '.') ++lo;
        if (lo == s.size()) { lo = 0; return; }
        while (hi > lo && s[hi - 1] == '.') --hi;
    };
    size_t alo, ahi, blo, bhi;
    bounds(self, alo, ahi);
    bounds(other, blo, bhi);
    const int na = (int)(ahi - alo), nb = (int)(bhi - blo);
    if (na < nb) return false;
    const int off = (modpos - (int)blo) - (modpos - (int)alo);
    if (off > 0) return false;
    for (int i = 0; i < na; ++i) {
        const int k = i + off;
        if (k < 0) continue;
        if (k >= nb) return true;
        const char tb = other[blo + k], ta = self[alo + i];
        if (tb != '.') {
            // _tok_subset for single characters: '.' is a subset of '.' only; letters must be equal
            const bool subset = ta == '.' ? false : ta == tb;
            if (!subset) return false;
        }
    }
    return true;
}

// motif string -> small integer.  Open addressing with the key INLINE — the motif packed at 3 bits per position (two words
// for the default 41-column frame) next to its value: a lookup is one probe sequence in one array.  The std::unordered_map
// of strings this replaces took three dependent cache misses per find (bucket, node, the string's heap block), and with a
// thousand searches taking turns nothing of a task is in cache when its turn comes: 45 % of the state machines' time.
class MotifIndex {
public:
    int find(const std::string &s) const {
        if (cap_ == 0) return -1;
        uint64_t key[MAXK];
        const uint32_t K = pack(s, key);
        const uint64_t *slots = slots_.data();
        for (uint32_t i = (uint32_t)hash(key, K) & (cap_ - 1);; i = (i + 1) & (cap_ - 1)) {
            const uint64_t *e = slots + (size_t)i * (K + 1);
            if (e[K] == 0) return -1;
            bool same = true;
            for (uint32_t k = 0; k < K; ++k) same &= e[k] == key[k];
            if (same) return (int)(e[K] - 1);
        }
    }
    void insert(const std::string &s, int value) {          // s must not be present
        uint64_t key[MAXK];
        const uint32_t K = pack(s, key);
        if (cap_ == 0 || (n_ + 1) * 2 > cap_) grow(K);
        place(key, K, (uint64_t)value + 1);
        n_ += 1;
    }
    void clear() { slots_.clear(); cap_ = 0; n_ = 0; }

private:
    static constexpr uint32_t MAXK = (NM_WIN_MAX_WIDTH * 3 + 63) / 64;
    static uint32_t pack(const std::string &s, uint64_t *key) {
        const uint32_t K = ((uint32_t)s.size() * 3 + 63) / 64;
        for (uint32_t k = 0; k < K; ++k) key[k] = 0;
        for (uint32_t j = 0; j < s.size(); ++j) {
            const char ch = s[j];
            const uint64_t code = ch == '.' ? 0 : ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'T' ? 4 : 5;
            const uint32_t bit = j * 3;
            key[bit >> 6] |= code << (bit & 63);
            if ((bit & 63) > 61) key[(bit >> 6) + 1] |= code >> (64 - (bit & 63));
        }
        return K;
    }
    static uint64_t hash(const uint64_t *key, uint32_t K) {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (uint32_t k = 0; k < K; ++k) h = (h ^ key[k]) * 0xD6E8FEB86659FD93ull, h ^= h >> 32;
        return h;
    }
    void place(const uint64_t *key, uint32_t K, uint64_t stored) {
        for (uint32_t i = (uint32_t)hash(key, K) & (cap_ - 1);; i = (i + 1) & (cap_ - 1)) {
            uint64_t *e = slots_.data() + (size_t)i * (K + 1);
            if (e[K] == 0) {
                for (uint32_t k = 0; k < K; ++k) e[k] = key[k];
                e[K] = stored;
                return;
            }
        }
    }
    void grow(uint32_t K) {
        std::vector<uint64_t> old;
        old.swap(slots_);
        const uint32_t old_cap = cap_;
        cap_ = cap_ ? cap_ * 2 : 64;
        slots_.assign((size_t)cap_ * (K + 1), 0);
        for (uint32_t i = 0; i < old_cap; ++i) {
            const uint64_t *e = old.data() + (size_t)i * (K + 1);
            if (e[K]) place(e, K, e[K]);
        }
    }
    std::vector<uint64_t> slots_;                        // cap_ entries of K key words + (value + 1), 0 = empty
    uint32_t cap_ = 0, n_ = 0;
};

struct Node {
    std::string motif;
    Model model;
    double score = 0, priority = 0;
    int depth = 0;
    bool visited = false;
    uint32_t seen_epoch = 0;                            // MotifSearcher.run's `visited` set of the current outer iteration
    std::vector<int> succ, pred;
};

struct Graph {                                      // MotifTree (motif.py:577-607): insertion-ordered nodes
    std::vector<Node> nodes;
    MotifIndex index;
    int find(const std::string &m) const { return index.find(m); }
    int add(const std::string &m) {
        const int id = (int)nodes.size();
        nodes.emplace_back();
        nodes.back().motif = m;
        index.insert(m, id);
        return id;
    }
    bool has_edge(int u, int v) const { return std::find(nodes[u].succ.begin(), nodes[u].succ.end(), v) != nodes[u].succ.end(); }
    void add_edge(int u, int v) {
        nodes[u].succ.push_back(v);
        nodes[v].pred.push_back(u);
    }
    void reach(int start, bool forward, std::vector<char> &seen) const {
        seen.assign(nodes.size(), 0);
        std::vector<int> stack(forward ? nodes[start].succ : nodes[start].pred);
        while (!stack.empty()) {
            const int n = stack.back();
            stack.pop_back();
            if (seen[n]) continue;
            seen[n] = 1;
            const auto &nx = forward ? nodes[n].succ : nodes[n].pred;
            stack.insert(stack.end(), nx.begin(), nx.end());
        }
        seen[start] = 0;
    }
};

struct Params {
    uint32_t width, padding;
    double min_kl, score_threshold, remaining_threshold, freq_threshold;
    uint32_t max_dead_ends, max_rounds, max_motif_length;
};

enum ReqKind { REQ_NONE = 0, REQ_SCORE, REQ_PSSM, REQ_REMOVE, REQ_DONE };

struct HeapEntry {
    double priority;
    int depth;
    std::string motif;
    int id;                                             // graph node of the motif (not part of the order)
};
struct HeapCmp {                                    // min-heap on the Python tuple (priority, depth, motif)
    bool operator()(const HeapEntry &x, const HeapEntry &y) const {
        if (x.priority != y.priority) return x.priority > y.priority;
        if (x.depth != y.depth) return x.depth > y.depth;
        return x.motif > y.motif;
    }
};

// (the container of the heap, for Task::predict_next: which entry the loop would take next without taking it)
struct Heap : std::priority_queue<HeapEntry, std::vector<HeapEntry>, HeapCmp> {
    const std::vector<HeapEntry> &entries() const { return c; }
};

// The reply to a PSSM request as a task keeps it when it was asked for AHEAD of time (run_tasks: the tail of the search).
struct WinReply {
    int64_t n_active = 0;
    std::vector<int64_t> counts;                    // [4][W]
    int spec_col = -1;
    uint32_t spec_bases = 0;
    int64_t spec_counts[4][2] = {};
};

// One (bin, mod type) search as a resumable state machine.  `resume` consumes the reply to the pending request and
// runs to the next one.
struct Task {
    const Params *P = nullptr;
    const double *bg = nullptr;                     // [4][W] rows A, T, G, C
    uint64_t total = 0;
    char canonical = 'A';
    std::string root;
    Graph g;
    bool graph_made = false;
    std::vector<std::string> best;
    // pending request
    ReqKind req = REQ_NONE;
    std::vector<std::string> req_motifs;            // REQ_SCORE: motifs; REQ_PSSM / REQ_REMOVE: one motif
    std::vector<std::string> ask_all;               // REQ_SCORE: the whole request (req_motifs = the part not in the memo)
    MotifIndex memo;                                // every motif this task has had scored -> memo_models
    std::vector<Model> memo_models;
    bool use_memo = true;
    bool exact_kl = false;                          // NM_SEARCH_EXACT_KL: every KL column in double precision (no pre-selection)
    // replies
    std::vector<Model> rep_models;
    int64_t rep_a = 0, rep_b = 0;                   // pssm: n_active ; remove: before, left
    int64_t rep_counts[4][MAXW];
    // speculative children that came with a PSSM reply (engine back end, nmspec.h): the column the device picked (-1: none), the
    // base rows A T G C it scored there, their counts in the order of the set bits
    int spec_col = -1;
    uint32_t spec_bases = 0;
    int64_t spec_counts[4][2];
    int nb_col = -1;                                // children_of: the column and the rows of `neighbors`
    int nb_row[4];
    uint32_t spec_hits = 0, spec_misses = 0;        // children answered by the speculation / asked for after all
    // PSSM replies that arrived before they were asked for (run_tasks sends the request of the node this search would expand NEXT while
    // the current one is on the device): valid while the task's windows are what they were — until its next REMOVE
    std::unordered_map<std::string, WinReply> win_ahead;
    uint32_t ahead_hits = 0, ahead_sent = 0;
    // --- state
    int state = 0;
    uint32_t dead_ends = 0;
    // MotifSearcher.run
    Heap pq;
    uint32_t epoch = 0;                             // `visited` of MotifSearcher.run = nodes whose seen_epoch is this
    Model root_model;
    double best_score = 0;
    std::string best_guess;
    uint32_t rounds = 0;
    std::string cur;
    int cur_id = -1;
    std::vector<std::string> neighbors, fresh;
    std::vector<int> neighbor_ids;                  // graph nodes of `neighbors` when the children were requested, -1 = new
    // pruning
    std::string guess, temp;
    std::vector<int> to_prune;                      // cumulative set of positions
    std::vector<int> parent_pos;
    std::vector<std::string> parent_motifs;
    Model child_model;
    double mean_parent = 0;
    bool single = false;
    bool result_none = false;

    void init(const Params *p, const double *bg_, uint64_t total_, char canonical_) {
        P = p; bg = bg_; total = total_; canonical = canonical_;
        root.assign(p->width, '.');
        root[p->padding] = canonical_;
        result_none = false;
        use_memo = getenv("NM_SEARCH_NO_MEMO") == nullptr;
        exact_kl = getenv("NM_SEARCH_EXACT_KL") != nullptr;
    }

    // ---- MotifSearcher._motif_child_nodes_kl_dist_max (find_motifs_bin.py:957-1023)
    void children_of(const std::string &motif, int64_t n_active) {
        neighbors.clear();
        const int W = (int)P->width;
        double meth[4][MAXW], kl[MAXW];
        for (int r = 0; r < 4; ++r)
            for (int j = 0; j < W; ++j) meth[r][j] = (double)rep_counts[r][j] / (double)n_active;
        // Only the ARGMAX column (and whether its value reaches min_kl) leaves this function, and the 164 double-precision
        // logarithms of the full vector were 40 % of the state machines' time.  A single-precision estimate with a rigorous
        // error bound sorts out which columns can hold the maximum at all; those — one or two as a rule — are computed the way
        // scipy does, bit for bit, the others only need to stay below them (-inf here).  Anything that is not a plain finite
        // number on the way (a zero background entry: inf, an empty column: NaN) goes the exact way.
        bool any_dot = false;
        double est[MAXW], err[MAXW], sps[MAXW], sqs[MAXW];
        bool must[MAXW];
        double floor_of_max = 0.0;                   // the specified positions hold exact zeros
        bool any_specified = false;
        for (int j = 0; j < W; ++j) {
            if (motif[j] != '.') {                  // specified positions are zeroed out of the KL vector (find_motifs_bin.py:976-980)
                kl[j] = 0.0;
                any_specified = true;
                continue;
            }
            any_dot = true;
            const double sp = ((meth[0][j] + meth[1][j]) + meth[2][j]) + meth[3][j];
            const double sq = ((bg[0 * W + j] + bg[1 * W + j]) + bg[2 * W + j]) + bg[3 * W + j];
            sps[j] = sp;
            sqs[j] = sq;
            double sum = 0.0, mag = 0.0;
            bool plain = true;
            for (int r = 0; r < 4; ++r) {
                const double x = meth[r][j] / sp, y = bg[r * W + j] / sq;
                if (x > 0 && y > 0) {
                    const double t = x * (double)logf((float)(x / y));
                    sum += t;
                    mag += std::fabs(t);
                } else if (!(x == 0 && y >= 0)) {
                    plain = false;                  // inf or NaN in scipy's rel_entr
                }
            }
            must[j] = !plain || !std::isfinite(sum) || exact_kl;
            est[j] = sum;
            err[j] = 1e-6 * (mag + 1.0);            // float division + logf: < 2e-7 relative per term; five-fold margin
        }
        if (!any_dot) return;
        bool have_floor = any_specified;
        for (int j = 0; j < W; ++j)
            if (motif[j] == '.' && !must[j] && (!have_floor || est[j] - err[j] > floor_of_max)) {
                floor_of_max = est[j] - err[j];
                have_floor = true;
            }
        for (int j = 0; j < W; ++j) {
            if (motif[j] != '.') continue;
            if (!must[j] && have_floor && est[j] + err[j] < floor_of_max) {
                kl[j] = -INFINITY;                  // provably below the maximum: cannot win, cannot tie
                continue;
            }
            double e[4];
            for (int r = 0; r < 4; ++r) e[r] = rel_entr(meth[r][j] / sps[j], bg[r * W + j] / sqs[j]);
            kl[j] = ((e[0] + e[1]) + e[2]) + e[3];
        }
        // np.max / np.argmax: NaN propagates and wins, otherwise the first maximum
        int pos = 0;
        bool nan = false;
        for (int j = 0; j < W; ++j) {
            if (std::isnan(kl[j])) { pos = j; nan = true; break; }
            if (kl[j] > kl[pos]) pos = j;
        }
        if (!nan && kl[pos] < P->min_kl) return;
        static const char BASES[4] = {'A', 'T', 'G', 'C'};
        nb_col = pos;
        for (int r = 0; r < 4; ++r) {
            if (meth[r][pos] > bg[r * W + pos] * 0.5 && meth[r][pos] > P->freq_threshold) {
                std::string c = motif;
                c[pos] = BASES[r];
                nb_row[neighbors.size()] = r;
                neighbors.push_back(std::move(c));
            }
        }
    }

    // A new child whose counts came with the PSSM reply goes into the memo: request_score then finds it there, and a request that
    // is answered completely costs no round.  The host's own pick above stays authoritative: a speculative count is used only for
    // the very motif it was computed for (same column, same base) — it is the same kernel's count of the same candidate.
    void take_speculation() {
        if (spec_col < 0 || !use_memo) return;
        for (size_t ni = 0; ni < neighbors.size(); ++ni) {
            if (neighbor_ids[ni] >= 0 || memo.find(neighbors[ni]) >= 0) continue;
            if (spec_col != nb_col || !((spec_bases >> nb_row[ni]) & 1u)) { spec_misses += 1; continue; }
            const int k = __builtin_popcount(spec_bases & ((1u << nb_row[ni]) - 1u));
            memo.insert(neighbors[ni], (int)memo_models.size());
            memo_models.push_back(Model::from_counts(spec_counts[k][0], spec_counts[k][1]));
            spec_hits += 1;
        }
        spec_col = -1;
    }

    // Counts are a pure function of (task, motif): what this task has had scanned before — the root at the start of
    // every outer iteration, the guess and the parents on its search path when pruning starts — is answered from the
    // memo, and a request that is answered completely costs no lock-step round at all.  Returns whether a round is needed.
    bool request_score(std::vector<std::string> motifs) {
        std::vector<std::string> missing;
        for (const auto &m : motifs)
            if ((!use_memo || memo.find(m) < 0) && std::find(missing.begin(), missing.end(), m) == missing.end()) missing.push_back(m);
        ask_all = std::move(motifs);
        req_motifs = std::move(missing);
        if (req_motifs.empty()) {
            rep_models.clear();
            absorb();
            return false;
        }
        req = REQ_SCORE;
        return true;
    }
    void absorb() {                                 // replies of the pending request -> memo; rep_models = the full request's models
        for (size_t k = 0; k < req_motifs.size() && k < rep_models.size(); ++k) {
            const int at = memo.find(req_motifs[k]);
            if (at >= 0) memo_models[at] = rep_models[k];
            else {
                memo.insert(req_motifs[k], (int)memo_models.size());
                memo_models.push_back(rep_models[k]);
            }
        }
        std::vector<Model> full;
        full.reserve(ask_all.size());
        for (const auto &m : ask_all) full.push_back(memo_models[memo.find(m)]);
        rep_models.swap(full);
        ask_all.clear();
    }
    void request_win(ReqKind k, const std::string &m) {
        req = k;
        req_motifs.assign(1, m);
        if (k == REQ_REMOVE) win_ahead.clear();        // the windows change: what was counted ahead is void
    }
    // the reply to ("pssm", m) if it is already here: becomes the current reply, exactly as if it had just arrived
    bool take_ahead(const std::string &m) {
        if (win_ahead.empty()) return false;
        auto it = win_ahead.find(m);
        if (it == win_ahead.end()) return false;
        const WinReply &r = it->second;
        const uint32_t W = P->width;
        rep_a = r.n_active;
        rep_b = 0;
        for (int q = 0; q < 4; ++q)
            for (uint32_t j = 0; j < W; ++j) rep_counts[q][j] = r.counts[(size_t)q * W + j];
        spec_col = r.spec_col;
        spec_bases = r.spec_bases;
        memcpy(spec_counts, r.spec_counts, sizeof spec_counts);
        win_ahead.erase(it);
        ahead_hits += 1;
        return true;
    }
    // The node MotifSearcher.run would expand after the one whose counts are on their way — if none of that one's children gets in
    // front of it: the best entry of the heap that the loop would not skip (find_motifs_bin.py:1040-1052).  False: nothing to predict.
    bool predict_next(std::string &out) const {
        if (state != 2 || req != REQ_PSSM || rounds >= P->max_rounds) return false;
        const HeapEntry *bestp = nullptr;
        HeapCmp worse;
        for (const HeapEntry &e : pq.entries()) {
            const Node &n = g.nodes[e.id];
            if (n.seen_epoch == epoch) continue;
            if (n.model.n_mod() + n.model.n_nomod() < 10) continue;
            if (stripped_length(e.motif) > P->max_motif_length) continue;
            if (!bestp || worse(*bestp, e)) bestp = &e;
        }
        if (!bestp || win_ahead.count(bestp->motif)) return false;
        out = bestp->motif;
        return true;
    }

    // get_parent_scores_co request for `temp` (find_motifs_bin.py:1382-1433)
    bool request_parents() {
        parent_pos.clear();
        parent_motifs.clear();
        for (int i = 0; i < (int)temp.size(); ++i) {
            if (i == (int)P->padding || temp[i] == '.' || temp[i] == 'N') continue;
            std::string q = temp;
            q[i] = '.';
            parent_motifs.push_back(std::move(q));
            parent_pos.push_back(i);
        }
        std::vector<std::string> r;
        r.push_back(temp);
        r.insert(r.end(), parent_motifs.begin(), parent_motifs.end());
        return request_score(std::move(r));
    }

    void finish() {
        // get_missed_candidates + the sub-motif filter + sorted extension (find_motifs_bin.py:826-833, motif.py:594-607)
        if (!graph_made || g.nodes.empty()) { result_none = true; req = REQ_DONE; return; }
        std::vector<char> high(g.nodes.size(), 0), in_best(g.nodes.size(), 0);
        for (size_t i = 0; i < g.nodes.size(); ++i) high[i] = g.nodes[i].score > P->score_threshold;
        for (const auto &b : best) {
            const int id = g.find(b);
            if (id >= 0) in_best[id] = 1;
        }
        std::vector<std::string> missed;
        std::vector<char> seen;
        for (size_t i = 0; i < g.nodes.size(); ++i) {
            if (!high[i] || in_best[i]) continue;
            g.reach((int)i, false, seen);
            bool bad = false;
            for (size_t k = 0; k < seen.size() && !bad; ++k) bad = seen[k] && high[k];
            if (bad) continue;
            g.reach((int)i, true, seen);
            for (size_t k = 0; k < seen.size() && !bad; ++k) bad = seen[k] && in_best[k];
            if (bad) continue;
            const std::string &c = g.nodes[i].motif;
            bool sub_of_any = false, any_sub_of_c = false;
            for (const auto &b : best) {
                sub_of_any |= sub_motif_of(c, b, (int)P->padding);
                any_sub_of_c |= sub_motif_of(b, c, (int)P->padding);
            }
            if (!sub_of_any || !any_sub_of_c) missed.push_back(c);
        }
        std::sort(missed.begin(), missed.end());
        best.insert(best.end(), missed.begin(), missed.end());
        req = REQ_DONE;
    }

    void resume() {
        if (!ask_all.empty()) absorb();
        switch (state) {
            case 0: goto OUTER;
            case 1: goto ROOT_SCORED;
            case 2: goto PSSM_DONE;
            case 3: goto CHILDREN_SCORED;
            case 4: goto PARENTS_SCORED;
            case 5: goto REMOVED;
            default: req = REQ_DONE; return;
        }
    OUTER:
        if (dead_ends >= P->max_dead_ends) { finish(); return; }
        // ---- MotifSearcher.run (find_motifs_bin.py:1026-1182); the graph persists across outer iterations
        best_guess = root;
        if (request_score({root})) {
            state = 1;
            return;
        }
    ROOT_SCORED:
        root_model = rep_models[0];
        best_score = evaluation_score(root_model, root_model);
        rounds = 0;
        epoch += 1;
        graph_made = true;
        if (g.find(root) < 0) {
            const int id = g.add(root);
            g.nodes[id].model = root_model;
            g.nodes[id].score = best_score;
        }
        pq = Heap();
        pq.push(HeapEntry{0.0, 0, root, g.find(root)});
        while (!pq.empty()) {
            cur_id = pq.top().id;
            if (g.nodes[cur_id].seen_epoch == epoch) {
                pq.pop();
                continue;
            }
            cur = pq.top().motif;
            pq.pop();
            {
                const Node &n = g.nodes[cur_id];
                if (n.model.n_mod() + n.model.n_nomod() < 10) continue;
                if (stripped_length(cur) > P->max_motif_length) continue;
            }
            g.nodes[cur_id].seen_epoch = epoch;
            g.nodes[cur_id].visited = true;
            rounds += 1;
            if (!take_ahead(cur)) {
                request_win(REQ_PSSM, cur);
                state = 2;
                return;
            }
        PSSM_DONE:
            if (rep_a == 0) continue;
            children_of(cur, rep_a);
            fresh.clear();
            neighbor_ids.clear();
            for (const auto &m : neighbors) {
                neighbor_ids.push_back(g.find(m));
                if (neighbor_ids.back() < 0) fresh.push_back(m);
            }
            take_speculation();
            if (fresh.empty()) rep_models.clear();
            else if (request_score(fresh)) {
                state = 3;
                return;
            }
        CHILDREN_SCORED:
            {
                const Model cur_model = g.nodes[cur_id].model;
                const int cur_depth = g.nodes[cur_id].depth;
                for (size_t ni = 0; ni < neighbors.size(); ++ni) {
                    const std::string &nxt = neighbors[ni];
                    int id = neighbor_ids[ni];
                    Model nm;
                    if (id >= 0) nm = g.nodes[id].model;
                    else {
                        const size_t k = std::find(fresh.begin(), fresh.end(), nxt) - fresh.begin();
                        nm = rep_models[k];
                    }
                    const double score = evaluation_score(nm, cur_model);
                    const int n_iso = count_isolated(nxt, 1);
                    double d_alpha = 1.0 - ((double)nm.a / (double)root_model.a);
                    double d_beta = (double)nm.b / (double)root_model.b;
                    double priority = d_alpha * d_beta;
                    if (n_iso > 0) priority *= std::pow(10.0, n_iso);
                    if (id >= 0) {
                        if (g.nodes[id].score < score) g.nodes[id].score = score;
                    } else {
                        id = g.add(nxt);
                        Node &n = g.nodes[id];
                        n.model = nm;
                        n.score = score;
                        n.priority = priority;
                        n.depth = cur_depth + 1;
                    }
                    if (!g.has_edge(cur_id, id)) g.add_edge(cur_id, id);
                    if (g.nodes[id].seen_epoch != epoch) pq.push(HeapEntry{g.nodes[id].priority, g.nodes[id].depth, nxt, id});
                    if (score > best_score) {
                        best_score = score;
                        best_guess = nxt;
                        rounds = 0;
                    }
                }
            }
            if (rounds >= P->max_rounds) break;
        }
        // ---- back in find_best_candidates
        guess = best_guess;
        if (guess == root) { finish(); return; }
        temp = guess;
        to_prune.clear();
        single = false;
    PRUNE:
        if (request_parents()) {
            state = 4;
            return;
        }
    PARENTS_SCORED:
        {
            child_model = rep_models[0];
            std::vector<double> scores;
            for (size_t k = 0; k < parent_motifs.size(); ++k) {
                const double s = evaluation_score(child_model, rep_models[1 + k]);
                scores.push_back(s);
                if (s < 0.4 && std::find(to_prune.begin(), to_prune.end(), parent_pos[k]) == to_prune.end()) to_prune.push_back(parent_pos[k]);
            }
            mean_parent = np_mean(scores);
            if (!to_prune.empty()) {
                std::string pruned = temp;
                for (int i : to_prune) pruned[i] = '.';
                size_t specified = 0;
                for (char ch : pruned) specified += ch != '.';
                if (specified == 1) single = true;
                else if (pruned != temp) {
                    temp = pruned;
                    goto PRUNE;
                }
            }
        }
        {
            const int gid = g.find(guess);
            if (single || mean_parent < P->score_threshold || temp == guess) {
                g.nodes[gid].score = mean_parent;
            } else {
                int id = g.find(temp);
                if (id < 0) id = g.add(temp);
                Node &n = g.nodes[id];
                n.model = child_model;
                n.visited = true;
                n.score = mean_parent;
                n.priority = 0;
                n.depth = 0;
                guess = temp;
            }
        }
        request_win(REQ_REMOVE, guess);
        state = 5;
        return;
    REMOVED:
        {
            const int64_t before = rep_a, left = rep_b;
            (void)before;
            if (left == 0) { finish(); return; }
            if (g.nodes[g.find(guess)].score < P->score_threshold) {
                dead_ends += 1;
                goto OUTER;
            }
            best.push_back(guess);
            if ((double)left / (double)total < P->remaining_threshold) { finish(); return; }
        }
        goto OUTER;
    }
};

}  // namespace

struct nm_search_result {
    uint32_t width = 0;
    std::vector<Task> tasks;
    uint64_t rounds = 0, candidates = 0, window_requests = 0;
    uint64_t iterations = 0, spec_hits = 0, spec_misses = 0;     // lock-step iterations; children answered by / asked for despite the speculation
    std::string gml;                      // nm_search_result_gml: the search graphs as GML text, task after task ...
    std::vector<uint64_t> gml_off;        // ... [n_tasks + 1]
};

namespace {

// The state machines of the searches are independent of each other (a Task touches its own data and the constant
// parameters): between two batches they advance on a few host threads.  Workers live for one run_tasks call; a round is
// handed to them through a generation counter, indices are claimed in chunks.
class Workers {
public:
    explicit Workers(unsigned n) {
        for (unsigned i = 1; i < n; ++i) pool_.emplace_back([this] { loop(); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true);
            gen_.fetch_add(1);
        }
        cv_.notify_all();
        for (auto &t : pool_) t.join();
    }
    // fn(i) for i in [0, n): the calling thread takes part; returns when all indices are done.  Two batches of a search
    // are some hundred microseconds apart: a worker spins that long for the next one before it goes to sleep on the
    // condition variable (a futex wake-up per phase cost as much as the phase's work).
    void run(size_t n, const std::function<void(size_t)> &fn) {
        if (pool_.empty() || n < min_parallel_) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        fn_ = &fn;
        n_ = n;
        chunk_ = std::max<size_t>(1, std::min<size_t>(16, n / ((pool_.size() + 1) * 3)));
        next_.store(0, std::memory_order_relaxed);
        busy_.store((unsigned)pool_.size());
        gen_.fetch_add(1);                                   // publishes fn_, n_, next_, busy_
        if (asleep_.load() > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
        drain();
        while (busy_.load(std::memory_order_acquire) != 0) relax();
        fn_ = nullptr;
    }

private:
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void drain() {
        for (;;) {
            const size_t lo = next_.fetch_add(chunk_, std::memory_order_relaxed);
            if (lo >= n_) return;
            const size_t hi = std::min(n_, lo + chunk_);
            for (size_t i = lo; i < hi; ++i) (*fn_)(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (gen_.load() == seen) {
                relax();
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                    std::unique_lock<std::mutex> lk(m_);
                    asleep_.fetch_add(1);
                    cv_.wait(lk, [&] { return gen_.load() != seen; });
                    asleep_.fetch_sub(1);
                    break;
                }
            }
            seen = gen_.load();
            if (stop_.load()) return;
            drain();
            busy_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> pool_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, chunk_ = 16;
    size_t min_parallel_ = getenv("NM_SEARCH_MIN_PARALLEL") ? (size_t)std::max(1, atoi(getenv("NM_SEARCH_MIN_PARALLEL"))) : 8;
    std::atomic<size_t> next_{0};
    std::atomic<unsigned> busy_{0}, asleep_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
};

// The two back ends of a lock-step round.  Callbacks (nm_search_run_custom) answer at once; the engine's halves
// (nm_*_begin / _end) let a round enqueue its window batch and its scoring batch back to back, collect the windows,
// advance the tasks that asked for them WHILE the scoring kernel runs, and only then wait for the counts.
struct Backend {
    nm_search_score_fn score = nullptr;
    nm_search_window_fn window = nullptr;
    // the engine's halves; flight = which of the two groups of tasks the batch belongs to (each has its own batches in flight)
    int (*score_begin)(void *user, int flight, uint32_t n, const uint32_t *task, const char *motifs) = nullptr;
    int (*score_end)(void *user, int flight, uint32_t n, int64_t *out) = nullptr;
    int (*window_begin)(void *user, int flight, uint32_t n, const uint32_t *task, const uint8_t *kind, const char *motifs) = nullptr;
    // also fills spec_info[n][2] and spec_counts[n][4][2] (speculative children, nmspec.h; column -1 where there are none)
    int (*window_end)(void *user, int flight, uint32_t n, int32_t *out, int32_t *spec_info, int64_t *spec_counts) = nullptr;
    uint32_t (*group_of)(void *user, uint32_t task) = nullptr;      // which flight a task travels with (both mod types of a bin together)
    bool threaded_issue = false;                                    // the *_begin halves may run on another thread than the *_end halves
    void *user = nullptr;
};

// One thread that runs issue(f) for the flights handed to it, in the order they were handed over (or the caller itself: threaded = false).
// The hand-over is two atomics per flight; both sides spin (the gaps are tens of microseconds), the thread sleeps after 400 us without work.
class Sender {
public:
    Sender(bool threaded, std::function<int(int)> issue) : issue_(std::move(issue)) {
        for (auto &s : state_) s.store(IDLE);
        if (threaded) th_ = std::thread([this] { loop(); });
    }
    ~Sender() {
        if (!th_.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true);
        }
        cv_.notify_all();
        th_.join();
    }
    void submit(int f) {
        if (!th_.joinable()) { rc_[f] = issue_(f); state_[f].store(DONE); return; }
        state_[f].store(QUEUED, std::memory_order_relaxed);
        queue_[tail_ % QN] = f;
        tail_.store(tail_ + 1, std::memory_order_release);
        if (asleep_.load()) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_all();
        }
    }
    // the flight's batches are out (or failed: the code, with the sending thread's message as this thread's nm_last_error)
    int wait(int f) {
        while (state_[f].load(std::memory_order_acquire) != DONE) relax();
        state_[f].store(IDLE, std::memory_order_relaxed);
        if (rc_[f] && th_.joinable()) return nm_set_error(rc_[f], "%s", err_[f].c_str());
        return rc_[f];
    }

private:
    enum { IDLE = 0, QUEUED = 1, DONE = 2 };
    static constexpr unsigned QN = 2 * NM_SEARCH_MAX_FLIGHTS;
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void loop() {
        unsigned head = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (tail_.load(std::memory_order_acquire) == head) {
                if (stop_.load()) return;
                relax();
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                    std::unique_lock<std::mutex> lk(m_);
                    asleep_.store(1);
                    cv_.wait(lk, [&] { return tail_.load() != head || stop_.load(); });
                    asleep_.store(0);
                }
            }
            const int f = queue_[head % QN];
            head += 1;
            rc_[f] = issue_(f);
            if (rc_[f]) err_[f] = nm_last_error();
            state_[f].store(DONE, std::memory_order_release);
        }
    }
    std::function<int(int)> issue_;
    std::thread th_;
    std::atomic<int> state_[NM_SEARCH_MAX_FLIGHTS];
    int rc_[NM_SEARCH_MAX_FLIGHTS] = {};
    std::string err_[NM_SEARCH_MAX_FLIGHTS];
    int queue_[QN] = {};
    std::atomic<unsigned> tail_{0};
    std::atomic<bool> stop_{false};
    std::atomic<int> asleep_{0};
    std::mutex m_;
    std::condition_variable cv_;
};

int run_tasks(nm_search_result *res, const Params &P, const Backend &B) {
    auto &tasks = res->tasks;
    const uint32_t W = P.width, WS = width_stride(W);
    // NM_SEARCH_TIMING: where the wall time of the lock-step loop goes (stderr, one line)
    const bool timing = getenv("NM_SEARCH_TIMING") != nullptr;
    double t_resume = 0, t_gather = 0, t_window = 0, t_score = 0, t_reply = 0, t_window_begin = 0, t_score_begin = 0, t_sender = 0;   // (t_gather and *_begin: the sending thread's)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // NM_SEARCH_THREADS: host threads that advance the searches between two batches (default: half the hardware threads, at
    // most 12; a phase with fewer than NM_SEARCH_MIN_PARALLEL = 8 tasks runs on the calling thread).  1 Gbp run, 18 485 resumes, 10 ms of them the KL columns
    // of children_of (164 logarithms per expansion, bit-exact with scipy): 24 ms on one thread, 16-17 on four, 14-15 on eight
    // (round 4, same box: 4 threads 51 ms of native search, 8: 46.5, 12: 42.6, 16: 42.1 — the box shows 256 hardware threads and
    // grants 16 CPUs; the default went from 8 to 12)
    unsigned n_threads = std::max(1u, std::min(12u, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("NM_SEARCH_THREADS")) n_threads = (unsigned)std::max(1, std::min(64, atoi(e)));
    if (tasks.size() < 8) n_threads = 1;
    Workers workers(n_threads);
    double t0 = now();
    workers.run(tasks.size(), [&](size_t i) { tasks[i].resume(); });
    t_resume += now() - t0;
    auto resume_these = [&](const std::vector<uint32_t> &which) {
        const double t1 = now();
        workers.run(which.size(), [&](size_t k) {
            Task &t = tasks[which[k]];
            t.req = REQ_NONE;
            t.resume();
        });
        t_resume += now() - t1;
    };
    // Two FLIGHTS (engine back end): the tasks are split into two groups, each with its own window batch and scoring batch; while
    // one group's batches are on the device the other group's replies are taken in, its state machines advanced and its next
    // batches sent.  A task's requests depend only on its own replies, so the grouping changes no result.  Callbacks answer at
    // once: one group.
    int n_flights = 1;
    if (B.window_begin && B.score_begin && tasks.size() >= 16 && getenv("NM_SEARCH_ONE_FLIGHT") == nullptr) {
        n_flights = NM_SEARCH_DEFAULT_FLIGHTS;
        if (const char *e = getenv("NM_SEARCH_FLIGHTS")) n_flights = std::max(1, std::min(NM_SEARCH_MAX_FLIGHTS, atoi(e)));
        n_flights = (int)std::min<size_t>((size_t)n_flights, tasks.size() / 8);
    }
    struct Flight {
        std::vector<uint32_t> members, s_task, w_task, s_owner;
        std::vector<char> s_motifs, w_motifs;
        std::vector<uint8_t> w_kind;
        std::vector<int64_t> counts, spec_counts;
        std::vector<int32_t> wout, spec_info;
        bool flying = false;
        // requests sent AHEAD (below): the tasks, the motifs of the nodes they would expand next, the replies
        std::vector<uint32_t> a_task;
        std::vector<char> a_motifs;
        std::vector<uint8_t> a_kind;
        std::vector<int32_t> a_out, a_info;
        std::vector<int64_t> a_counts;
        int a_slot = -1;
    } fl[NM_SEARCH_MAX_FLIGHTS];
    // THE TAIL: once every flight but one has ended, the last one runs alone — a hundred iterations of a handful of slow searches, one
    // round trip each (a third of the search's time at 1 Gbp).  The ended flight's batch slot is free then: with every PSSM request the
    // last flight also sends, through that slot, the request of the node each search would expand NEXT if none of the current node's
    // children gets in front of it (Task::predict_next).  The reply waits in the task (win_ahead) until the search asks for exactly that
    // motif — then it costs no round trip — or until the task's windows change.  Replies are pure functions of (windows, motif):
    // asking early changes no result.  NM_SEARCH_NO_AHEAD=1: off.
    std::atomic<int> ahead_slot{-1};
    const bool ahead_on = n_flights >= 2 && B.window_begin && B.window_end && B.threaded_issue && getenv("NM_SEARCH_NO_AHEAD") == nullptr;
    constexpr size_t AHEAD_MAX_REQUESTS = 256;
    for (uint32_t i = 0; i < tasks.size(); ++i) fl[(B.group_of ? B.group_of(B.user, i) : i) % (uint32_t)n_flights].members.push_back(i);
    // the requests of a group go out: its window batch first, its scoring batch behind it
    auto issue = [&](int f) -> int {
        Flight &F = fl[f];
        F.s_task.clear(); F.w_task.clear(); F.s_motifs.clear(); F.w_motifs.clear(); F.w_kind.clear(); F.s_owner.clear();
        double t1 = now();
        for (uint32_t i : F.members) {
            Task &t = tasks[i];
            if (t.req == REQ_SCORE) {
                F.s_owner.push_back(i);
                for (const auto &m : t.req_motifs) {
                    F.s_task.push_back(i);
                    F.s_motifs.insert(F.s_motifs.end(), m.begin(), m.end());
                }
            } else if (t.req == REQ_PSSM || t.req == REQ_REMOVE) {
                F.w_task.push_back(i);
                F.w_kind.push_back(t.req == REQ_REMOVE ? 1 : 0);
                F.w_motifs.insert(F.w_motifs.end(), t.req_motifs[0].begin(), t.req_motifs[0].end());
            }
        }
        t_gather += now() - t1;
        F.flying = !(F.s_task.empty() && F.w_task.empty());
        if (!F.flying) return NM_OK;
        res->iterations += 1;
        if (!F.w_task.empty()) {
            t1 = now();
            F.wout.assign(F.w_task.size() * (size_t)(2 + 4 * WS), 0);
            const int rc = B.window_begin ? B.window_begin(B.user, f, (uint32_t)F.w_task.size(), F.w_task.data(), F.w_kind.data(), F.w_motifs.data())
                                          : B.window(B.user, (uint32_t)F.w_task.size(), F.w_task.data(), F.w_kind.data(), F.w_motifs.data(), F.wout.data());
            if (rc) return rc;
            res->window_requests += F.w_task.size();
            t_window_begin += now() - t1;
        }
        if (!F.s_task.empty()) {
            t1 = now();
            F.counts.assign(F.s_task.size() * 2, 0);
            const int rc = B.score_begin ? B.score_begin(B.user, f, (uint32_t)F.s_task.size(), F.s_task.data(), F.s_motifs.data())
                                         : B.score(B.user, (uint32_t)F.s_task.size(), F.s_task.data(), F.s_motifs.data(), F.counts.data());
            if (rc) return rc;
            res->rounds += 1;
            res->candidates += F.s_task.size();
            t_score_begin += now() - t1;
        }
        F.a_slot = -1;
        const int slot = ahead_slot.load(std::memory_order_acquire);
        if (ahead_on && slot >= 0 && slot != f && !F.w_task.empty() && F.w_task.size() <= AHEAD_MAX_REQUESTS) {
            t1 = now();
            F.a_task.clear(); F.a_motifs.clear();
            std::string next;
            for (uint32_t i : F.w_task) {
                Task &t = tasks[i];
                if (!t.predict_next(next)) continue;
                F.a_task.push_back(i);
                F.a_motifs.insert(F.a_motifs.end(), next.begin(), next.end());
                t.ahead_sent += 1;
            }
            if (!F.a_task.empty()) {
                F.a_kind.assign(F.a_task.size(), 0);
                F.a_out.assign(F.a_task.size() * (size_t)(2 + 4 * WS), 0);
                const int rc = B.window_begin(B.user, slot, (uint32_t)F.a_task.size(), F.a_task.data(), F.a_kind.data(), F.a_motifs.data());
                if (rc) return rc;
                F.a_slot = slot;
                res->window_requests += F.a_task.size();
            }
            t_window_begin += now() - t1;
        }
        return NM_OK;
    };
    // the replies of a group: the windows come back first, their tasks advance while the scoring kernel may still be running
    auto collect = [&](int f) -> int {
        Flight &F = fl[f];
        if (!F.flying) return NM_OK;
        F.flying = false;
        if (!F.w_task.empty()) {
            double t1 = now();
            F.spec_info.assign(F.w_task.size() * 2, -1);
            F.spec_counts.assign(F.w_task.size() * 8, 0);
            if (B.window_end) {
                const int rc = B.window_end(B.user, f, (uint32_t)F.w_task.size(), F.wout.data(), F.spec_info.data(), F.spec_counts.data());
                if (rc) return rc;
            }
            if (F.a_slot >= 0) {                         // what was sent ahead has arrived too (its chain ran beside the regular one): into the tasks
                F.a_info.assign(F.a_task.size() * 2, -1);
                F.a_counts.assign(F.a_task.size() * 8, 0);
                const int rc = B.window_end(B.user, F.a_slot, (uint32_t)F.a_task.size(), F.a_out.data(), F.a_info.data(), F.a_counts.data());
                F.a_slot = -1;
                if (rc) return rc;
                for (size_t k = 0; k < F.a_task.size(); ++k) {
                    Task &t = tasks[F.a_task[k]];
                    const int32_t *o = F.a_out.data() + k * (size_t)(2 + 4 * WS);
                    WinReply &r = t.win_ahead[std::string(F.a_motifs.data() + k * (size_t)W, W)];
                    r.n_active = o[0];
                    r.counts.resize(4 * (size_t)W);
                    for (int q = 0; q < 4; ++q)
                        for (uint32_t j = 0; j < W; ++j) r.counts[(size_t)q * W + j] = o[2 + q * WS + j];
                    r.spec_col = F.a_info[2 * k];
                    r.spec_bases = (uint32_t)F.a_info[2 * k + 1];
                    for (int q = 0; q < 4; ++q) { r.spec_counts[q][0] = F.a_counts[8 * k + 2 * q]; r.spec_counts[q][1] = F.a_counts[8 * k + 2 * q + 1]; }
                }
            }
            t_window += now() - t1;
            t1 = now();
            for (size_t k = 0; k < F.w_task.size(); ++k) {
                Task &t = tasks[F.w_task[k]];
                const int32_t *o = F.wout.data() + k * (size_t)(2 + 4 * WS);
                t.rep_a = o[0];
                t.rep_b = o[1];
                t.spec_col = -1;
                if (F.w_kind[k] == 0) {
                    t.spec_col = F.spec_info[2 * k];
                    t.spec_bases = (uint32_t)F.spec_info[2 * k + 1];
                    for (int q = 0; q < 4; ++q) { t.spec_counts[q][0] = F.spec_counts[8 * k + 2 * q]; t.spec_counts[q][1] = F.spec_counts[8 * k + 2 * q + 1]; }
                    for (int r = 0; r < 4; ++r)
                        for (uint32_t j = 0; j < W; ++j) t.rep_counts[r][j] = o[2 + r * WS + j];
                }
            }
            t_reply += now() - t1;
            resume_these(F.w_task);
        }
        if (!F.s_task.empty()) {
            double t1 = now();
            if (B.score_end) {
                const int rc = B.score_end(B.user, f, (uint32_t)F.s_task.size(), F.counts.data());
                if (rc) return rc;
            }
            t_score += now() - t1;
            t1 = now();
            size_t si = 0;
            while (si < F.s_task.size()) {
                Task &t = tasks[F.s_task[si]];
                t.rep_models.clear();
                for (size_t k = 0; k < t.req_motifs.size(); ++k, ++si) t.rep_models.push_back(Model::from_counts(F.counts[2 * si], F.counts[2 * si + 1]));
            }
            t_reply += now() - t1;
            resume_these(F.s_owner);
        }
        return NM_OK;
    };
    // The SENDING THREAD (engine back end with several flights, one GPU): gathering a flight's requests and sending its batches — a
    // dozen HIP calls, 30 us of this thread's time per iteration at 1 Gbp, 8.5 ms of a 33 ms search — happens on a thread of its own
    // while this one waits for, takes in and resumes the next flight.  NM_SEARCH_NO_SENDER=1: everything on this thread.
    const bool threaded = n_flights >= 2 && B.threaded_issue && getenv("NM_SEARCH_NO_SENDER") == nullptr;
    Sender sender(threaded, [&](int f) { return issue(f); });
    bool active[NM_SEARCH_MAX_FLIGHTS] = {};
    for (int f = 0; f < n_flights; ++f) {
        active[f] = true;
        sender.submit(f);
    }
    for (bool any = true; any;) {
        any = false;
        for (int f = 0; f < n_flights; ++f) {
            if (!active[f]) continue;
            const double t1 = now();
            int rc = sender.wait(f);
            t_sender += now() - t1;
            if (rc) return rc;
            if (!fl[f].flying) {                                       // nothing left to ask for: the flight's searches have ended
                active[f] = false;
                int n_active = 0;
                for (int q = 0; q < n_flights; ++q) n_active += active[q];
                if (n_active == 1) ahead_slot.store(f, std::memory_order_release);   // its batch slot serves the last flight's requests ahead
                continue;
            }
            rc = collect(f);
            if (rc) return rc;
            sender.submit(f);
            any = true;
        }
    }
    uint64_t ahead_hits = 0, ahead_sent = 0;
    for (const auto &t : tasks) { res->spec_hits += t.spec_hits; res->spec_misses += t.spec_misses; ahead_hits += t.ahead_hits; ahead_sent += t.ahead_sent; }
    if (timing)
        fprintf(stderr, "[nm_search] %llu lock-step iterations, speculative children: %llu answered, %llu asked for after all; window counts asked for ahead: %llu, used %llu\n",
                (unsigned long long)res->iterations, (unsigned long long)res->spec_hits, (unsigned long long)res->spec_misses,
                (unsigned long long)ahead_sent, (unsigned long long)ahead_hits);
    if (timing)
        fprintf(stderr, "[nm_search] %zu tasks, %llu scoring rounds, %d flights%s: resume %.1f ms, waiting for window batches %.1f ms, for scoring batches %.1f ms, "
                        "replies %.1f ms, waiting for the sender %.1f ms | sending: request gathering %.1f ms, window batches %.1f ms, scoring batches %.1f ms\n",
                tasks.size(), (unsigned long long)res->rounds, n_flights, threaded ? ", batches sent from a thread of their own" : "", t_resume * 1e3,
                t_window * 1e3, t_score * 1e3, t_reply * 1e3, t_sender * 1e3, t_gather * 1e3, t_window_begin * 1e3, t_score_begin * 1e3);
    return NM_OK;
}

int check_params(const nm_search_params *p) {
    if (!p) return nm_set_error(NM_EINVAL, "params is NULL");
    if (p->padding == 0 || 2 * p->padding + 1 > (uint32_t)MAXW) return nm_set_error(NM_ERANGE, "padding %u: windows of 2 * padding + 1 must fit %d positions", p->padding, MAXW);
    return NM_OK;
}

Params make_params(const nm_search_params *p) {
    return Params{2 * p->padding + 1, p->padding, p->min_kl, p->score_threshold, p->remaining_threshold, p->freq_threshold,
                  p->max_dead_ends, p->max_rounds_since_new_best, p->max_motif_length};
}

// engine-backed callbacks
struct EngineUser {
    nm_ctx *ctx;
    const uint32_t *task_bin, *task_slot, *task_win;
    uint32_t width, padding;
    nm_search_reduce_fn reduce;
    void *reduce_user;
    std::vector<uint32_t> bins, offs, wtask;
    std::vector<uint8_t> slots, lens, modpos, masks, sets;
    std::vector<int64_t> tmp64;
    // speculative children (nmspec.h): single-GPU searches whose windows reach at most 31 positions; NM_SEARCH_NO_SPEC=1 turns it off
    bool spec = false;
    double min_kl = 0, freq_threshold = 0;
    std::vector<uint32_t> sbin, stask;
    std::vector<uint8_t> sslot;
};

inline uint8_t set_of(char ch) { return ch == 'A' ? NM_BASE_A : ch == 'C' ? NM_BASE_C : ch == 'G' ? NM_BASE_G : ch == 'T' ? NM_BASE_T : 15; }

int engine_score_begin(void *user, int flight, uint32_t n, const uint32_t *task, const char *motifs) {
    EngineUser &u = *static_cast<EngineUser *>(user);
    const uint32_t W = u.width;
    u.bins.resize(n); u.offs.resize(n); u.slots.resize(n); u.lens.resize(n); u.modpos.resize(n);
    u.masks.clear();
    for (uint32_t i = 0; i < n; ++i) {
        const char *m = motifs + (size_t)i * W;
        uint32_t lo = 0, hi = W;
        while (lo < hi && m[lo] == '.') ++lo;
        while (hi > lo && m[hi - 1] == '.') --hi;
        u.bins[i] = u.task_bin[task[i]];
        u.slots[i] = (uint8_t)u.task_slot[task[i]];
        u.lens[i] = (uint8_t)(hi - lo);
        u.modpos[i] = (uint8_t)(u.padding - lo);
        u.offs[i] = (uint32_t)u.masks.size();
        for (uint32_t j = lo; j < hi; ++j) u.masks.push_back(set_of(m[j]));
    }
    return nmdetail::score_batch_flight_begin(u.ctx, flight, n, u.bins.data(), u.slots.data(), u.lens.data(), u.modpos.data(), u.offs.data(), u.masks.data());
}

int engine_score_end(void *user, int flight, uint32_t n, int64_t *out) {
    EngineUser &u = *static_cast<EngineUser *>(user);
    int rc = nmdetail::score_batch_flight_end(u.ctx, flight, out);
    if (rc) return rc;
    if (u.reduce) rc = u.reduce(u.reduce_user, out, (uint64_t)n * 2);
    return rc;
}

int engine_window_begin(void *user, int flight, uint32_t n, const uint32_t *task, const uint8_t *kind, const char *motifs) {
    EngineUser &u = *static_cast<EngineUser *>(user);
    const uint32_t W = u.width;
    u.wtask.resize(n);
    const uint32_t WS = width_stride(W);
    u.sets.assign((size_t)n * WS, 15);
    for (uint32_t i = 0; i < n; ++i) {
        u.wtask[i] = u.task_win[task[i]];
        for (uint32_t j = 0; j < W; ++j) u.sets[(size_t)i * WS + j] = set_of(motifs[(size_t)i * W + j]);
    }
    if (!u.spec) return nmdetail::win_batch_spec_begin(u.ctx, flight, n, u.wtask.data(), kind, u.sets.data(), WS, nullptr);
    u.sbin.resize(n); u.sslot.resize(n); u.stask.assign(task, task + n);
    for (uint32_t i = 0; i < n; ++i) { u.sbin[i] = u.task_bin[task[i]]; u.sslot[i] = (uint8_t)u.task_slot[task[i]]; }
    const nmdetail::WinSpec ws{u.sbin.data(), u.sslot.data(), u.stask.data(), W, u.padding, u.min_kl, u.freq_threshold};
    return nmdetail::win_batch_spec_begin(u.ctx, flight, n, u.wtask.data(), kind, u.sets.data(), WS, &ws);
}

int engine_window_end(void *user, int flight, uint32_t n, int32_t *out, int32_t *spec_info, int64_t *spec_counts) {
    EngineUser &u = *static_cast<EngineUser *>(user);
    int rc = nmdetail::win_batch_spec_end(u.ctx, flight, n, out, spec_info, spec_counts);
    if (rc) return rc;
    if (u.reduce) {                                  // contig-sharded run: every rank holds the windows of its contigs (no speculation there)
        const size_t m = (size_t)n * (2 + 4 * width_stride(u.width));
        u.tmp64.resize(m);
        for (size_t i = 0; i < m; ++i) u.tmp64[i] = out[i];
        rc = u.reduce(u.reduce_user, u.tmp64.data(), m);
        for (size_t i = 0; i < m; ++i) out[i] = (int32_t)u.tmp64[i];
    }
    return rc;
}

uint32_t engine_group_of(void *user, uint32_t task) { return static_cast<EngineUser *>(user)->task_bin[task]; }

int start(uint32_t n_tasks, const nm_search_params *p, const double *bg_pssm, const uint64_t *total_windows, const uint8_t *canonical,
          nm_search_result **out, Params &P) {
    if (!out) return nm_set_error(NM_EINVAL, "out is NULL");
    *out = nullptr;
    int rc = check_params(p);
    if (rc) return rc;
    if (n_tasks && (!bg_pssm || !total_windows || !canonical)) return nm_set_error(NM_EINVAL, "NULL argument");
    P = make_params(p);
    nm_search_result *res = new (std::nothrow) nm_search_result();
    if (!res) return nm_set_error(NM_ENOMEM, "out of host memory");
    res->width = P.width;
    res->tasks.resize(n_tasks);
    for (uint32_t i = 0; i < n_tasks; ++i) {
        if (canonical[i] != 'A' && canonical[i] != 'C') {
            delete res;
            return nm_set_error(NM_EINVAL, "task %u: canonical base must be 'A' or 'C'", i);
        }
        res->tasks[i].init(&P, bg_pssm + (size_t)i * 4 * P.width, total_windows[i], (char)canonical[i]);
    }
    *out = res;
    return NM_OK;
}

}  // namespace

namespace {

int run_search(uint32_t n_tasks, const nm_search_params *params, const double *bg_pssm, const uint64_t *total_windows, const uint8_t *canonical,
               const Backend &B, nm_search_result **out) {
    Params P;
    int rc = start(n_tasks, params, bg_pssm, total_windows, canonical, out, P);
    if (rc) return rc;
    rc = run_tasks(*out, P, B);
    for (auto &t : (*out)->tasks) t.P = nullptr;
    if (rc) {
        delete *out;
        *out = nullptr;
    }
    return rc;
}

}  // namespace

bool nm_search_task_best(const nm_search_result *res, uint32_t t, std::vector<nmsearch::BestRow> &out) {
    out.clear();
    const Task &T = res->tasks[t];
    if (T.result_none) return false;
    std::vector<char> in_best(T.g.nodes.size(), 0);
    for (const auto &b : T.best) {
        const int id = T.g.find(b);
        if (id >= 0) in_best[id] = 1;
    }
    for (size_t k = 0; k < T.g.nodes.size(); ++k)
        if (in_best[k]) out.push_back(nmsearch::BestRow{T.g.nodes[k].motif, T.g.nodes[k].model, T.g.nodes[k].score});
    return true;
}

uint32_t nm_search_task_count(const nm_search_result *res) { return (uint32_t)res->tasks.size(); }
uint32_t nm_search_width(const nm_search_result *res) { return res->width; }

extern "C" {

int nm_search_run_custom(uint32_t n_tasks, const nm_search_params *params, const double *bg_pssm, const uint64_t *total_windows,
                         const uint8_t *canonical, nm_search_score_fn score_fn, nm_search_window_fn window_fn, void *user,
                         nm_search_result **out) {
    if (!score_fn || !window_fn) return nm_set_error(NM_EINVAL, "NULL callback");
    Backend B;
    B.score = score_fn;
    B.window = window_fn;
    B.user = user;
    return run_search(n_tasks, params, bg_pssm, total_windows, canonical, B, out);
}

int nm_search_run(nm_ctx *ctx, uint32_t n_tasks, const uint32_t *task_bin, const uint32_t *task_slot, const uint32_t *task_window,
                  const nm_search_params *params, const double *bg_pssm, const uint64_t *total_windows, const uint8_t *canonical,
                  nm_search_reduce_fn reduce, void *reduce_user, nm_search_result **out) {
    if (!ctx || (n_tasks && (!task_bin || !task_slot || !task_window))) return nm_set_error(NM_EINVAL, "NULL argument");
    int rc = check_params(params);
    if (rc) return rc;
    EngineUser u{ctx, task_bin, task_slot, task_window, 2 * params->padding + 1, params->padding, reduce, reduce_user, {}, {}, {}, {}, {}, {}, {}, {}, {}};
    // speculative children: every rank of a contig-sharded run sees only its own windows (the counts are summed by `reduce` on the
    // host), so the device cannot pick the column there; windows wider than 63 columns reach beyond the narrow scoring kernels
    u.spec = reduce == nullptr && params->padding <= 31 && n_tasks > 0 && getenv("NM_SEARCH_NO_SPEC") == nullptr;
    u.min_kl = params->min_kl;
    u.freq_threshold = params->freq_threshold;
    if (u.spec) {
        rc = nmdetail::spec_setup(ctx, n_tasks, 2 * params->padding + 1, bg_pssm);
        if (rc) return rc;
    }
    Backend B;
    B.score_begin = engine_score_begin;
    B.score_end = engine_score_end;
    B.window_begin = engine_window_begin;
    B.window_end = engine_window_end;
    B.group_of = engine_group_of;
    B.threaded_issue = u.reduce == nullptr;          // (a contig-sharded run all-reduces inside the *_end halves on the ctx's communicator: one thread)
    B.user = &u;
    rc = run_search(n_tasks, params, bg_pssm, total_windows, canonical, B, out);
    if (rc) {                                        // a round that failed half way: nothing stays open on the ctx
        for (int f = 0; f < NM_SEARCH_MAX_FLIGHTS; ++f) {
            (void)nmdetail::win_batch_spec_end(ctx, f, 0, nullptr, nullptr, nullptr);
            (void)nmdetail::score_batch_flight_end(ctx, f, nullptr);
        }
    }
    if (u.spec) (void)nmdetail::spec_setup(ctx, 0, 0, nullptr);
    return rc;
}

int nm_search_result_sizes(const nm_search_result *res, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_best, uint64_t stats[3]) {
    if (!res || !n_nodes || !n_edges || !n_best) return nm_set_error(NM_EINVAL, "NULL argument");
    uint64_t nn = 0, ne = 0, nb = 0;
    for (const auto &t : res->tasks) {
        if (t.result_none) continue;
        nn += t.g.nodes.size();
        nb += t.best.size();
        for (const auto &n : t.g.nodes) ne += n.succ.size();
    }
    *n_nodes = nn; *n_edges = ne; *n_best = nb;
    if (stats) { stats[0] = res->rounds; stats[1] = res->candidates; stats[2] = res->window_requests; }
    return NM_OK;
}

int nm_search_result_speculation(const nm_search_result *res, uint64_t stats[3]) {
    if (!res || !stats) return nm_set_error(NM_EINVAL, "NULL argument");
    stats[0] = res->iterations;
    stats[1] = res->spec_hits;
    stats[2] = res->spec_misses;
    return NM_OK;
}

int nm_search_result_export(const nm_search_result *res, uint64_t *node_off, uint64_t *edge_off, uint64_t *best_off, uint8_t *task_none,
                            char *node_motif, int64_t *node_counts, double *node_score, double *node_priority, int32_t *node_depth,
                            uint8_t *node_visited, int32_t *edges, int32_t *best) {
    if (!res || !node_off || !edge_off || !best_off || !task_none) return nm_set_error(NM_EINVAL, "NULL argument");
    uint64_t nn = 0, ne = 0, nb = 0;
    const uint32_t W = res->width;
    for (size_t i = 0; i < res->tasks.size(); ++i) {
        const Task &t = res->tasks[i];
        node_off[i] = nn; edge_off[i] = ne; best_off[i] = nb;
        task_none[i] = t.result_none ? 1 : 0;
        if (t.result_none) continue;
        for (size_t k = 0; k < t.g.nodes.size(); ++k) {
            const Node &n = t.g.nodes[k];
            if (node_motif) memcpy(node_motif + (nn + k) * W, n.motif.data(), W);
            if (node_counts) { node_counts[2 * (nn + k)] = n.model.n_mod(); node_counts[2 * (nn + k) + 1] = n.model.n_nomod(); }
            if (node_score) node_score[nn + k] = n.score;
            if (node_priority) node_priority[nn + k] = n.priority;
            if (node_depth) node_depth[nn + k] = n.depth;
            if (node_visited) node_visited[nn + k] = n.visited ? 1 : 0;
            for (int v : n.succ) {
                if (edges) { edges[2 * ne] = (int32_t)k; edges[2 * ne + 1] = v; }
                ++ne;
            }
        }
        for (const auto &b : t.best) {
            if (best) best[nb] = t.g.find(b);
            ++nb;
        }
        nn += t.g.nodes.size();
    }
    node_off[res->tasks.size()] = nn;
    edge_off[res->tasks.size()] = ne;
    best_off[res->tasks.size()] = nb;
    return NM_OK;
}

// The search graph of every task as the GML text the command line leaves under temp/<bin>/motif_graph_<mod>.gml (find_motifs_bin.py:521-535
// writes the networkx graph): nodes in insertion order with id, label (the motif without its padding), score, priority, depth, visited; a
// node's edges in the order they were made; floats as Python's repr.  Task t = (*text)[(*off)[t], (*off)[t + 1]), empty for a task
// without a result.  Byte for byte what nanomotif_amd.native_search.SearchResults.artifacts builds from nm_search_result_export's arrays.
int nm_search_result_gml(nm_search_result *res, const char **text, const uint64_t **off, uint64_t *n_off) {
    if (!res || !text || !off || !n_off) return nm_set_error(NM_EINVAL, "NULL argument");
    std::string &o = res->gml;
    o.clear();
    res->gml_off.assign(res->tasks.size() + 1, 0);
    const uint32_t W = res->width;
    auto put_int = [&](long long v) {
        char b[24];
        const auto r = std::to_chars(b, b + sizeof b, v);
        o.append(b, r.ptr);
    };
    for (size_t i = 0; i < res->tasks.size(); ++i) {
        const Task &t = res->tasks[i];
        res->gml_off[i] = o.size();
        if (t.result_none) continue;
        o += "graph [\n  directed 1\n";
        for (size_t k = 0; k < t.g.nodes.size(); ++k) {
            const Node &n = t.g.nodes[k];
            size_t a = 0, b = W;                                        // label: the motif's characters without the '.' at either end
            while (a < b && n.motif[a] == '.') ++a;
            while (b > a && n.motif[b - 1] == '.') --b;
            o += "  node [\n    id "; put_int((long long)k);
            o += "\n    label \""; o.append(n.motif.data() + a, b - a);
            o += "\"\n    score "; nmsearch::append_py_repr(o, n.score);
            o += "\n    priority "; nmsearch::append_py_repr(o, n.priority);
            o += "\n    depth "; put_int(n.depth);
            o += "\n    visited "; o += n.visited ? '1' : '0';
            o += "\n  ]\n";
        }
        for (size_t k = 0; k < t.g.nodes.size(); ++k)
            for (int v : t.g.nodes[k].succ) {
                o += "  edge [\n    source "; put_int((long long)k);
                o += "\n    target "; put_int(v);
                o += "\n  ]\n";
            }
        o += "]\n";
    }
    res->gml_off.back() = o.size();
    *text = o.data();
    *off = res->gml_off.data();
    *n_off = res->gml_off.size();
    return NM_OK;
}

int nm_psi_posint(int64_t n, double *out) {
    if (!out) return nm_set_error(NM_EINVAL, "out is NULL");
    if (n < 1) return nm_set_error(NM_EINVAL, "nm_psi_posint needs n >= 1, got %lld", (long long)n);
    *out = psi_int((double)n);
    return NM_OK;
}

int nm_search_result_free(nm_search_result *res) {
    delete res;
    return NM_OK;
}

}  // extern "C"
