// nmpost — the post-processing of every (bin, mod type) task's best candidates, natively and in lock-step.
//
// Restates nanomotif/find_motifs_bin.py:537-596 (process_subpileup after the search: noise -> merge -> sub-motifs ->
// complements), postprocess.py:7-109, find_motifs_bin.py:1436-1537 (merge_motifs_in_df) with motif.py:98-158
// (sub_string_of, distance), :268-352 (merge_no_strip), :362-416 (align, explode_with_mask), :484-560 (merge_motifs,
// merge_and_find_new_variants) and the derived IUPAC columns (motif.py:774-818) — what nanomotif_amd/postprocess.py
// runs as Python coroutines (0.04 s of interpreter time for the 1 000 tasks of a 1 Gbp metagenome).  Here the merge
// stage of ALL tasks is two scoring batches: the merged motifs with their pre-merge variants, then the accepted merged
// motifs with their parents.
//
// A motif is a sequence of 4-bit base sets (15 = '.'); its string is the canonical regex form the Python side prints
// (single letters, '.', sorted "[..]" groups), which is also what equality, set membership and every sort in the
// reference compare.  Maximal cliques come out in a fixed order (Bron-Kerbosch, pivot = most neighbours among the
// candidates, ties to the smallest motif); the reference's networkx order depends on hash seeds and nothing downstream
// depends on it.
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iterator>
#include <set>
#include <thread>
#include <unordered_set>

#include "nmsearch_internal.h"

struct nm_ctx;

namespace {

using nmsearch::Model;
using nmsearch::evaluation_score;

constexpr uint8_t DOT = 15;
constexpr int N_STAGES = 5;          // motifs, -noise, -merge, -sub, -complement

struct PMotif {
    std::vector<uint8_t> sets;
    int modpos = 0;
    std::string str;
    bool same(const PMotif &o) const { return modpos == o.modpos && str == o.str; }
};

bool motif_less(const PMotif &a, const PMotif &b) { return a.str != b.str ? a.str < b.str : a.modpos < b.modpos; }

inline bool single_base(uint8_t m) { return m && (m & (m - 1)) == 0; }

std::string token_string(const std::vector<uint8_t> &sets) {          // motif.py set_to_token per position
    std::string s;
    for (uint8_t m : sets) {
        if (m == DOT) s += '.';
        else if (single_base(m)) s += m == NM_BASE_A ? 'A' : m == NM_BASE_C ? 'C' : m == NM_BASE_G ? 'G' : 'T';
        else {
            s += '[';
            if (m & NM_BASE_A) s += 'A';
            if (m & NM_BASE_C) s += 'C';
            if (m & NM_BASE_G) s += 'G';
            if (m & NM_BASE_T) s += 'T';
            s += ']';
        }
    }
    return s;
}

PMotif make_motif(std::vector<uint8_t> sets, int modpos) {
    PMotif m;
    m.sets = std::move(sets);
    m.modpos = modpos;
    m.str = token_string(m.sets);
    return m;
}

PMotif parse_plain(const std::string &s, int modpos) {                 // a search-window motif: A C G T .
    std::vector<uint8_t> sets(s.size());
    for (size_t i = 0; i < s.size(); ++i)
        sets[i] = s[i] == 'A' ? NM_BASE_A : s[i] == 'C' ? NM_BASE_C : s[i] == 'G' ? NM_BASE_G : s[i] == 'T' ? NM_BASE_T : DOT;
    PMotif m;
    m.sets = std::move(sets);
    m.modpos = modpos;
    m.str = s;
    return m;
}

void dot_bounds(const std::vector<uint8_t> &sets, size_t &lo, size_t &hi) {
    lo = 0;
    hi = sets.size();
    while (lo < hi && sets[lo] == DOT) ++lo;
    while (hi > lo && sets[hi - 1] == DOT) --hi;
}

PMotif stripped(const PMotif &m) {                                     // Motif.new_stripped_motif (motif.py:213-224)
    size_t lo, hi;
    dot_bounds(m.sets, lo, hi);
    if (lo == m.sets.size()) return m;                                 // all dots: nothing to strip
    return make_motif(std::vector<uint8_t>(m.sets.begin() + lo, m.sets.begin() + hi), m.modpos - (int)lo);
}

int trimmed_length(const PMotif &m) {
    int n = 0;
    for (uint8_t v : m.sets) n += v != DOT;
    return n;
}

// Motif.distance (motif.py:127-158): positions relative to the modified base, overhangs count their specified positions
int distance(const PMotif &a, const PMotif &b) {
    const int s0 = -a.modpos, s1 = -b.modpos;
    const int e0 = (int)a.sets.size() - a.modpos, e1 = (int)b.sets.size() - b.modpos;
    int d = 0;
    for (int i = std::min(s0, s1); i < std::max(e0, e1); ++i) {
        if (i < s0) d += b.sets[i - s1] != DOT;
        else if (i < s1) d += a.sets[i - s0] != DOT;
        else if (i >= e0) d += b.sets[i - s1] != DOT;
        else if (i >= e1) d += a.sets[i - s0] != DOT;
        else d += a.sets[i - s0] != b.sets[i - s1];
    }
    return d;
}

// set(ta) <= set(tb) on the reference's character sets: '.' is a subset of '.' only, nothing specified is a subset of '.'
inline bool tok_subset(uint8_t a, uint8_t b) {
    if (a == DOT) return b == DOT;
    if (b == DOT) return false;
    return (a & ~b) == 0;
}

bool sub_string_of(const PMotif &self, const PMotif &other) {          // motif.py:98-125
    const PMotif a = stripped(self), b = stripped(other);
    if (a.str == b.str) return false;
    const int na = (int)a.sets.size(), nb = (int)b.sets.size();
    for (int shift = 0; shift < na - nb + 1; ++shift) {
        bool all = true;
        for (int j = 0; j < nb && all; ++j)
            all = j + shift >= na || b.sets[j] == DOT || tok_subset(a.sets[j + shift], b.sets[j]);
        if (all) return true;
    }
    return false;
}

std::string iupac_of(const PMotif &st) {                               // regex_to_iupac of the stripped motif
    static const char TABLE[16] = {'?', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
    std::string s(st.sets.size(), 'N');
    for (size_t i = 0; i < st.sets.size(); ++i) s[i] = TABLE[st.sets[i] & 15];
    return s;
}

std::string iupac_revcomp(const std::string &s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[s.size() - 1 - i];
        switch (c) {
            case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break; case 'C': c = 'G'; break;
            case 'R': c = 'Y'; break; case 'Y': c = 'R'; break; case 'K': c = 'M'; break; case 'M': c = 'K'; break;
            case 'B': c = 'V'; break; case 'V': c = 'B'; break; case 'D': c = 'H'; break; case 'H': c = 'D'; break;
            default: break;                                            // N S W
        }
        r[i] = c;
    }
    return r;
}

struct PRow {
    PMotif m;                        // as stored in the row (search motifs unstripped, merged motifs stripped)
    Model model;
    double score = 0;
    int complement = -1;             // index into the task's rows of the stage before (complement stage only)
    int model_id = -1;               // which model OBJECT the row holds: its index among the models the task created
};

// motifs.unique() (find_motifs_bin.py:570, 579, 588) compares every cell; the `model` cells sit in an Object column and py-polars
// compares those through Python's __hash__ / __eq__ — identity for BetaBernoulliModel (model.py defines neither).  Two rows are
// one only when they hold the SAME model objects: two merge clusters that produce the same motif get a model each
// (:1497-1504) and both rows stay (fixture g13).
bool same_row(const PRow &a, const PRow &b, const std::vector<PRow> *prev) {     // MotifRow.key()
    if (!a.m.same(b.m) || a.model_id != b.model_id || !(a.score == b.score)) return false;
    if ((a.complement < 0) != (b.complement < 0)) return false;
    if (a.complement < 0) return true;
    return (*prev)[a.complement].m.same((*prev)[b.complement].m) && (*prev)[a.complement].model_id == (*prev)[b.complement].model_id;
}

std::vector<PRow> unique_rows(const std::vector<PRow> &rows, const std::vector<PRow> *prev = nullptr) {
    std::vector<PRow> out;
    for (const auto &r : rows) {
        bool seen = false;
        for (const auto &o : out)
            if (same_row(r, o, prev)) { seen = true; break; }
        if (!seen) out.push_back(r);
    }
    return out;
}

// ---- merge_motifs (motif.py:522-560): maximal cliques of the distance <= 2 graph, each merged and exploded
struct Cluster {
    PMotif merged;                   // stripped
    std::vector<int> members;        // indices into the kept motifs
    std::vector<PMotif> pre;         // stripped pre-merge variants, sorted by (string, mod position)
    bool has_new = false;
};

void explode_with_mask(const std::vector<uint8_t> &sets, const std::vector<int> &mask, std::unordered_set<std::string> &out) {
    // motif.py:389-416: '.' -> A C G T, a bracket -> its letters, a letter -> itself, at the mask positions only
    std::string base(sets.size(), '.');
    std::vector<std::string> opts(mask.size());
    size_t total = 1;
    for (size_t k = 0; k < mask.size(); ++k) {
        const uint8_t m = sets[mask[k]];
        if (m & NM_BASE_A) opts[k] += 'A';
        if (m & NM_BASE_C) opts[k] += 'C';
        if (m & NM_BASE_G) opts[k] += 'G';
        if (m & NM_BASE_T) opts[k] += 'T';
        total *= opts[k].size();
    }
    std::vector<size_t> at(mask.size(), 0);
    for (size_t n = 0; n < total; ++n) {
        for (size_t k = 0; k < mask.size(); ++k) base[mask[k]] = opts[k][at[k]];
        out.insert(base);
        for (size_t k = mask.size(); k-- > 0;) {
            if (++at[k] < opts[k].size()) break;
            at[k] = 0;
        }
    }
}

bool merge_cluster(const std::vector<PMotif> &keep, const std::vector<int> &members, Cluster &c) {
    // align_motifs (motif.py:362-387)
    int mx = 0;
    for (int i : members) mx = std::max(mx, keep[i].modpos);
    size_t width = 0;
    std::vector<std::vector<uint8_t>> al;
    for (int i : members) {
        std::vector<uint8_t> s((size_t)(mx - keep[i].modpos), DOT);
        s.insert(s.end(), keep[i].sets.begin(), keep[i].sets.end());
        width = std::max(width, s.size());
        al.push_back(std::move(s));
    }
    for (auto &s : al) s.resize(width, DOT);
    std::vector<int> mask;
    for (size_t j = 0; j < width; ++j) {
        bool any = false;
        for (const auto &s : al) any |= s[j] != DOT;
        if (any) mask.push_back((int)j);
    }
    std::unordered_set<std::string> pre, all_new;
    for (const auto &s : al) explode_with_mask(s, mask, pre);
    // reduce(merge_no_strip): same width and mod position after aligning -> position-wise union, '.' absorbs
    std::vector<uint8_t> merged = al[0];
    for (size_t k = 1; k < al.size(); ++k)
        for (size_t j = 0; j < width; ++j) merged[j] = (merged[j] == DOT || al[k][j] == DOT) ? DOT : (uint8_t)(merged[j] | al[k][j]);
    explode_with_mask(merged, mask, all_new);
    c.has_new = false;
    for (const auto &v : all_new)
        if (!pre.count(v)) { c.has_new = true; break; }
    c.merged = stripped(make_motif(merged, mx));
    if (trimmed_length(c.merged) < 4) return false;
    c.members = members;
    c.pre.clear();
    for (const auto &v : pre) c.pre.push_back(stripped(parse_plain(v, mx)));
    std::sort(c.pre.begin(), c.pre.end(), motif_less);
    c.pre.erase(std::unique(c.pre.begin(), c.pre.end(), [](const PMotif &x, const PMotif &y) { return x.same(y); }), c.pre.end());
    return true;
}

struct Cliques {
    const std::vector<std::vector<char>> &adj;
    std::vector<std::vector<int>> out;
    void expand(std::vector<int> r, std::vector<int> p, std::vector<int> x) {
        if (p.empty() && x.empty()) {
            std::sort(r.begin(), r.end());
            out.push_back(r);
            return;
        }
        int pivot = -1, best = -1;
        auto consider = [&](int u) {
            int n = 0;
            for (int v : p) n += adj[u][v];
            if (n > best || (n == best && u < pivot)) { best = n; pivot = u; }
        };
        for (int u : p) consider(u);
        for (int u : x) consider(u);
        std::vector<int> todo;
        for (int v : p)
            if (!adj[pivot][v]) todo.push_back(v);
        std::sort(todo.begin(), todo.end());
        for (int v : todo) {
            std::vector<int> r2 = r, p2, x2;
            r2.push_back(v);
            for (int u : p) if (adj[v][u]) p2.push_back(u);
            for (int u : x) if (adj[v][u]) x2.push_back(u);
            expand(std::move(r2), std::move(p2), std::move(x2));
            p.erase(std::find(p.begin(), p.end(), v));
            x.push_back(v);
        }
    }
};

std::vector<Cluster> merge_motifs(const std::vector<PMotif> &motifs, std::vector<PMotif> &keep) {
    keep.clear();
    for (const auto &m : motifs) {
        if (trimmed_length(m) <= 4) continue;
        bool dup = false;
        for (const auto &k : keep) dup |= k.same(m);
        if (!dup) keep.push_back(m);
    }
    std::sort(keep.begin(), keep.end(), motif_less);                  // index order = the reference's sorted() order
    const int n = (int)keep.size();
    std::vector<std::vector<char>> adj(n, std::vector<char>(n, 0));
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (distance(keep[i], keep[j]) <= 2) adj[i][j] = adj[j][i] = 1;
    Cliques cq{adj, {}};
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    if (n) cq.expand({}, all, {});
    std::vector<Cluster> res;
    for (const auto &members : cq.out) {
        if (members.size() == 1) continue;
        Cluster c;
        if (merge_cluster(keep, members, c)) res.push_back(std::move(c));
    }
    return res;
}

// ---- one task
struct Request {
    uint32_t task;
    PMotif m;
};

struct PostTask {
    bool none = true;
    std::vector<PRow> stage[N_STAGES];
    int n_stages = 0;                // stages reached with rows
    // merge state
    std::vector<PMotif> keep;
    std::vector<Cluster> clusters;
    std::vector<int> need;           // clusters whose merge creates new variants
    std::vector<size_t> need_at;     // first reply of each
    std::vector<PMotif> accepted;
    std::vector<std::vector<PMotif>> parents;
    std::vector<size_t> accepted_at;
    std::vector<PRow> kept_rows;
    bool merged_any = false;
};

void prepare(PostTask &t, const std::vector<nmsearch::BestRow> &best, int padding, uint32_t task, std::vector<Request> &req) {
    // graph_to_rows: score-descending, ties in node order
    std::vector<PRow> rows;
    for (const auto &b : best) rows.push_back(PRow{parse_plain(b.motif, padding), b.model, b.score, -1, (int)rows.size()});
    std::stable_sort(rows.begin(), rows.end(), [](const PRow &a, const PRow &b) { return -a.score < -b.score; });
    if (rows.empty()) return;
    t.stage[0] = rows;
    t.n_stages = 1;
    // remove_noisy_motifs (postprocess.py:7-25)
    std::set<std::string> clean;
    for (const auto &r : rows)
        if (nmsearch::count_isolated(r.m.str, 3) == 0) clean.insert(r.m.str);
    if (!clean.empty()) {
        std::vector<PRow> kept;
        for (const auto &r : rows)
            if (clean.count(r.m.str)) kept.push_back(r);
        rows.swap(kept);
    }
    if (rows.empty()) return;
    t.stage[1] = rows;
    t.n_stages = 2;
    // merge stage, first batch: every merged motif with its pre-merge variants
    std::vector<PMotif> motifs;
    for (const auto &r : rows) motifs.push_back(r.m);
    t.clusters = merge_motifs(motifs, t.keep);
    for (size_t k = 0; k < t.clusters.size(); ++k) {
        if (!t.clusters[k].has_new) continue;
        t.need.push_back((int)k);
        t.need_at.push_back(req.size());
        req.push_back(Request{task, t.clusters[k].merged});
        for (const auto &p : t.clusters[k].pre) req.push_back(Request{task, p});
    }
}

void decide(PostTask &t, const int64_t *counts, uint32_t task, std::vector<Request> &req) {
    if (t.n_stages < 2) return;
    std::vector<char> verdict(t.clusters.size(), 1);                  // clusters without new variants are accepted as they are
    for (size_t q = 0; q < t.need.size(); ++q) {
        const Cluster &c = t.clusters[t.need[q]];
        const size_t at = t.need_at[q];
        const Model merged = Model::from_counts(counts[2 * at], counts[2 * at + 1]);
        Model pre;
        for (size_t j = 0; j < c.pre.size(); ++j) {
            pre.a += counts[2 * (at + 1 + j)];
            pre.b += counts[2 * (at + 1 + j) + 1];
        }
        verdict[t.need[q]] = evaluation_score(pre, merged) < 0.5;
    }
    std::set<std::string> pre_strings;
    for (size_t k = 0; k < t.clusters.size(); ++k) {
        if (!verdict[k]) continue;
        t.accepted.push_back(t.clusters[k].merged);
        for (int i : t.clusters[k].members) pre_strings.insert(t.keep[i].str);
        t.merged_any = true;
    }
    if (!t.merged_any) return;
    for (const auto &r : t.stage[1])
        if (!pre_strings.count(r.m.str)) t.kept_rows.push_back(r);
    // second batch: get_parent_scores of every accepted merged motif (find_motifs_bin.py:1382-1433)
    for (const auto &m : t.accepted) {
        std::vector<PMotif> ps;
        for (int i = 0; i < (int)m.sets.size(); ++i) {
            if (i == m.modpos || m.sets[i] == DOT) continue;
            std::vector<uint8_t> s = m.sets;
            s[i] = DOT;
            ps.push_back(make_motif(std::move(s), m.modpos));
        }
        t.accepted_at.push_back(req.size());
        req.push_back(Request{task, m});
        for (const auto &p : ps) req.push_back(Request{task, p});
        t.parents.push_back(std::move(ps));
    }
}

void finish(PostTask &t, const int64_t *counts) {
    if (t.n_stages < 2) return;
    std::vector<PRow> rows;
    if (!t.merged_any) rows = t.stage[1];
    else {
        rows = t.kept_rows;
        for (size_t k = 0; k < t.accepted.size(); ++k) {
            const size_t at = t.accepted_at[k];
            const Model child = Model::from_counts(counts[2 * at], counts[2 * at + 1]);
            double score = -1;
            if (!t.parents[k].empty()) {
                std::vector<double> scores;
                for (size_t j = 0; j < t.parents[k].size(); ++j)
                    scores.push_back(evaluation_score(child, Model::from_counts(counts[2 * (at + 1 + j)], counts[2 * (at + 1 + j) + 1])));
                score = nmsearch::np_mean(scores);
            }
            rows.push_back(PRow{t.accepted[k], child, score, -1, (int)(t.stage[0].size() + k)});
        }
    }
    rows = unique_rows(rows);
    if (rows.empty()) return;
    t.stage[2] = rows;
    t.n_stages = 3;
    // remove_sub_motifs (postprocess.py:41-82)
    if (rows.size() >= 2) {
        const std::vector<PRow> group = rows;
        std::vector<std::pair<int, int>> rel;                          // (parent, child) as indices into group
        for (int i = 0; i < (int)group.size(); ++i)
            for (int j = 0; j < (int)group.size(); ++j) {
                if (i == j || !sub_string_of(group[i].m, group[j].m)) continue;
                bool listed = false;                                   // "(m2, m1) not in rel" compares motifs by value
                for (const auto &e : rel) listed |= group[e.first].m.same(group[j].m) && group[e.second].m.same(group[i].m);
                if (!listed) rel.emplace_back(j, i);
            }
        auto model_of = [&](const PMotif &m) {
            for (const auto &r : group)
                if (r.m.same(m)) return r.model;
            return Model{};
        };
        for (const auto &e : rel) {
            const PMotif &parent = group[e.first].m, &child = group[e.second].m;
            const double s = evaluation_score(model_of(child), model_of(parent));
            const PMotif &drop = s > 0.5 ? parent : child;
            std::vector<PRow> left;
            for (const auto &r : rows)
                if (!r.m.same(drop)) left.push_back(r);
            rows.swap(left);
        }
    }
    rows = unique_rows(rows);
    if (rows.empty()) return;
    t.stage[3] = rows;
    t.n_stages = 4;
    // join_motif_complements (postprocess.py:85-109)
    std::vector<std::string> iu(rows.size()), rc(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) {
        iu[i] = iupac_of(stripped(rows[i].m));
        rc[i] = iupac_revcomp(iu[i]);
    }
    std::vector<PRow> joined;
    for (size_t i = 0; i < rows.size(); ++i) {
        bool any = false;
        for (size_t o = 0; o < rows.size(); ++o) {
            if (rc[o] != iu[i]) continue;
            any = true;
            if (iu[i] >= iu[o]) {
                PRow r = rows[i];
                r.complement = (int)o;
                joined.push_back(std::move(r));
            }
        }
        if (!any) joined.push_back(rows[i]);
    }
    joined = unique_rows(joined, &t.stage[3]);
    if (joined.empty()) return;
    t.stage[4] = joined;
    t.n_stages = 5;
}

}  // namespace

struct nm_post_result {
    std::vector<PostTask> tasks;
    uint64_t batches = 0, candidates = 0;
    std::string tables;                   // nm_post_tables: the stage tables' text, back to back ...
    std::vector<uint64_t> table_off;      // ... and where table (task, stage) begins: [task * 5 + stage], one more at the end
};

namespace {

typedef int (*score_requests_fn)(void *user, const std::vector<Request> &req, std::vector<int64_t> &counts);

typedef std::function<bool(uint32_t, std::vector<nmsearch::BestRow> &)> best_rows_fn;

int run_post(uint32_t n, int padding, const best_rows_fn &task_best, score_requests_fn score, void *user, nm_post_result **out) {
    if (!out) return nm_set_error(NM_EINVAL, "NULL argument");
    *out = nullptr;
    nm_post_result *pr = new (std::nothrow) nm_post_result();
    if (!pr) return nm_set_error(NM_ENOMEM, "out of host memory");
    pr->tasks.resize(n);
    std::vector<Request> req;
    std::vector<int64_t> counts;
    // NM_POST_TIMING=1: where the time of this call goes (stderr, one line)
    const bool timing = getenv("NM_POST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now(), t_part[5] = {0, 0, 0, 0, 0};
    auto lap = [&](int k) { const double t = now(); t_part[k] += t - t_mark; t_mark = t; };
    // The tasks are independent between two scoring batches: their host work (noise filter, distances, cliques, merges; sub-motifs and
    // complements) runs on a few threads over contiguous ranges of tasks — 3 ms of a 1 000-task run on one thread —, every range
    // collecting its requests in a list of its own; the lists are joined in task order, so the batches are the serial ones.
    unsigned n_thr = n >= 64 ? std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2)) : 1u;
    if (const char *e = getenv("NM_POST_THREADS")) n_thr = (unsigned)std::max(1, std::min(64, atoi(e)));
    n_thr = std::min<unsigned>(n_thr, std::max<uint32_t>(n, 1u));
    auto range_of = [&](unsigned c, uint32_t *lo, uint32_t *hi) { *lo = (uint32_t)((uint64_t)n * c / n_thr); *hi = (uint32_t)((uint64_t)n * (c + 1) / n_thr); };
    auto on_ranges = [&](const std::function<void(unsigned, uint32_t, uint32_t)> &fn) {
        std::vector<std::thread> pool;
        for (unsigned c = 1; c < n_thr; ++c) pool.emplace_back([&, c] { uint32_t lo, hi; range_of(c, &lo, &hi); fn(c, lo, hi); });
        uint32_t lo, hi;
        range_of(0, &lo, &hi);
        fn(0, lo, hi);
        for (auto &th : pool) th.join();
    };
    // joins the ranges' request lists in task order; `at_of` names the task's list of indices into its range's list (shifted to the joined one)
    auto join = [&](std::vector<std::vector<Request>> &part, std::vector<Request> &into, std::vector<size_t> PostTask::*at_of) {
        into.clear();
        for (unsigned c = 0; c < n_thr; ++c) {
            uint32_t lo, hi;
            range_of(c, &lo, &hi);
            const size_t base = into.size();
            if (base)
                for (uint32_t t = lo; t < hi; ++t)
                    for (size_t &x : pr->tasks[t].*at_of) x += base;
            into.insert(into.end(), std::make_move_iterator(part[c].begin()), std::make_move_iterator(part[c].end()));
        }
    };
    {
        std::vector<std::vector<Request>> part(n_thr);
        on_ranges([&](unsigned c, uint32_t lo, uint32_t hi) {
            std::vector<nmsearch::BestRow> mine;
            for (uint32_t t = lo; t < hi; ++t) {
                if (!task_best(t, mine)) continue;
                pr->tasks[t].none = false;
                prepare(pr->tasks[t], mine, padding, t, part[c]);
            }
        });
        join(part, req, &PostTask::need_at);
    }
    lap(0);
    auto batch = [&]() -> int {
        counts.assign(req.size() * 2, 0);
        if (req.empty()) return NM_OK;
        pr->batches += 1;
        pr->candidates += req.size();
        return score(user, req, counts);
    };
    const size_t n_first = req.size();
    int rc = batch();
    lap(1);
    if (!rc) {
        std::vector<Request> req2;
        {
            std::vector<std::vector<Request>> part(n_thr);
            on_ranges([&](unsigned c, uint32_t lo, uint32_t hi) {
                for (uint32_t t = lo; t < hi; ++t) decide(pr->tasks[t], counts.data(), t, part[c]);
            });
            join(part, req2, &PostTask::accepted_at);
        }
        req.swap(req2);
        lap(2);
        rc = batch();
        lap(3);
    }
    if (rc) {
        delete pr;
        return rc;
    }
    on_ranges([&](unsigned, uint32_t lo, uint32_t hi) {
        for (uint32_t t = lo; t < hi; ++t) finish(pr->tasks[t], counts.data());
    });
    lap(4);
    if (timing)
        fprintf(stderr, "[nm_post] %u tasks: noise filter + merge candidates %.2f ms, scoring batch 1 (%zu candidates) %.2f ms, merge decisions + sub-motif candidates %.2f ms, "
                        "scoring batch 2 (%zu) %.2f ms, sub-motifs + complements %.2f ms\n", n, t_part[0] * 1e3, n_first, t_part[1] * 1e3, t_part[2] * 1e3, req.size(),
                t_part[3] * 1e3, t_part[4] * 1e3);
    *out = pr;
    return NM_OK;
}

struct EngineUser {
    nm_ctx *ctx;
    const uint32_t *task_bin, *task_slot;
    nm_search_reduce_fn reduce;
    void *reduce_user;
};

int engine_score(void *user, const std::vector<Request> &req, std::vector<int64_t> &counts) {
    EngineUser &u = *static_cast<EngineUser *>(user);
    const uint32_t n = (uint32_t)req.size();
    std::vector<uint32_t> bins(n), offs(n);
    std::vector<uint8_t> slots(n), lens(n), modpos(n), masks;
    for (uint32_t i = 0; i < n; ++i) {
        const PMotif &m = req[i].m;
        size_t lo, hi;
        dot_bounds(m.sets, lo, hi);
        if (lo == m.sets.size() || hi - lo > NM_MAX_MOTIF_LEN || m.modpos < (int)lo || m.modpos >= (int)hi)
            return nm_set_error(NM_EINVAL, "post-processing built motif %s (mod position %d) that the engine cannot score", m.str.c_str(), m.modpos);
        bins[i] = u.task_bin[req[i].task];
        slots[i] = (uint8_t)u.task_slot[req[i].task];
        lens[i] = (uint8_t)(hi - lo);
        modpos[i] = (uint8_t)(m.modpos - (int)lo);
        offs[i] = (uint32_t)masks.size();
        masks.insert(masks.end(), m.sets.begin() + lo, m.sets.begin() + hi);
    }
    int rc = nm_score_batch(u.ctx, n, bins.data(), slots.data(), lens.data(), modpos.data(), offs.data(), masks.data(), counts.data());
    if (rc) return rc;
    if (u.reduce) rc = u.reduce(u.reduce_user, counts.data(), (uint64_t)n * 2);
    return rc;
}

struct CustomUser {
    nm_post_score_fn fn;
    void *user;
};

int custom_score(void *user, const std::vector<Request> &req, std::vector<int64_t> &counts) {
    CustomUser &u = *static_cast<CustomUser *>(user);
    const uint32_t n = (uint32_t)req.size();
    std::vector<uint32_t> task(n), off(n + 1, 0);
    std::vector<int32_t> modpos(n);
    std::string text;
    for (uint32_t i = 0; i < n; ++i) {
        task[i] = req[i].task;
        modpos[i] = req[i].m.modpos;
        off[i] = (uint32_t)text.size();
        text += req[i].m.str;
    }
    off[n] = (uint32_t)text.size();
    return u.fn(u.user, n, task.data(), text.data(), off.data(), modpos.data(), counts.data());
}

}  // namespace

extern "C" {

int nm_post_run(nm_ctx *ctx, const nm_search_result *res, const uint32_t *task_bin, const uint32_t *task_merge_slot,
                nm_search_reduce_fn reduce, void *reduce_user, nm_post_result **out) {
    if (!ctx || !res || (nm_search_task_count(res) && (!task_bin || !task_merge_slot))) return nm_set_error(NM_EINVAL, "NULL argument");
    EngineUser u{ctx, task_bin, task_merge_slot, reduce, reduce_user};
    return run_post(nm_search_task_count(res), (int)(nm_search_width(res) / 2),
                    [res](uint32_t t, std::vector<nmsearch::BestRow> &best) { return nm_search_task_best(res, t, best); }, engine_score, &u, out);
}

int nm_post_run_custom(const nm_search_result *res, nm_post_score_fn score_fn, void *user, nm_post_result **out) {
    if (!res || !score_fn) return nm_set_error(NM_EINVAL, "NULL argument");
    CustomUser u{score_fn, user};
    return run_post(nm_search_task_count(res), (int)(nm_search_width(res) / 2),
                    [res](uint32_t t, std::vector<nmsearch::BestRow> &best) { return nm_search_task_best(res, t, best); }, custom_score, &u, out);
}

int nm_post_run_rows_custom(uint32_t n_tasks, uint32_t width, const uint64_t *row_off, const char *motifs, const int64_t *counts,
                            const double *score, nm_post_score_fn score_fn, void *user, nm_post_result **out) {
    if (!score_fn || (n_tasks && (!row_off || !motifs || !counts || !score))) return nm_set_error(NM_EINVAL, "NULL argument");
    if (width == 0 || width % 2 == 0 || width > NM_WIN_MAX_WIDTH) return nm_set_error(NM_ERANGE, "width %u: an odd window width up to %d", width, NM_WIN_MAX_WIDTH);
    CustomUser u{score_fn, user};
    return run_post(n_tasks, (int)(width / 2),
                    [=](uint32_t t, std::vector<nmsearch::BestRow> &best) {
                        best.clear();
                        for (uint64_t i = row_off[t]; i < row_off[t + 1]; ++i)
                            best.push_back(nmsearch::BestRow{std::string(motifs + i * width, width), Model::from_counts(counts[2 * i], counts[2 * i + 1]), score[i]});
                        return true;
                    }, custom_score, &u, out);
}

int nm_post_sizes(const nm_post_result *pr, uint64_t *n_rows, uint64_t *text_bytes, uint64_t stats[2]) {
    if (!pr || !n_rows || !text_bytes) return nm_set_error(NM_EINVAL, "NULL argument");
    uint64_t nr = 0, nb = 0;
    for (const auto &t : pr->tasks)
        for (int s = 0; s < t.n_stages; ++s)
            for (const auto &r : t.stage[s]) {
                nr += 1;
                nb += r.m.str.size() + stripped(r.m).sets.size();
            }
    *n_rows = nr;
    *text_bytes = nb;
    if (stats) { stats[0] = pr->batches; stats[1] = pr->candidates; }
    return NM_OK;
}

int nm_post_export(const nm_post_result *pr, uint32_t *row_task, uint8_t *row_stage, uint64_t *text_off, char *text, int32_t *mod_position,
                   int32_t *mod_position_iupac, int64_t *counts, double *score, int64_t *complement) {
    if (!pr || !row_task || !row_stage || !text_off || !text || !mod_position || !mod_position_iupac || !counts || !score || !complement)
        return nm_set_error(NM_EINVAL, "NULL argument");
    uint64_t i = 0, at = 0;
    for (size_t t = 0; t < pr->tasks.size(); ++t) {
        const PostTask &T = pr->tasks[t];
        uint64_t stage3_first = 0;
        for (int s = 0; s < T.n_stages; ++s) {
            if (s == 3) stage3_first = i;
            for (const auto &r : T.stage[s]) {
                const PMotif st = stripped(r.m);
                const std::string iu = iupac_of(st);
                row_task[i] = (uint32_t)t;
                row_stage[i] = (uint8_t)s;
                text_off[2 * i] = at;
                memcpy(text + at, r.m.str.data(), r.m.str.size());
                at += r.m.str.size();
                text_off[2 * i + 1] = at;
                memcpy(text + at, iu.data(), iu.size());
                at += iu.size();
                mod_position[i] = r.m.modpos;
                mod_position_iupac[i] = st.modpos;
                counts[2 * i] = r.model.n_mod();
                counts[2 * i + 1] = r.model.n_nomod();
                score[i] = r.score;
                complement[i] = r.complement < 0 ? -1 : (int64_t)(stage3_first + (uint64_t)r.complement);
                ++i;
            }
        }
    }
    text_off[2 * i] = at;
    return NM_OK;
}

// The per-stage tables of every task as the text nanomotif_amd.postprocess.format_motifs writes (motif.py:891-897: all non-object columns,
// rows sorted by reference, mod type, motif): table (t, s) = text[off[t * 5 + s], off[t * 5 + s + 1]), the header alone for a stage without
// rows.  task_reference / task_mod_type: the two constant columns of task t.  The text belongs to `post` (until nm_post_free).
int nm_post_tables(nm_post_result *pr, const char *const *task_reference, const char *const *task_mod_type, const char **text, const uint64_t **off,
                   uint64_t *n_off) {
    if (!pr || !text || !off || !n_off || (!pr->tasks.empty() && (!task_reference || !task_mod_type))) return nm_set_error(NM_EINVAL, "NULL argument");
    static const char *HEAD = "reference\tmotif\tmod_type\tmod_position\tscore\tn_mod\tn_nomod\tmotif_iupac\tmod_position_iupac";
    static const char *HEAD_COMP = "\tmotif_complement\tmod_position_complement\tscore_complement\tn_mod_complement\tn_nomod_complement\tmotif_iupac_complement"
                                   "\tmod_position_iupac_complement";
    std::string &o = pr->tables;
    o.clear();
    pr->table_off.assign(pr->tasks.size() * N_STAGES + 1, 0);
    auto put_int = [&](long long v) {
        char b[24];
        const auto r = std::to_chars(b, b + sizeof b, v);
        o.append(b, r.ptr);
    };
    std::vector<uint32_t> order;
    for (size_t t = 0; t < pr->tasks.size(); ++t) {
        const PostTask &T = pr->tasks[t];
        if (!task_reference[t] || !task_mod_type[t]) return nm_set_error(NM_EINVAL, "task %zu: NULL reference / mod type", t);
        for (int s = 0; s < N_STAGES; ++s) {
            pr->table_off[t * N_STAGES + s] = o.size();
            const std::vector<PRow> &rows = s < T.n_stages ? T.stage[s] : T.stage[N_STAGES - 1];
            const size_t n = s < T.n_stages ? rows.size() : 0;
            const bool comp = n != 0 && s == 4;                          // (MotifRow.has_complement_columns: the rows of the last stage)
            o += HEAD;
            if (comp) o += HEAD_COMP;
            o += '\n';
            order.resize(n);
            for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rows[a].m.str < rows[b].m.str; });
            for (size_t k = 0; k < n; ++k) {
                const PRow &r = rows[order[k]];
                const PMotif st = stripped(r.m);
                o += task_reference[t]; o += '\t';
                o += r.m.str; o += '\t';
                o += task_mod_type[t]; o += '\t';
                put_int(r.m.modpos); o += '\t';
                nmsearch::append_py_repr(o, r.score); o += '\t';
                put_int(r.model.n_mod()); o += '\t';
                put_int(r.model.n_nomod()); o += '\t';
                o += iupac_of(st); o += '\t';
                put_int(st.modpos);
                if (comp) {
                    if (r.complement < 0 || (size_t)r.complement >= T.stage[3].size()) o += "\t\t\t\t\t\t\t";
                    else {
                        const PRow &c = T.stage[3][(size_t)r.complement];
                        const PMotif cst = stripped(c.m);
                        o += '\t'; o += c.m.str;
                        o += '\t'; put_int(c.m.modpos);
                        o += '\t'; nmsearch::append_py_repr(o, c.score);
                        o += '\t'; put_int(c.model.n_mod());
                        o += '\t'; put_int(c.model.n_nomod());
                        o += '\t'; o += iupac_of(cst);
                        o += '\t'; put_int(cst.modpos);
                    }
                }
                o += '\n';
            }
        }
    }
    pr->table_off.back() = o.size();
    *text = o.data();
    *off = pr->table_off.data();
    *n_off = pr->table_off.size();
    return NM_OK;
}

int nm_post_free(nm_post_result *pr) {
    delete pr;
    return NM_OK;
}

}  // extern "C"
