// nmsearch_internal.h — what the native search (nmsearch.cpp) and the native post-processing (nmpost.cpp) share: the
// Beta-Bernoulli model in the reference's float64 operation order, numpy's mean, the isolated-base count, and the
// view of a finished search that post-processing starts from.
#pragma once
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nmscan.h"

int nm_set_error(int code, const char *fmt, ...);

namespace nmsearch {

// repr(float) of CPython (float_repr_style "short"): the shortest digits that round-trip, fixed notation for decimal exponents -4 .. 15,
// else d.ddde+XX with at least two exponent digits
inline void append_py_repr(std::string &out, double x) {
    if (std::isnan(x)) { out += "nan"; return; }
    if (std::isinf(x)) { out += x < 0 ? "-inf" : "inf"; return; }
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf - 1, x, std::chars_format::scientific);
    *r.ptr = '\0';                                                      // (atoi below reads the exponent up to here)
    const char *p = buf, *end = r.ptr;
    if (*p == '-') { out += '-'; ++p; }
    const char *epos = p;
    while (epos < end && *epos != 'e') ++epos;
    char digits[32];
    size_t nd = 0;
    for (const char *q = p; q < epos; ++q)
        if (*q != '.') digits[nd++] = *q;
    const int e = atoi(epos + 1);
    if (-4 <= e && e < 16) {
        if (e >= 0) {
            const size_t ip = (size_t)e + 1;
            out.append(digits, std::min(nd, ip));
            if (nd < ip) out.append(ip - nd, '0');
            out += '.';
            if (nd > ip) out.append(digits + ip, nd - ip);
            else out += '0';
        } else {
            out += "0.";
            out.append((size_t)(-e - 1), '0');
            out.append(digits, nd);
        }
    } else {
        out += digits[0];
        if (nd > 1) { out += '.'; out.append(digits + 1, nd - 1); }
        char eb[16];
        snprintf(eb, sizeof eb, "e%c%02d", e < 0 ? '-' : '+', e < 0 ? -e : e);
        out += eb;
    }
}



// scipy.special.psi for positive integers (Cephes psi: exact harmonic sum for x <= 10, asymptotic series beyond)
inline double psi_int(double x) {
    static const double A[7] = {8.33333333333333333333E-2, -2.10927960927960927961E-2, 7.57575757575757575758E-3,
                                -4.16666666666666666667E-3, 3.96825396825396825397E-3, -8.33333333333333333333E-3,
                                8.33333333333333333333E-2};
    if (x <= 10.0) {
        double y = 0.0;
        const int n = (int)x;
        for (int i = 1; i < n; ++i) y += 1.0 / i;
        y -= 0.577215664901532860606512090082402431;
        return y;
    }
    const double z = 1.0 / (x * x);
    double ans = A[0];
    for (int i = 1; i < 7; ++i) ans = ans * z + A[i];
    const double y = z * ans;
    return std::log(x) - (0.5 / x) - y;
}

struct Model {                      // BetaBernoulliModel with the default prior 5 / 5 (model.py:11-33)
    int64_t a = 5, b = 5;
    static Model from_counts(int64_t n_mod, int64_t n_nomod) { return Model{5 + n_mod, 5 + n_nomod}; }
    int64_t n_mod() const { return a - 5; }
    int64_t n_nomod() const { return b - 5; }
    double mean() const { return (double)a / (double)(a + b); }
    double ppo(int64_t np, int64_t nn) const {      // posterior_predictive_per_obs (model.py:78-92)
        const int64_t n = np + nn;
        if (n == 0) return 0.0;
        const double both = psi_int((double)(a + b));
        const double pp = (double)np * (psi_int((double)a) - both) + (double)nn * (psi_int((double)b) - both);
        return pp / (double)n;
    }
};

inline double evaluation_score(const Model &next, const Model &cur) {    // find_motifs_bin.py:1360-1379
    const double pp_next = next.ppo(next.a, next.b);
    const double pp_extra = next.ppo(cur.a - next.a, cur.b - next.b);
    return (next.mean() / cur.mean()) * (pp_next - pp_extra);
}

inline double np_mean(const std::vector<double> &v) {      // numpy's pairwise summation (blocks of 8 accumulators), then / n
    const size_t n = v.size();
    if (n == 0) return std::nan("");
    double res;
    size_t i;
    if (n < 8) {
        res = 0.0;
        for (i = 0; i < n; ++i) res += v[i];
    } else {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = v[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += v[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += v[i];
    }
    return res / (double)n;
}

inline double rel_entr(double x, double y) {               // scipy.special.rel_entr
    if (std::isnan(x) || std::isnan(y)) return std::nan("");
    if (x > 0 && y > 0) return x * std::log(x / y);
    if (x == 0 && y >= 0) return 0.0;
    return INFINITY;
}

inline int count_isolated(const std::string &s, int k) {   // Motif.count_isolated_bases (motif.py:178-194) incl. the n-1 cap
    const int n = (int)s.size();
    int cnt = 0;
    for (int p = 0; p < n; ++p) {
        if (s[p] == '.') continue;
        int len = 0;
        bool all_dot = true, all_n = true;
        for (int q = std::max(p - k, 0); q < p; ++q) { ++len; all_dot &= s[q] == '.'; all_n &= s[q] == 'N'; }
        for (int q = p + 1; q < std::min(p + k + 1, n - 1); ++q) { ++len; all_dot &= s[q] == '.'; all_n &= s[q] == 'N'; }
        if (len && all_dot) ++cnt;
        if (len && all_n) ++cnt;
    }
    return cnt;
}

// One row of graph_to_rows (find_motifs_bin.py:537-549): a graph node that is a best candidate
struct BestRow {
    std::string motif;              // W characters: A C G T .
    Model model;
    double score;
};

}  // namespace nmsearch

// graph nodes of task t that are best candidates, in node (insertion) order; false when the task has no result
bool nm_search_task_best(const nm_search_result *res, uint32_t t, std::vector<nmsearch::BestRow> &out);
uint32_t nm_search_task_count(const nm_search_result *res);
uint32_t nm_search_width(const nm_search_result *res);
