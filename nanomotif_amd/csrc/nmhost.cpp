// nmhost — host-side helpers of libnmscan that keep the window-extraction step off the Python interpreter:
//   * nm_py_random_sample: bit-exact replica of CPython's random.Random.sample(range(n), k) on a caller-supplied
//     MT19937 state (the reference draws its background windows with random.sample, seq.py:202-225; reproducing
//     the draws natively keeps results identical while removing ~0.5 us of interpreter time per drawn index);
//   * nm_window_letter_counts: per-column A/T/G/C counts of fixed-width windows (EqualLengthDNASet.pssm, seq.py:391-422).
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/nmscan.h"

int nm_set_error(int code, const char *fmt, ...);

namespace {

struct MT {                       // MT19937 exactly as CPython's _randommodule.c (genrand_uint32)
    uint32_t *mt;                 // 624 words
    uint32_t *idx;                // position
    uint32_t next() {
        static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
        constexpr int N = 624, M = 397;
        constexpr uint32_t UPPER = 0x80000000U, LOWER = 0x7fffffffU;
        uint32_t y;
        if (*idx >= N) {
            int kk;
            for (kk = 0; kk < N - M; kk++) {
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ mag01[y & 0x1U];
            }
            for (; kk < N - 1; kk++) {
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ mag01[y & 0x1U];
            }
            y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ mag01[y & 0x1U];
            *idx = 0;
        }
        y = mt[(*idx)++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= (y >> 18);
        return y;
    }
    // Random._randbelow_with_getrandbits for 0 < n < 2**32 (random.py:239-247)
    uint32_t randbelow(uint32_t n) {
        const int k = 32 - __builtin_clz(n);          // n.bit_length()
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
};

}  // namespace

// n_draws outputs of genrand_uint32 from mt_state (advanced in place); out may be NULL.  For the device-side draws of
// nm_plan_windows: the raw sequence every generator stream reads, and the state a stream is left in.
void nm_mt_outputs(uint32_t mt_state[625], uint64_t n_draws, uint32_t *out) {
    MT g{mt_state, mt_state + 624};
    if (out)
        for (uint64_t i = 0; i < n_draws; ++i) out[i] = g.next();
    else
        for (uint64_t i = 0; i < n_draws; ++i) (void)g.next();
}

extern "C" {

int nm_py_random_sample(uint32_t mt_state[625], uint64_t n, uint64_t k, uint32_t *out_indices) {
    if (!mt_state || (k && !out_indices)) return nm_set_error(NM_EINVAL, "NULL argument");
    if (k > n) return nm_set_error(NM_EINVAL, "Sample larger than population or is negative");
    if (n >= 0xFFFFFFFFull) return nm_set_error(NM_ERANGE, "population of %llu is beyond 32 bits", (unsigned long long)n);
    if (mt_state[624] > 624) return nm_set_error(NM_EINVAL, "invalid MT19937 position");
    MT g{mt_state, mt_state + 624};
    // random.py:449-466 (CPython 3.10): pool-based selection for small populations, set-based otherwise
    double setsize = 21;
    if (k > 5) setsize += std::pow(4.0, std::ceil(std::log((double)k * 3.0) / std::log(4.0)));
    if ((double)n <= setsize) {
        std::vector<uint32_t> pool(n);
        for (uint64_t i = 0; i < n; ++i) pool[i] = (uint32_t)i;
        for (uint64_t i = 0; i < k; ++i) {
            const uint32_t j = g.randbelow((uint32_t)(n - i));
            out_indices[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
    } else {
        // membership test only: a bitmap is equivalent to the reference's set()
        std::vector<uint64_t> seen((n + 63) / 64, 0);
        for (uint64_t i = 0; i < k; ++i) {
            uint32_t j = g.randbelow((uint32_t)n);
            while (seen[j >> 6] & (1ull << (j & 63))) j = g.randbelow((uint32_t)n);
            seen[j >> 6] |= 1ull << (j & 63);
            out_indices[i] = j;
        }
    }
    return NM_OK;
}

int nm_py_random_sample_many(uint32_t mt_state[625], uint32_t m, const uint64_t *n, const uint64_t *k, uint32_t *out_indices) {
    if (m && (!n || !k)) return nm_set_error(NM_EINVAL, "NULL argument");
    uint64_t at = 0;
    for (uint32_t i = 0; i < m; ++i) {          // consecutive random.sample(range(n[i]), k[i]) calls on one generator
        const int rc = nm_py_random_sample(mt_state, n[i], k[i], out_indices ? out_indices + at : nullptr);
        if (rc) return rc;
        at += k[i];
    }
    return NM_OK;
}

int nm_py_random_sample_groups(uint32_t n_groups, const uint32_t *init_state /*[n_groups][625]*/, const uint64_t *group_off /*[n_groups + 1]*/,
                                const uint64_t *n, const uint64_t *k, uint32_t *out_indices, uint32_t final_state[625]) {
    if (n_groups && (!init_state || !group_off || !n || !k || !final_state)) return nm_set_error(NM_EINVAL, "NULL argument");
    if (n_groups == 0) return NM_OK;
    const uint64_t total = group_off[n_groups];
    std::vector<uint64_t> out_off(total + 1, 0);
    for (uint64_t i = 0; i < total; ++i) out_off[i + 1] = out_off[i] + k[i];
    // every group is its own generator stream (the reference seeds each task afresh, find_motifs_bin.py:152-171): the
    // groups are drawn concurrently, the draws inside a group stay in order
    std::vector<int> rcs(n_groups, NM_OK);
    std::vector<std::string> errs(n_groups);
    const unsigned threads = std::max(1u, std::min<unsigned>({16u, std::thread::hardware_concurrency(), n_groups}));
    std::vector<std::vector<uint32_t>> last(threads);
    auto work = [&](unsigned t) {
        std::vector<uint32_t> st(625);
        for (uint32_t g = t; g < n_groups; g += threads) {
            memcpy(st.data(), init_state + (size_t)g * 625, 625 * 4);
            for (uint64_t i = group_off[g]; i < group_off[g + 1] && rcs[g] == NM_OK; ++i) {
                rcs[g] = nm_py_random_sample(st.data(), n[i], k[i], out_indices ? out_indices + out_off[i] : nullptr);
                if (rcs[g] != NM_OK) errs[g] = nm_last_error();
            }
            if (g == n_groups - 1) last[t] = st;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (uint32_t g = 0; g < n_groups; ++g)
        if (rcs[g] != NM_OK) return nm_set_error(rcs[g], "%s", errs[g].c_str());
    for (unsigned t = 0; t < threads; ++t)
        if (!last[t].empty()) memcpy(final_state, last[t].data(), 625 * 4);
    return NM_OK;
}

int nm_window_letter_counts(const uint8_t *seq, uint64_t seq_len, const int64_t *starts, uint64_t n_windows, uint32_t width,
                            int64_t *counts /*[4][width], rows A,T,G,C*/) {
    if (!seq || !counts || (n_windows && !starts)) return nm_set_error(NM_EINVAL, "NULL argument");
    for (uint64_t i = 0; i < n_windows; ++i)
        if (starts[i] < 0 || (uint64_t)starts[i] + width > seq_len) return nm_set_error(NM_EINVAL, "window %llu outside the sequence", (unsigned long long)i);
    unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n_windows < 20000) threads = 1;
    auto count_range = [&](int64_t *c, uint64_t first, uint64_t step) {
        for (uint64_t i = first; i < n_windows; i += step) {
            const uint8_t *w = seq + starts[i];
            for (uint32_t j = 0; j < width; ++j) {
                const uint8_t ch = w[j];
                const int row = ch == 'A' ? 0 : ch == 'T' ? 1 : ch == 'G' ? 2 : ch == 'C' ? 3 : -1;
                if (row >= 0) c[(size_t)row * width + j] += 1;
            }
        }
    };
    if (threads == 1) {                                   // the common case (one contig's ~1 % sample): no thread spawn
        memset(counts, 0, sizeof(int64_t) * 4 * width);
        count_range(counts, 0, 1);
        return NM_OK;
    }
    std::vector<std::vector<int64_t>> part(threads, std::vector<int64_t>(4 * (size_t)width, 0));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(count_range, part[t].data(), (uint64_t)t, (uint64_t)threads);
    for (auto &th : pool) th.join();
    memset(counts, 0, sizeof(int64_t) * 4 * width);
    for (auto &p : part)
        for (size_t i = 0; i < 4 * (size_t)width; ++i) counts[i] += p[i];
    return NM_OK;
}

}  // extern "C"
