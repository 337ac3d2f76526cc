// ThreadSanitizer run of the search state machines on the worker pool (nmsearch.cpp: Workers), CPU only:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tools/tsan_search/driver.cpp nanomotif_amd/csrc/nmsearch.cpp -o /tmp/tsan_search && /tmp/tsan_search
// 300 synthetic searches; counts and window replies are hashes of (task, motif), so that the searches branch, prune and finish
// at different rounds.  The run is repeated with 1 and 8 threads and the exported graphs are compared.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nmscan.h"

int nm_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}
extern "C" {
// the engine entry points nm_search_run links against (never called here: the driver runs on callbacks)
int nm_score_batch_begin(nm_ctx *, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, const uint8_t *, const uint32_t *, const uint8_t *) { return NM_ESTATE; }
int nm_score_batch_end(nm_ctx *, int64_t *) { return NM_ESTATE; }
int nm_win_batch_w_begin(nm_ctx *, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, uint32_t) { return NM_ESTATE; }
int nm_win_batch_w_end(nm_ctx *, int32_t *) { return NM_ESTATE; }
}  // extern "C" — the engine's speculative window batch (nmspec.h) is C++
#include "../../nanomotif_amd/csrc/nmspec.h"
int nmdetail::spec_setup(nm_ctx *, uint32_t, uint32_t, const double *) { return NM_ESTATE; }
int nmdetail::win_batch_spec_begin(nm_ctx *, int, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, uint32_t, const WinSpec *) { return NM_ESTATE; }
int nmdetail::win_batch_spec_end(nm_ctx *, int, uint32_t, int32_t *, int32_t *, int64_t *) { return NM_ESTATE; }
int nmdetail::score_batch_flight_begin(nm_ctx *, int, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, const uint8_t *, const uint32_t *, const uint8_t *) { return NM_ESTATE; }
int nmdetail::score_batch_flight_end(nm_ctx *, int, int64_t *) { return NM_ESTATE; }
extern "C" {
}

static uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static const uint32_t W = 41;
static uint64_t motif_hash(uint32_t task, const char *m) {
    uint64_t h = 1469598103934665603ULL ^ (task % 7);
    for (uint32_t j = 0; j < W; ++j) h = (h ^ (uint8_t)m[j]) * 1099511628211ULL;
    return mix(h);
}
static int score_fn(void *, uint32_t n, const uint32_t *task, const char *motifs, int64_t *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const char *m = motifs + (size_t)i * W;
        uint32_t spec = 0;
        for (uint32_t j = 0; j < W; ++j) spec += m[j] != '.';
        const uint64_t h = motif_hash(task[i], m);
        const int64_t sites = 200000 >> (2 * (spec > 1 ? spec - 1 : 0));
        const bool hot = (h & 7) < 3;                    // some branches are "real motifs"
        out[2 * i] = hot ? sites * 9 / 10 : sites / 50;
        out[2 * i + 1] = sites - out[2 * i];
    }
    return 0;
}
// the window store of the driver: every search owns `left` windows; a removal takes 40 % of them (called from the thread
// that runs the batches only, like the real window engine)
static int window_fn(void *user, uint32_t n, const uint32_t *task, const uint8_t *kind, const char *motifs, int32_t *out) {
    const int STRIDE = 2 + 4 * 64;
    std::vector<int64_t> &left = *static_cast<std::vector<int64_t> *>(user);
    for (uint32_t i = 0; i < n; ++i) {
        const char *m = motifs + (size_t)i * W;
        const uint64_t h = motif_hash(task[i], m);
        int32_t *o = out + (size_t)i * STRIDE;
        uint32_t spec = 0;
        for (uint32_t j = 0; j < W; ++j) spec += m[j] != '.';
        if (kind[i]) {
            o[0] = (int32_t)left[task[i]];
            left[task[i]] = left[task[i]] * 6 / 10;
            o[1] = (int32_t)left[task[i]];
            continue;
        }
        const int32_t active = (int32_t)(left[task[i]] >> (spec > 1 ? spec - 1 : 0));
        o[0] = active;
        for (uint32_t j = 0; j < W; ++j) {
            const uint64_t g = mix(h + j);
            int32_t c[4] = {active / 4, active / 4, active / 4, active - 3 * (active / 4)};
            if ((g & 3) == 0) { const int k = (g >> 2) & 3; const int32_t take = c[(k + 1) & 3] / 2; c[k] += take; c[(k + 1) & 3] -= take; }
            for (int r = 0; r < 4; ++r) o[2 + r * 64 + j] = c[r];
        }
    }
    return 0;
}

struct Export {
    std::vector<uint64_t> off;
    std::vector<char> motifs;
    std::vector<int64_t> counts;
    std::vector<double> score;
    uint64_t stats[3];
    bool operator==(const Export &o) const {
        return off == o.off && motifs == o.motifs && counts == o.counts && score == o.score && !memcmp(stats, o.stats, sizeof stats);
    }
};

static Export run(uint32_t n_tasks, const char *threads) {
    setenv("NM_SEARCH_THREADS", threads, 1);
    nm_search_params p{20, 25, 30, 25, 0.05, 1.5, 0.001, 0.15};
    std::vector<double> bg((size_t)n_tasks * 4 * W, 0.25);
    std::vector<uint64_t> total(n_tasks, 4000);
    std::vector<uint8_t> can(n_tasks);
    for (uint32_t i = 0; i < n_tasks; ++i) can[i] = (i & 1) ? 'C' : 'A';
    nm_search_result *res = nullptr;
    std::vector<int64_t> left(n_tasks, 4000);
    if (nm_search_run_custom(n_tasks, &p, bg.data(), total.data(), can.data(), score_fn, window_fn, &left, &res)) exit(2);
    Export e;
    uint64_t nn, ne, nb;
    nm_search_result_sizes(res, &nn, &ne, &nb, e.stats);
    e.off.resize(3 * (n_tasks + 1));
    std::vector<uint8_t> none(n_tasks);
    e.motifs.resize(nn * W); e.counts.resize(nn * 2); e.score.resize(nn);
    nm_search_result_export(res, e.off.data(), e.off.data() + n_tasks + 1, e.off.data() + 2 * (n_tasks + 1), none.data(), e.motifs.data(),
                            e.counts.data(), e.score.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    nm_search_result_free(res);
    printf("threads %s: %llu scoring rounds, %llu candidates, %llu window requests, %llu nodes\n", threads, (unsigned long long)e.stats[0],
           (unsigned long long)e.stats[1], (unsigned long long)e.stats[2], (unsigned long long)nn);
    return e;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 300;
    if (argc > 2) {                      // timing mode: <n> <threads> [<threads> ...], NM_SEARCH_TIMING=1 prints the split
        for (int k = 2; k < argc; ++k) run(n, argv[k]);
        return 0;
    }
    const Export a = run(n, "1"), b = run(n, "8"), c = run(n, "3");
    if (!(a == b) || !(a == c)) { printf("MISMATCH between thread counts\n"); return 1; }
    printf("identical exports\n");
    return 0;
}
