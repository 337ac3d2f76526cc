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
const char *nm_last_error(void) { return ""; }
// the engine entry points nm_search_run links against (the public begin / end halves are never called by the search)
int nm_score_batch_begin(nm_ctx *, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, const uint8_t *, const uint32_t *, const uint8_t *) { return NM_ESTATE; }
int nm_score_batch_end(nm_ctx *, int64_t *) { return NM_ESTATE; }
int nm_win_batch_w_begin(nm_ctx *, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, uint32_t) { return NM_ESTATE; }
int nm_win_batch_w_end(nm_ctx *, int32_t *) { return NM_ESTATE; }
}  // extern "C" — the engine's speculative window batch (nmspec.h) is C++
#include "../../nanomotif_amd/csrc/nmspec.h"

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

// ---- engine mode: a stand-in for the flights' begin / end halves (nmspec.h) — the replies are computed in the begin half, on whichever
// thread sends the batch, into the flight's own buffers, and handed out by the end half on the collecting thread: what ThreadSanitizer
// then sees is the search's own hand-over between the two (nmsearch.cpp: Sender).  No speculation (column -1 for every request).
static uint64_t bytes_hash(uint64_t seed, const uint8_t *p, size_t n) {
    uint64_t h = 1469598103934665603ULL ^ seed;
    for (size_t j = 0; j < n; ++j) h = (h ^ p[j]) * 1099511628211ULL;
    return mix(h);
}
struct FlightBox {
    std::vector<int32_t> win;
    std::vector<int64_t> counts;
    bool win_open = false, score_open = false;
} g_box[NM_SEARCH_MAX_FLIGHTS];
static std::vector<int64_t> g_left;                      // windows left per window task (a task belongs to one flight)
int nmdetail::spec_setup(nm_ctx *, uint32_t, uint32_t, const double *) { return NM_OK; }
int nmdetail::win_batch_spec_begin(nm_ctx *, int f, uint32_t n, const uint32_t *task, const uint8_t *kind, const uint8_t *sets, uint32_t ws, const WinSpec *) {
    FlightBox &b = g_box[f];
    if (b.win_open) return NM_ESTATE;
    const size_t stride = 2 + 4 * (size_t)ws;
    b.win.assign(n * stride, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *m = sets + (size_t)i * ws;
        const uint64_t h = bytes_hash(task[i] % 7, m, W);
        int32_t *o = b.win.data() + i * stride;
        uint32_t spec = 0;
        for (uint32_t j = 0; j < W; ++j) spec += m[j] != 15;
        if (kind[i]) {
            o[0] = (int32_t)g_left[task[i]];
            g_left[task[i]] = g_left[task[i]] * 6 / 10;
            o[1] = (int32_t)g_left[task[i]];
            continue;
        }
        const int32_t active = (int32_t)(g_left[task[i]] >> (spec > 1 ? spec - 1 : 0));
        o[0] = active;
        for (uint32_t j = 0; j < W; ++j) {
            const uint64_t g = mix(h + j);
            int32_t c[4] = {active / 4, active / 4, active / 4, active - 3 * (active / 4)};
            if ((g & 3) == 0) { const int k = (g >> 2) & 3; const int32_t take = c[(k + 1) & 3] / 2; c[k] += take; c[(k + 1) & 3] -= take; }
            for (int r = 0; r < 4; ++r) o[2 + r * ws + j] = c[r];
        }
    }
    b.win_open = true;
    return NM_OK;
}
int nmdetail::win_batch_spec_end(nm_ctx *, int f, uint32_t n, int32_t *out, int32_t *spec_info, int64_t *spec_counts) {
    FlightBox &b = g_box[f];
    if (!b.win_open) return NM_ESTATE;
    b.win_open = false;
    if (out) memcpy(out, b.win.data(), b.win.size() * 4);
    if (spec_info) for (uint32_t r = 0; r < n; ++r) { spec_info[2 * r] = -1; spec_info[2 * r + 1] = 0; }
    if (spec_counts) memset(spec_counts, 0, (size_t)n * 8 * sizeof(int64_t));
    return NM_OK;
}
int nmdetail::score_batch_flight_begin(nm_ctx *, int f, uint32_t n, const uint32_t *bin, const uint8_t *slot, const uint8_t *len, const uint8_t *modpos,
                                       const uint32_t *off, const uint8_t *masks) {
    FlightBox &b = g_box[f];
    if (b.score_open) return NM_ESTATE;
    b.counts.assign((size_t)n * 2, 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t spec = 0;
        for (uint32_t j = 0; j < len[i]; ++j) spec += masks[off[i] + j] != 15;
        const uint64_t h = bytes_hash((bin[i] * 2 + slot[i]) % 7 + 131ull * modpos[i], masks + off[i], len[i]);
        const int64_t sites = 200000 >> (2 * (spec > 1 ? spec - 1 : 0));
        const bool hot = (h & 7) < 3;
        b.counts[2 * i] = hot ? sites * 9 / 10 : sites / 50;
        b.counts[2 * i + 1] = sites - b.counts[2 * i];
    }
    b.score_open = true;
    return NM_OK;
}
int nmdetail::score_batch_flight_end(nm_ctx *, int f, int64_t *out) {
    FlightBox &b = g_box[f];
    if (!b.score_open) return NM_ESTATE;
    b.score_open = false;
    if (out) memcpy(out, b.counts.data(), b.counts.size() * 8);
    return NM_OK;
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

static Export run(uint32_t n_tasks, const char *threads, int engine_flights = 0, bool sender = true) {
    setenv("NM_SEARCH_THREADS", threads, 1);
    if (engine_flights) {
        char buf[16];
        snprintf(buf, sizeof buf, "%d", engine_flights);
        setenv("NM_SEARCH_FLIGHTS", buf, 1);
        if (sender) unsetenv("NM_SEARCH_NO_SENDER"); else setenv("NM_SEARCH_NO_SENDER", "1", 1);
    }
    nm_search_params p{20, 25, 30, 25, 0.05, 1.5, 0.001, 0.15};
    std::vector<double> bg((size_t)n_tasks * 4 * W, 0.25);
    std::vector<uint64_t> total(n_tasks, 4000);
    std::vector<uint8_t> can(n_tasks);
    for (uint32_t i = 0; i < n_tasks; ++i) can[i] = (i & 1) ? 'C' : 'A';
    nm_search_result *res = nullptr;
    std::vector<int64_t> left(n_tasks, 4000);
    if (engine_flights) {                // the engine back end of nm_search_run on the stand-in above (the ctx is never looked into)
        std::vector<uint32_t> tbin(n_tasks), tslot(n_tasks), twin(n_tasks);
        // (skew: all but a handful of tasks travel with flight 0 — bins even —, so flight 1 ends early and the last flight's requests
        //  are sent AHEAD through its slot: the tail path of run_tasks)
        const bool skew = getenv("NM_TSAN_SKEW") != nullptr;
        for (uint32_t i = 0; i < n_tasks; ++i) { tbin[i] = skew ? (i + 8 < n_tasks ? 2 * i : 2 * i + 1) : i / 2; tslot[i] = i & 1; twin[i] = i; }
        g_left.assign(n_tasks, 4000);
        if (nm_search_run(reinterpret_cast<nm_ctx *>(&g_left), n_tasks, tbin.data(), tslot.data(), twin.data(), &p, bg.data(), total.data(), can.data(),
                          nullptr, nullptr, &res)) exit(2);
    } else if (nm_search_run_custom(n_tasks, &p, bg.data(), total.data(), can.data(), score_fn, window_fn, &left, &res)) exit(2);
    Export e;
    uint64_t nn, ne, nb;
    nm_search_result_sizes(res, &nn, &ne, &nb, e.stats);
    e.off.resize(3 * (n_tasks + 1));
    std::vector<uint8_t> none(n_tasks);
    e.motifs.resize(nn * W); e.counts.resize(nn * 2); e.score.resize(nn);
    nm_search_result_export(res, e.off.data(), e.off.data() + n_tasks + 1, e.off.data() + 2 * (n_tasks + 1), none.data(), e.motifs.data(),
                            e.counts.data(), e.score.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    nm_search_result_free(res);
    if (engine_flights) printf("engine stand-in, %d flight(s)%s, ", engine_flights, sender && engine_flights > 1 ? " + sending thread" : "");
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
    // the flights of the engine back end and the thread that sends their batches: the same graphs however the tasks are grouped
    // (the batch counters differ: more flights, more and smaller batches)
    const Export e1 = run(n, "8", 1), e2 = run(n, "3", 2, false), e3 = run(n, "8", 2, true), e4 = run(n, "3", 3, true);
    setenv("NM_TSAN_SKEW", "1", 1);
    setenv("NM_SEARCH_TIMING", "1", 1);                 // (prints how many window counts were asked for ahead and used)
    const Export e5 = run(n, "8", 2, true);
    setenv("NM_SEARCH_NO_AHEAD", "1", 1);               // (the skew renames the bins, which the stand-in's counts hang on: its own pair of runs)
    const Export e6 = run(n, "8", 2, true);
    unsetenv("NM_SEARCH_NO_AHEAD");
    unsetenv("NM_SEARCH_TIMING");
    unsetenv("NM_TSAN_SKEW");
    auto same_graphs = [](const Export &x, const Export &y) { return x.off == y.off && x.motifs == y.motifs && x.counts == y.counts && x.score == y.score; };
    if (!same_graphs(e1, e2) || !same_graphs(e1, e3) || !same_graphs(e1, e4)) { printf("MISMATCH between flight counts\n"); return 1; }
    if (!same_graphs(e5, e6)) { printf("MISMATCH between requests sent ahead and not\n"); return 1; }
    printf("identical graphs over 1 / 2 / 3 flights, with and without the sending thread, with requests sent ahead in the tail\n");
    return 0;
}
