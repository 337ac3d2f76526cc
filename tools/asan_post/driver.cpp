// AddressSanitizer + UBSan run of the native post-processing (nmpost.cpp: noise filter, clique merge, sub-motif removal,
// complement join) on random motif families, CPU only:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread tools/asan_post/driver.cpp \
//       nanomotif_amd/csrc/nmpost.cpp nanomotif_amd/csrc/nmsearch.cpp -o /tmp/asan_post && /tmp/asan_post
// 400 tasks of 1..14 rows built from a few cores (variants, flank extensions, gaps, reverse complements, junk); the scorer
// answers with hashes of (task, motif text, position), so that merges are accepted and rejected at random.  The run is
// repeated and the exports compared.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nmscan.h"

extern "C" const char *nm_last_error(void) { return ""; }
int nm_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}
extern "C" {
// the engine entry points nmsearch.cpp / nmpost.cpp link against (never called here: everything runs on callbacks)
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
int nm_score_batch(nm_ctx *, uint32_t, const uint32_t *, const uint8_t *, const uint8_t *, const uint8_t *, const uint32_t *, const uint8_t *, int64_t *) { return NM_ESTATE; }
}

static uint64_t rng_state = 88172645463325252ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

static int score_fn(void *, uint32_t n, const uint32_t *task, const char *text, const uint32_t *off, const int32_t *pos, int64_t *out) {
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t h = 1469598103934665603ULL ^ task[i];
        for (uint32_t k = off[i]; k < off[i + 1]; ++k) h = (h ^ (uint8_t)text[k]) * 1099511628211ULL;
        h = mix(h ^ (uint64_t)pos[i]);
        const int kind = (int)(h % 4);                        // methylated, unmethylated, sparse, mixed
        out[2 * i] = kind == 0 ? 300 + (int64_t)(h >> 8) % 500 : kind == 1 ? (int64_t)(h >> 8) % 10 : kind == 2 ? (int64_t)(h >> 8) % 40 : 100 + (int64_t)(h >> 8) % 100;
        out[2 * i + 1] = kind == 0 ? (int64_t)(h >> 20) % 30 : kind == 1 ? 300 + (int64_t)(h >> 20) % 500 : (int64_t)(h >> 20) % 40 + (kind == 3 ? 60 : 0);
    }
    return 0;
}

static const uint32_t W = 41, PAD = 20;
static std::string wide(const std::string &core, int pos) {
    std::string s(W, '.');
    const int left = (int)PAD - pos;
    for (size_t k = 0; k < core.size(); ++k)
        if (left + (int)k >= 0 && left + (int)k < (int)W) s[left + k] = core[k];
    return s;
}

int main() {
    const char *cores[] = {"GATC", "CCAGG", "CCTGG", "ACCCA", "CCAAAT", "GA.GAAGC", "GG.GAAGC", "GA.GAAGT", "GCAC......GTT", "AAC......GTGC", "CAG", "A", "TTAA"};
    const int cpos[] = {1, 1, 1, 4, 4, 5, 5, 5, 2, 1, 1, 0, 3};
    const char comp[256] = {};
    (void)comp;
    uint64_t first_hash = 0;
    for (int rep = 0; rep < 2; ++rep) {
        rng_state = 88172645463325252ULL;
        std::vector<uint64_t> row_off{0};
        std::string motifs;
        std::vector<int64_t> counts;
        std::vector<double> score;
        const uint32_t n_tasks = 400;
        for (uint32_t t = 0; t < n_tasks; ++t) {
            const int n_rows = 1 + (int)(rnd() % 14);
            std::vector<std::string> seen;
            for (int r = 0; r < n_rows; ++r) {
                const int k = (int)(rnd() % 13);
                std::string c = cores[k];
                int p = cpos[k];
                const char base = c[p];
                switch (rnd() % 6) {
                    case 0: { const size_t q = rnd() % c.size(); if ((int)q != p && c[q] != '.') c[q] = "ACGT"[rnd() % 4]; break; }
                    case 1: if (rnd() & 1) { c = std::string(1, "ACGT"[rnd() % 4]) + c; p += 1; } else c += "ACGT"[rnd() % 4]; break;
                    case 2: { const int g = 2 + (int)(rnd() % 5); c += std::string(g, '.') + "ACGT"[rnd() % 4]; break; }
                    case 3: if (c.size() > 3 && p > 0) { c = c.substr(1); p -= 1; } break;
                    default: break;
                }
                if (c[p] != base) continue;
                const std::string w = wide(c, p);
                bool dup = false;
                for (const auto &s : seen) dup |= s == w;
                if (dup) continue;
                seen.push_back(w);
                motifs += w;
                const uint64_t h = mix(rnd());
                counts.push_back(50 + (int64_t)(h % 600));
                counts.push_back((int64_t)((h >> 16) % 80));
                score.push_back(0.5 + (double)((h >> 32) % 4000) / 1000.0);
            }
            row_off.push_back(motifs.size() / W);
        }
        nm_post_result *post = nullptr;
        const int rc = nm_post_run_rows_custom(n_tasks, W, row_off.data(), motifs.data(), counts.data(), score.data(), score_fn, nullptr, &post);
        if (rc) { fprintf(stderr, "nm_post_run_rows_custom failed: %d\n", rc); return 1; }
        uint64_t n_rows = 0, text_bytes = 0, stats[2] = {0, 0};
        if (nm_post_sizes(post, &n_rows, &text_bytes, stats)) return 1;
        std::vector<uint32_t> row_task(n_rows);
        std::vector<uint8_t> row_stage(n_rows);
        std::vector<uint64_t> text_off(2 * n_rows + 1);      // motif and IUPAC form per record
        std::vector<char> text(text_bytes + 1);
        std::vector<int32_t> mp(n_rows), mpi(n_rows);
        std::vector<int64_t> cnt(n_rows * 2), compl_(n_rows);
        std::vector<double> sc(n_rows);
        if (nm_post_export(post, row_task.data(), row_stage.data(), text_off.data(), text.data(), mp.data(), mpi.data(), cnt.data(), sc.data(), compl_.data())) return 1;
        uint64_t h = 1469598103934665603ULL;
        for (uint64_t i = 0; i < n_rows; ++i) {
            h = (h ^ row_task[i] ^ ((uint64_t)row_stage[i] << 20) ^ ((uint64_t)mp[i] << 24) ^ ((uint64_t)cnt[2 * i] << 32) ^ (uint64_t)cnt[2 * i + 1] ^ (uint64_t)compl_[i]) * 1099511628211ULL;
            for (uint64_t k = text_off[2 * i]; k < text_off[2 * i + 2]; ++k) h = (h ^ (uint8_t)text[k]) * 1099511628211ULL;
        }
        printf("run %d: %llu tasks, %llu input rows -> %llu exported rows, %llu scoring batches, %llu candidates scored, hash %016llx\n", rep, (unsigned long long)n_tasks,
               (unsigned long long)(motifs.size() / W), (unsigned long long)n_rows, (unsigned long long)stats[0], (unsigned long long)stats[1], (unsigned long long)h);
        nm_post_free(post);
        if (rep == 0) first_hash = h;
        else if (h != first_hash) { fprintf(stderr, "the two runs differ\n"); return 1; }
    }
    printf("identical exports\n");
    return 0;
}
