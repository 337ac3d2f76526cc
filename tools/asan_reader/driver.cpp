// AddressSanitizer + UBSan run of the host readers (nmbed.cpp: bedMethyl text / gzip / BGZF + tabix, FASTA) on good and
// DAMAGED files, CPU only (sanitizers are not available on the GPU pool):
//   python3 tools/asan_reader/run.py        (writes the files, builds this driver with -fsanitize=address,undefined, runs it)
// Every file must either load or be refused with an error message; the sanitizers must stay silent.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nmscan.h"

static std::string g_err;
int nm_set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
extern "C" const char *nm_last_error(void) { return g_err.c_str(); }

static uint64_t checksum(nm_bed *b) {
    uint64_t n = 0, h = 1469598103934665603ULL;
    uint32_t nc = 0;
    if (nm_bed_shape(b, &n, &nc)) return 0;
    const uint32_t *cid; const int64_t *pos, *cov; const int8_t *mod; const uint8_t *strand; const double *frac;
    if (nm_bed_columns(b, &cid, &pos, &mod, &strand, &frac, &cov)) return 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t f;
        memcpy(&f, &frac[i], 8);
        h = (h ^ cid[i] ^ ((uint64_t)pos[i] << 7) ^ ((uint64_t)(uint8_t)mod[i] << 3) ^ strand[i] ^ f ^ (uint64_t)cov[i]) * 1099511628211ULL;
    }
    for (uint32_t i = 0; i < nc; ++i) {
        const char *name = nullptr;
        if (nm_bed_contig_name(b, i, &name) == 0 && name) h = (h ^ strlen(name)) * 1099511628211ULL;
    }
    // the ingest form (32-bit columns), twice (the second call only remaps)
    std::vector<uint32_t> lut(nc);
    for (uint32_t i = 0; i < nc; ++i) lut[i] = i % 3 == 2 ? 0xFFFFFFFFu : i;
    for (int rep = 0; rep < 2; ++rep) {
        const uint32_t *c32, *p32; const int32_t *v32;
        if (nm_bed_ingest_columns(b, lut.data(), nc, &c32, &p32, &mod, &strand, &frac, &v32) == 0)
            for (uint64_t i = 0; i < n; i += 97) h = (h ^ c32[i] ^ p32[i] ^ (uint32_t)v32[i]) * 1099511628211ULL;
    }
    return h ^ n;
}

int main(int argc, char **argv) {
    // lines of the list file: kind \t path [\t index \t name,name,...]
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "r");
    if (!f) return 2;
    char line[8192];
    int bad = 0;
    while (fgets(line, sizeof line, f)) {
        line[strcspn(line, "\n")] = 0;
        std::vector<std::string> col;
        for (char *p = line, *q; p; p = q ? q + 1 : nullptr) {
            q = strchr(p, '\t');
            col.emplace_back(p, q ? (size_t)(q - p) : strlen(p));
        }
        if (col.size() < 2) continue;
        g_err.clear();
        if (col[0] == "fasta") {
            nm_fasta *fa = nullptr;
            const int rc = nm_fasta_open(col[1].c_str(), 3, &fa);
            uint32_t nr = 0; uint64_t bp = 0;
            if (rc == 0) { nm_fasta_shape(fa, &nr, &bp); const uint8_t *s; nm_fasta_sequence(fa, &s); uint64_t h = 0; for (uint64_t i = 0; i < bp; ++i) h += s[i]; printf("fasta %s: %u records, %llu bp, sum %llu\n", col[1].c_str(), nr, (unsigned long long)bp, (unsigned long long)h); nm_fasta_close(fa); }
            else printf("fasta %s: refused (%d) %s\n", col[1].c_str(), rc, g_err.c_str());
            continue;
        }
        nm_bed *b = nullptr;
        int rc;
        if (col.size() >= 4) {
            std::string names;
            std::vector<uint32_t> off{0};
            for (size_t a = 0; a <= col[3].size();) {
                size_t e = col[3].find(',', a);
                if (e == std::string::npos) e = col[3].size();
                names += col[3].substr(a, e - a);
                off.push_back((uint32_t)names.size());
                a = e + 1;
            }
            uint64_t stats[4] = {0, 0, 0, 0};
            rc = nm_bed_open_indexed(col[1].c_str(), col[2].c_str(), (uint32_t)off.size() - 1, names.c_str(), off.data(), 4, &b, stats);
        } else if (col[0] == "counts") rc = nm_bed_open_counts(col[1].c_str(), 4, &b);
        else rc = nm_bed_open(col[1].c_str(), col[0] == "t1" ? 1 : 5, &b);
        if (rc == 0) {
            printf("%s %s: loaded, checksum %016llx\n", col[0].c_str(), col[1].c_str(), (unsigned long long)checksum(b));
            nm_bed_close(b);
        } else {
            printf("%s %s: refused (%d) %s\n", col[0].c_str(), col[1].c_str(), rc, g_err.c_str());
            if (g_err.empty()) bad = 1;
        }
    }
    fclose(f);
    return bad;
}
