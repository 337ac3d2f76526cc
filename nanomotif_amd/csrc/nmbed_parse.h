// Text-to-number helpers of the bedMethyl readers (nmbed.cpp: host parser; nmbedgpu.hip: the host side of the device
// parser uses them for the rows the kernel hands back) — one definition, so both parse a field to the same bits.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

namespace nmbedparse {

static const double POW10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                          1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

inline bool is_null(const char *p, const char *e) {
    const size_t n = (size_t)(e - p);
    return (n == 2 && p[0] == 'N' && p[1] == 'A') || (n == 4 && memcmp(p, "null", 4) == 0) || n == 0;
}

// Decimal text -> double, correctly rounded: mantissa < 2^53 and <= 22 decimals divide exactly once (Clinger's
// fast path); anything else goes through strtod.
inline bool parse_double(const char *p, const char *e, double *out) {
    const char *q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
    uint64_t mant = 0;
    int ndig = 0, dec = 0;
    bool seen_dot = false, ok = true;
    for (; q < e; ++q) {
        if (*q >= '0' && *q <= '9') {
            if (mant > (UINT64_MAX - 9) / 10) { ok = false; break; }
            mant = mant * 10 + (uint64_t)(*q - '0');
            dec += seen_dot;
            ++ndig;
        } else if (*q == '.' && !seen_dot) {
            seen_dot = true;
        } else { ok = false; break; }
    }
    ok = ok && ndig > 0;
    if (ok && mant < (1ull << 53) && dec <= 22) {
        const double v = (double)mant / POW10[dec];
        *out = neg ? -v : v;
        return true;
    }
    std::string tmp(p, e);
    char *endp = nullptr;
    const double v = strtod(tmp.c_str(), &endp);
    if (endp == tmp.c_str() || *endp != '\0') return false;
    *out = v;
    return true;
}

inline bool parse_int(const char *p, const char *e, int64_t *out) {
    if (p == e) return false;
    bool neg = false;
    if (*p == '-') { neg = true; ++p; }
    int64_t v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        v = v * 10 + (*p - '0');
    }
    *out = neg ? -v : v;
    return true;
}

}  // namespace nmbedparse
