// svt_fast_format.h -- the three printf conversions of a VCF sample column ('%.0f' for GL, '%0.2f' for SQ, '%.2g' for AB:
// svtyper/parsers.py:391-399, classic.py:466-469) without printf: snprintf parses its format and goes through the
// arbitrary-precision path for every double, ~0.3 us a value, five values a unit -- at a few hundred thousand sites per
// second the text was a tenth of the real-data route.  A double is m * 2^e exactly, so |v| * 10^d fits a 128-bit integer
// for the values a column holds and round-half-even of it is a shift and a compare: the same digits glibc's printf and
// CPython's '%' operator print (both round the exact binary value, ties to even).  Anything outside the fast range
// (huge, NaN, infinite, AB below 1e-4) returns 0 and the caller takes snprintf.  Checked against snprintf value by value in
// tests/test_cpp_helpers.py.
#pragma once

#include <cstdint>
#include <cstring>

namespace svt {

// round-half-even(|v| * 10^d) for finite |v| < 1e13, d <= 5; ok = false outside that range
inline uint64_t scaled_round(double v, int d, bool& ok)
{
    static const uint64_t kPow10[6] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull};
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    bits &= ~(1ull << 63);
    const int exp = (int)(bits >> 52);
    ok = exp < 1023 + 43 && d >= 0 && d <= 5;       // |v| < 2^43 (8.8e12): the product below stays under 2^64 after the shift
    if (!ok) return 0;
    if (exp == 0) return 0;                          // zero / subnormal: far below half a unit of any d
    const unsigned __int128 p = (unsigned __int128)((bits & ((1ull << 52) - 1)) | (1ull << 52)) * kPow10[d];   // < 2^70
    const int e = exp - 1075;                        // v = m * 2^e
    if (e >= 0) return (uint64_t)(p << e);           // (e < 43 - 52 < 0 in fact: kept for completeness)
    const int s = -e;
    if (s >= 72) return 0;                           // p / 2^s < 2^70 / 2^72 < 1/2
    const unsigned __int128 q = p >> s, rem = p & (((unsigned __int128)1 << s) - 1), half = (unsigned __int128)1 << (s - 1);
    return (uint64_t)q + ((rem > half || (rem == half && ((uint64_t)q & 1))) ? 1 : 0);
}

inline char* put_u64(char* p, uint64_t u)
{
    char buf[24];
    char* b = buf + sizeof buf;
    do { *--b = (char)('0' + u % 10); u /= 10; } while (u);
    const size_t n = (size_t)(buf + sizeof buf - b);
    std::memcpy(p, b, n);
    return p + n;
}

inline bool sign_of(double v)
{
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    return (bits >> 63) != 0;
}

// '%.{d}f' into out (at least 32 bytes); the number of characters, 0 = not handled
inline int format_fixed(char* out, double v, int d)
{
    bool ok;
    const uint64_t q = scaled_round(v, d, ok);
    if (!ok) return 0;
    static const uint64_t kPow10[6] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull};
    char* p = out;
    if (sign_of(v)) *p++ = '-';                      // printf keeps the sign of what rounds to zero: "-0", "-0.00"
    p = put_u64(p, q / kPow10[d]);
    if (d) {
        *p++ = '.';
        uint64_t f = q % kPow10[d];
        for (int k = d - 1; k >= 0; --k) { p[k] = (char)('0' + f % 10); f /= 10; }
        p += d;
    }
    return (int)(p - out);
}

// '%.2g' for 1e-4 <= v <= 1 and v == 0 (an allele balance); 0 = not handled
inline int format_g2(char* out, double v)
{
    if (v == 0.0 && !sign_of(v)) { out[0] = '0'; return 1; }
    if (!(v >= 1e-4 && v <= 1.0)) return 0;          // the constants are the smallest doubles >= 10^-k: the decades are exact
    const int d = v >= 1.0 ? 1 : v >= 0.1 ? 2 : v >= 0.01 ? 3 : v >= 0.001 ? 4 : 5;   // two significant digits
    int n = format_fixed(out, v, d);                 // a carry into the next decade ("0.100") loses its zeros below, as %g's does
    if (n <= 0) return 0;
    while (out[n - 1] == '0') --n;
    if (out[n - 1] == '.') --n;
    return n;
}

}  // namespace svt
