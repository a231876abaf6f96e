// fmt_g6.h — a double as the reference prints it: C++ `ostream << double` at the default precision, i.e. printf's "%g"
// (six significant digits, trailing zeros removed, scientific notation below 1e-4 and from 1e6), which is how every
// GPP / FPP value reaches the output (file.cpp:702-731).  The CLI prints 60 such numbers per ten-member site; libstdc++'s
// std::to_chars(general, 6) costs ≈ 150 ns each, which made formatting the slowest stage of `FamSeq vcf`.
//
// Exact, not approximate: the six digits are round-half-even of the EXACT binary value, as printf gives them.  For
// 1e-16 <= v < 1e6 (every Phred value: the smallest non-zero one is -10 log10(1 - 2^-53) = 4.8e-16, the largest 99999)
// v * 10^k with k = 5 - floor(log10 v) <= 22 is the 53-bit significand times 10^k < 2^127 shifted right: one 128-bit
// product, one shift, remainder compared with one half — taken only when the same product in double arithmetic
// (exact to 1.2e-10) comes within 1e-6 of a rounding boundary; otherwise that product's nearest integer is the six digits.  Everything else (zero, larger, smaller, subnormal, negative,
// non-finite) takes the general route.  tests/fmt_g6_check.cpp compares it with snprintf("%g") on 20 M values.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace famseq_fmt {

inline char *g6_general(char *out, double v) {
  if (std::isfinite(v)) return std::to_chars(out, out + 32, v, std::chars_format::general, 6).ptr;
  return out + std::snprintf(out, 32, "%g", v);
}

// Appends v at `out` (room for 32 characters), returns the end.
inline char *g6(char *out, double v) {
  if (v == 0 && !std::signbit(v)) {
    *out++ = '0';
    return out;
  }
  if (!(v >= 1e-16 && v < 1e6)) return g6_general(out, v);
  uint64_t bits;
  std::memcpy(&bits, &v, 8);
  const int be = int(bits >> 52) & 0x7ff;                     // >= 1 here (v >= 1e-16 is normal)
  const uint64_t m = (bits & ((uint64_t(1) << 52) - 1)) | (uint64_t(1) << 52);
  const int e2 = be - 1075;                                   // v = m * 2^e2, -106 <= e2 <= -33
  static const uint64_t kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                                      100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
                                      10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                                      100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
  // floor(log10 v) is X or X - 1 with X = floor((floor(log2 v) + 1) * log10 2); the test below decides
  const int b = be - 1023;
  int X = ((b + 1) * 78913) >> 18;                            // 78913 / 2^18 = log10 2 to 1e-8: exact floor for |b + 1| <= 60
  const int s = -e2;                                          // 33..106
  uint64_t digits;
  // Fast route: v * 10^k in double arithmetic is the exact product (10^k is exact for k <= 22) times (1 + e), |e| <= 2^-53,
  // i.e. within 1.2e-10 of it; unless that leaves the rounding (or the decade) in doubt, its nearest integer is the answer.
  static const double kPow10d[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                     1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  {
    double x = v * kPow10d[5 - X];
    if (x < 100000.0 - 1e-6) --X, x = v * kPow10d[5 - X];
    // nearest integer of a positive x < 2^52 as (x + 2^52) - 2^52 in the default rounding mode: two additions instead of a
    // libm call (without -msse4.1 std::nearbyint is one); a tie would be rounded to even, but ties are excluded just below
    const double r = (x + 4503599627370496.0) - 4503599627370496.0;
    if (x > 100000.0 + 1e-6 && std::fabs(x - r) < 0.5 - 1e-6) {
      digits = uint64_t(r);
      if (digits == 1000000) {
        digits = 100000;
        if (++X == 6) return g6_general(out, v);
      }
      goto have_digits;
    }
    X = ((b + 1) * 78913) >> 18;  // in doubt: the exact route decides, from the start
  }
  for (;;) {
    const int k = 5 - X;                                      // 0..22
    unsigned __int128 n = (unsigned __int128)m * kPow10[k < 19 ? k : 19];
    if (k > 19) n *= kPow10[k - 19];
    const unsigned __int128 q = n >> s;
    if (q < 100000) {  // v < 10^X
      --X;
      continue;
    }
    const unsigned __int128 rem = n & (((unsigned __int128)1 << s) - 1), half = (unsigned __int128)1 << (s - 1);
    digits = uint64_t(q);
    if (rem > half || (rem == half && (digits & 1))) ++digits;
    if (digits == 1000000) {
      digits = 100000;
      if (++X == 6) return g6_general(out, v);  // 999999.5 rounds up into scientific notation
    }
    break;
  }
have_digits:
  char d[16] = {0};  // six digits; copied eight bytes at a time below (branch-free: the digit counts are data-dependent)
  {
    const uint32_t v6 = uint32_t(digits), hi = v6 / 1000, lo = v6 - hi * 1000;  // two groups of three digits
    const uint32_t h0 = hi / 100, h12 = hi - h0 * 100, l0 = lo / 100, l12 = lo - l0 * 100;
    d[0] = char('0' + h0);
    d[1] = char('0' + h12 / 10);
    d[2] = char('0' + h12 % 10);
    d[3] = char('0' + l0);
    d[4] = char('0' + l12 / 10);
    d[5] = char('0' + l12 % 10);
  }
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0') --nd;
  if (X >= 0) {  // X < 6 here: digits, a point after the first X + 1, the rest (stripped digits are zeros: right as they stand)
    std::memcpy(out, d, 8);
    out[X + 1] = '.';
    std::memcpy(out + X + 2, d + X + 1, 8);
    return out + (nd > X + 1 ? nd + 1 : X + 1);
  }
  if (X >= -4) {  // 0.000ddd
    std::memcpy(out, "0.000000", 8);
    std::memcpy(out + 1 - X, d, 8);
    return out + 1 - X + nd;
  }
  out[0] = d[0];
  out[1] = '.';
  std::memcpy(out + 2, d + 1, 8);
  out += nd > 1 ? nd + 1 : 1;
  const int ax = -X;  // 5..16
  out[0] = 'e';
  out[1] = '-';
  out[2] = char('0' + ax / 10);
  out[3] = char('0' + ax % 10);
  return out + 4;
}

}  // namespace famseq_fmt
