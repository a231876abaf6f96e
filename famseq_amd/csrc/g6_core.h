// g6_core.h — printf("%g") of a double in [1e-16, 1e6), for the host AND the device.
//
// How every GPP / FPP value reaches the reference's output: C++ `ostream << double` at the default precision
// (/root/reference/src/file.cpp:702-731), i.e. six significant digits, round-half-even of the EXACT binary value,
// trailing zeros removed, scientific notation below 1e-4 and from 1e6.  Two pieces, shared by the command line's
// host formatter (host/fmt_g6.h) and by the device text kernel (io_kernels.hip: text_call_kernel) so that both print
// the same bytes by construction:
//   g6_digits  the six digits and the decimal exponent;
//   g6_emit    the characters, byte by byte (the device writes them into LDS; the host formatter has its own
//              eight-bytes-at-a-time form of this step and is checked against this one: tests/fmt_g6_check.cpp).
// Every Phred value is in range: 0 is printed apart, the smallest non-zero one is -10 log10(1 - 2^-53) = 4.8e-16, the
// largest the 99999 that stands for +inf.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define FS_G6_HD __host__ __device__ inline
#else
#define FS_G6_HD inline
#endif

namespace famseq_g6 {

// 10^k, k <= 19 as integers and k <= 22 as doubles (all exact); in functions so that host and device each get their own
FS_G6_HD uint64_t pow10_u64(int k) {
  const uint64_t t[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                          100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
                          10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                          100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
  return t[k];
}
FS_G6_HD double pow10_f64(int k) {
  const double t[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                        1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  return t[k];
}

// v in [1e-16, 1e6) -> digits in [100000, 999999] and X = floor(log10 of the ROUNDED value), so that the value printed is
// digits * 10^(X - 5).  Returns false when rounding carries to 1e6 (999999.5 and up: scientific notation, the caller's
// general route).  v * 10^k with k = 5 - X <= 22 is the 53-bit significand times 10^k < 2^127 shifted right: one 128-bit
// product, one shift, remainder compared with one half — taken only when the same product in double arithmetic (exact to
// 1.2e-10) comes within 1e-6 of a rounding boundary; otherwise that product's nearest integer is the six digits.
FS_G6_HD bool g6_digits(double v, uint32_t &digits_out, int &X_out) {
  union {
    double d;
    uint64_t u;
  } cv;
  cv.d = v;
  const uint64_t bits = cv.u;
  const int be = int(bits >> 52) & 0x7ff;  // >= 1 here (v >= 1e-16 is normal)
  const uint64_t m = (bits & ((uint64_t(1) << 52) - 1)) | (uint64_t(1) << 52);
  const int e2 = be - 1075;  // v = m * 2^e2, -106 <= e2 <= -33
  // floor(log10 v) is X or X - 1 with X = floor((floor(log2 v) + 1) * log10 2); the tests below decide
  const int b = be - 1023;
  // (78913 / 2^18 = log10 2 to 1e-8: exact floor for |b + 1| <= 60; 6 for v in [2^19, 1e6): capped, v < 1e6 — round 2's
  // formatter indexed its tables with -1 there and was right by the luck of what lay in front of them)
  const int X0 = ((b + 1) * 78913) >> 18 > 5 ? 5 : ((b + 1) * 78913) >> 18;
  int X = X0;
  const int s = -e2;                // 33..106
  uint64_t digits;
  {
    // fast route: v * 10^k in double arithmetic is the exact product (10^k is exact for k <= 22) times (1 + e),
    // |e| <= 2^-53; unless that leaves the rounding (or the decade) in doubt, its nearest integer is the answer
    double x = v * pow10_f64(5 - X);
    if (x < 100000.0 - 1e-6) --X, x = v * pow10_f64(5 - X);
    // nearest integer of a positive x < 2^52 as (x + 2^52) - 2^52 in the default rounding mode; a tie would be rounded
    // to even, but ties are excluded just below
    const double r = (x + 4503599627370496.0) - 4503599627370496.0;
    const double dx = x - r;
    if (x > 100000.0 + 1e-6 && (dx < 0 ? -dx : dx) < 0.5 - 1e-6) {
      digits = uint64_t(r);
      if (digits == 1000000) {
        digits = 100000;
        if (++X == 6) return false;
      }
      digits_out = uint32_t(digits), X_out = X;
      return true;
    }
    X = X0;  // in doubt: the exact route decides, from the start
  }
  for (;;) {
    const int k = 5 - X;  // 0..22
    unsigned __int128 n = (unsigned __int128)m * pow10_u64(k < 19 ? k : 19);
    if (k > 19) n *= pow10_u64(k - 19);
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
      if (++X == 6) return false;
    }
    break;
  }
  digits_out = uint32_t(digits), X_out = X;
  return true;
}

// The characters of digits * 10^(X - 5) as %g lays them out, one byte store each, at out[0..); returns their number
// (at most 11: "1.23457e-05").  -16 <= X <= 5.
template <class Byte>
FS_G6_HD int g6_emit(Byte *out, uint32_t digits, int X) {
  const uint32_t hi = digits / 1000, lo = digits - hi * 1000;  // two groups of three digits
  const uint32_t h0 = hi / 100, h12 = hi - h0 * 100, h1 = h12 / 10, h2 = h12 - h1 * 10;
  const uint32_t l0 = lo / 100, l12 = lo - l0 * 100, l1 = l12 / 10, l2 = l12 - l1 * 10;
  const uint32_t d0 = '0' + h0, d1 = '0' + h1, d2 = '0' + h2, d3 = '0' + l0, d4 = '0' + l1, d5 = '0' + l2;
  // significant digits left after the trailing zeros are removed
  const int nd = l2 ? 6 : (l1 ? 5 : (l0 ? 4 : (h2 ? 3 : (h1 ? 2 : 1))));
  if (X >= 0) {  // X + 1 integer digits (zeros included), then the point and the rest if there is a rest
    const int ip = X + 1;
    out[0] = Byte(d0);
    if (1 < ip || 1 < nd) out[1 + (1 > X)] = Byte(d1);
    if (2 < ip || 2 < nd) out[2 + (2 > X)] = Byte(d2);
    if (3 < ip || 3 < nd) out[3 + (3 > X)] = Byte(d3);
    if (4 < ip || 4 < nd) out[4 + (4 > X)] = Byte(d4);
    if (5 < ip || 5 < nd) out[5 + (5 > X)] = Byte(d5);
    if (nd > ip) {
      out[ip] = Byte('.');
      return nd + 1;
    }
    return ip;
  }
  if (X >= -4) {  // 0.000ddd
    out[0] = Byte('0');
    out[1] = Byte('.');
    if (X <= -2) out[2] = Byte('0');
    if (X <= -3) out[3] = Byte('0');
    if (X <= -4) out[4] = Byte('0');
    Byte *p = out + 1 - X;
    p[0] = Byte(d0);
    if (nd > 1) p[1] = Byte(d1);
    if (nd > 2) p[2] = Byte(d2);
    if (nd > 3) p[3] = Byte(d3);
    if (nd > 4) p[4] = Byte(d4);
    if (nd > 5) p[5] = Byte(d5);
    return 1 - X + nd;
  }
  out[0] = Byte(d0);
  int n = 1;
  if (nd > 1) {
    out[1] = Byte('.');
    out[2] = Byte(d1);
    if (nd > 2) out[3] = Byte(d2);
    if (nd > 3) out[4] = Byte(d3);
    if (nd > 4) out[5] = Byte(d4);
    if (nd > 5) out[6] = Byte(d5);
    n = nd + 1;
  }
  const int ax = -X;  // 5..16
  out[n] = Byte('e');
  out[n + 1] = Byte('-');
  out[n + 2] = Byte('0' + ax / 10);
  out[n + 3] = Byte('0' + ax % 10);
  return n + 4;
}

// A Phred value (0, or 4.8e-16 .. 99999) as the reference prints it; anything else — never produced for a site
// whose status byte says it was computed — is printed as "nan".  At most 11 characters.
template <class Byte>
FS_G6_HD int g6_phred(Byte *out, double v) {
  if (v == 0) {
    out[0] = Byte('0');
    return 1;
  }
  uint32_t digits;
  int X;
  if (!(v >= 1e-16 && v < 1e6) || !g6_digits(v, digits, X)) {
    out[0] = Byte('n'), out[1] = Byte('a'), out[2] = Byte('n');
    return 3;
  }
  return g6_emit(out, digits, X);
}

}  // namespace famseq_g6
