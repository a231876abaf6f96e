// fmt_g6.h — a double as the reference prints it: C++ `ostream << double` at the default precision, i.e. printf's "%g"
// (six significant digits, trailing zeros removed, scientific notation below 1e-4 and from 1e6), which is how every
// GPP / FPP value reaches the output (file.cpp:702-731).  The CLI prints 60 such numbers per ten-member site; libstdc++'s
// std::to_chars(general, 6) costs ≈ 150 ns each, which made formatting the slowest stage of `FamSeq vcf`.
//
// Exact, not approximate: the six digits are round-half-even of the EXACT binary value, as printf gives them.  For
// 1e-16 <= v < 1e6 (every Phred value: the smallest non-zero one is -10 log10(1 - 2^-53) = 4.8e-16, the largest 99999)
// they come from famseq_g6::g6_digits (../g6_core.h: shared with the device text kernel, which prints the same numbers).
// Everything else (zero, larger, smaller, subnormal, negative, non-finite) takes the general route.
// tests/fmt_g6_check.cpp compares it — and the core's byte-by-byte layout the device uses — with snprintf("%g") on 20 M values.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../g6_core.h"

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
  uint32_t digits;
  int X;
  if (!famseq_g6::g6_digits(v, digits, X)) return g6_general(out, v);  // 999999.5 rounds up into scientific notation
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
