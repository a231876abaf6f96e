// fmt_g6_check.cpp — famseq_fmt::g6 (csrc/host/fmt_g6.h) against snprintf("%g"), which is what the reference's
// `ostream << double` prints (file.cpp:702-731).  Built and run by tests/test_fmt_g6.py; argv[1] = number of random values.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fmt_g6.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {  // SplitMix64
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double unit() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

static long g_bad = 0, g_n = 0;
static void check(double v) {
  char a[64], b[64];
  *famseq_fmt::g6(a, v) = 0;
  std::snprintf(b, sizeof b, "%g", v);
  ++g_n;
  if (std::strcmp(a, b) != 0 && g_bad++ < 20) std::printf("MISMATCH %.17g: g6 '%s' printf '%s'\n", v, a, b);
  // the byte-by-byte layout the device text kernel uses (csrc/g6_core.h), over its domain: 0 and [1e-16, 1e6) short of the
  // carry into 1e6
  if ((v == 0 && !std::signbit(v)) || (v >= 1e-16 && v < 999999.5)) {
    unsigned char c[64];
    std::memset(c, '#', sizeof c);
    const int n = famseq_g6::g6_phred(c, v);
    if ((n != (int)std::strlen(b) || std::memcmp(c, b, (size_t)n) != 0 || c[n] != '#' || n > 11) && g_bad++ < 20)
      std::printf("MISMATCH %.17g: g6_phred '%.*s' printf '%s'\n", v, n, (const char *)c, b);
  }
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 2000000;
  // specials and the edges of the fast range
  const double specials[] = {0.0, -0.0, 1.0, 10.0, 100000.0, 999999.0, 999999.4, 999999.5, 999999.6, 1e6, 1e-4, 9.99999e-5,
                             9.999995e-5, 0.000123456, 1e-5, 1e-16, 9.9999999e-17, 1.00000001e-16, 4.8216e-16, 99999.0, 99999.5,
                             0.5, 1.5, 2.5, 123456.5, 123457.5, 12345.65, 1e22, 1e300, 5e-324, 2.2250738585072014e-308, -1.5, -99999.0,
                             INFINITY, -INFINITY, NAN, 2.83856e-06, 61.8469, 220.877, 3239.9999, 0.1, 0.2, 0.3, 1.0 / 3, 2.0 / 3};
  for (double v : specials) check(v);
  // exact ties at every digit position: (odd or even 6-digit integer + 1/2) * 2^-j is exactly representable
  for (int i = 0; i < 200000; ++i) {
    const double t = double(100000 + next_u64() % 900000) + 0.5;
    check(t);
    for (int j = 1; j <= 40; j += 3) check(std::ldexp(t, -j));
  }
  // one ulp around powers of ten and around six-digit decimals
  for (int x = -16; x <= 6; ++x) {
    const double p = std::pow(10.0, x);
    check(p);
    check(std::nextafter(p, 0));
    check(std::nextafter(p, 1e9));
    for (int i = 0; i < 2000; ++i) {
      const double d = double(100000 + next_u64() % 900000) * std::pow(10.0, x - 5), h = (double(100000 + next_u64() % 900000) + 0.5) * std::pow(10.0, x - 5);
      check(d);
      check(std::nextafter(d, 0));
      check(std::nextafter(d, 1e9));
      check(h);
      check(std::nextafter(h, 0));
      check(std::nextafter(h, 1e9));
    }
  }
  // log-uniform over the fast range and a little beyond, and Phred values of random probabilities
  for (long i = 0; i < n; ++i) {
    check(std::pow(10.0, -17.0 + 24.0 * unit()));
    const double p = i & 1 ? unit() : 1.0 - std::ldexp(unit(), -int(next_u64() % 53));
    check(std::fabs(-10.0 * std::log10(p)));
    check(std::fabs(-10.0 * std::log10(std::pow(10.0, -330.0 * unit()))));
  }
  std::printf("%ld values, %ld mismatches\n", g_n, g_bad);
  return g_bad != 0;
}
