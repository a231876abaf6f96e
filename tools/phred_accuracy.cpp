// tools/phred_accuracy.cpp — fs_phred (csrc/phred_src.h + phred_tab.h) on the host against long-double log10l:
//   g++ -O2 -std=c++17 -ffp-contract=off -Ifamseq_amd/csrc -o /tmp/phred_accuracy tools/phred_accuracy.cpp && /tmp/phred_accuracy
// Arguments: uniform random bit patterns over all positive doubles, probabilities next to 1 on both sides, bin edges,
// denormals, the special values.  Prints the largest relative error seen.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
struct fs_v2d { double x, y; };
static inline double fs_mant_(double x) { int e; return std::frexp(x, &e); }
static inline int fs_exp_(double x) { int e; std::frexp(x, &e); return e; }
static inline int fs_hi_(double x) { uint64_t u; std::memcpy(&u, &x, 8); return (int)(u >> 32); }
#define FS_FREXP_MANT(x) fs_mant_(x)
#define FS_FREXP_EXP(x) fs_exp_(x)
#define FS_HI32(x) fs_hi_(x)
#define FS_IS_POS_FINITE(x) ((x) > 0 && (x) < INFINITY)
#define FS_KEEP_BRANCH() (void)0
#define __device__
#define __forceinline__ inline
#define FS_PHRED_DEF(...) __VA_ARGS__
#include "phred_src.h"
#define FS_TAB_ROW(a, b) a, b,
alignas(16) static const double tab[258] = {
#include "phred_tab.h"
};
static double worst = 0, worst_x = 0;
static void check(double x) {
  const double got = fs_phred(x, tab);
  const long double want = fabsl(-10.0L * log10l((long double)x));
  const long double err = want == 0 ? fabsl(got) : fabsl((got - want) / want);
  if (err > worst) worst = (double)err, worst_x = x;
}
int main() {
  uint64_t z = 12345;
  auto next = [&] { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
  for (long i = 0; i < 100000000; ++i) {
    uint64_t u = next() & 0x7FEFFFFFFFFFFFFFull;  // every finite positive pattern (denormals included)
    double x;
    std::memcpy(&x, &u, 8);
    if (x > 0) check(x);
  }
  for (long i = 0; i < 20000000; ++i) {  // probabilities: next to 1 from below, next to 1 from above, around every bin edge
    const double u = (next() >> 11) * (1.0 / 9007199254740992.0);
    check(1.0 - std::ldexp(u, -(int)(next() % 52)));
    check(1.0 + std::ldexp(u, -(int)(next() % 52)));
    check(std::ldexp(0.5 + (next() % 512) / 512.0 * 0.5 + (u - 0.5) * 1e-9, (int)(next() % 40) - 20));
  }
  std::printf("largest relative error %.3e at x = %.17g; phred(1) = %g, phred(0) = %g, phred(-1) = %g, phred(nan) = %g, phred(inf) = %g\n", worst,
              worst_x, fs_phred(1.0, tab), fs_phred(0.0, tab), fs_phred(-1.0, tab), fs_phred(NAN, tab), fs_phred(INFINITY, tab));
  return worst < 1e-15 ? 0 : 1;
}
