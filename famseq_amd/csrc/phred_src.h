// phred_src.h — fabs(-10 log10 p) as the drivers print GPP / FPP (/root/reference/src/file.cpp:696-745:
// +inf -> 99999), ONE definition for both places it runs: compiled into io_kernels.hip (FS_PHRED_DEF
// expands to the code) and carried as text into every generated kernel (FS_PHRED_DEF stringifies it),
// so the fused call path and the separate phred_call stage produce the same bits.
//
// Table-driven (round 3; rounds 1-2 used the atanh series to s^23, 46 instructions): p = m0 2^e with m0 in [1/2, 1);
// the top eight bits of m0's fraction, rounded to seven, pick one of 129 bins k centred on m_k = (128 + k) / 256, whose
// table entry (phred_tab.h, staged into LDS by the kernel: one 16-byte read) holds inv_k ~ 1 / m_k and
// t_k = 10 log10(inv_k) of the ROUNDED inv_k, so that with r = m0 inv_k - 1 (one FMA, |r| <= 2^-8)
//     -10 log10 p = e A + t_k + B log1p(r),   A = -10 log10 2,  B = -10 / ln 10
// holds up to the truncation of log1p's series after r^6 (relative 5e-16).  Below k = 54 (m0 < 0.71) the entry is
// that of 2 m0 — t_k less 10 log10 2, e less one — so that |e A| and |t_k| never nearly cancel; the bins next to 1
// on either side (k = 128: inv = 1, k = 0: inv = 2) have t = 0 exactly and r = the distance to 1 exactly, so a
// probability next to 1 keeps its RELATIVE accuracy.  About 23 instructions; relative error < 1e-15 (tests/fmt and
// tools/phred_accuracy.cpp: 3e-16 observed over 10^8 arguments).  0 -> 99999; NaN or negative -> NaN.
// Everything but a positive finite p is rare: ONE class test (v_cmp_class_f64) and the special values behind a real
// branch (FS_KEEP_BRANCH keeps hipcc from turning it back into compare-and-select instructions per logarithm).
// Needs FS_RCP / FS_FREXP_MANT / FS_FREXP_EXP / FS_HI32 / FS_IS_POS_FINITE / FS_KEEP_BRANCH (device builtins; the host
// test shims them) and the type fs_v2d (two doubles, 16-byte aligned).
FS_PHRED_DEF(
/* the arithmetic alone: right for a positive finite p, finite garbage for anything else (no branch, so that several of
   these in one basic block overlap their table reads and FMA chains) */
static __device__ __forceinline__ double fs_phred_fast(double p, const double *lt) {
  const double m0 = FS_FREXP_MANT(p);
  int e = FS_FREXP_EXP(p);
  const unsigned a = (((((unsigned)FS_HI32(m0)) >> 12) & 0xFFu) + 1u) << 3 & ~15u;
  const fs_v2d tk = *(const fs_v2d *)((const char *)lt + a);
  e -= a < 864u ? 1 : 0;
  const double r = __builtin_fma(m0, tk.x, -1.0);
  double t = __builtin_fma(-1.0 / 6, r, 0.2);
  t = __builtin_fma(t, r, -0.25);
  t = __builtin_fma(t, r, 1.0 / 3);
  t = __builtin_fma(t, r, -0.5);
  const double ln1p = __builtin_fma(r * r, t, r);
  return __builtin_fabs(__builtin_fma(ln1p, -4.3429448190325175, __builtin_fma((double)e, -3.0102999566398120, tk.y)));
}
/* what is printed for the p the arithmetic is not for (q = fs_phred_fast's answer, kept when p is positive and finite) */
static __device__ __forceinline__ double fs_phred_fix(double p, double q) {
  return FS_IS_POS_FINITE(p) ? q : (p == 0.0 ? 99999.0 : (p == __builtin_inf() ? p : __builtin_nan("")));
}
static __device__ __forceinline__ double fs_phred(double p, const double *lt) {
  double q = fs_phred_fast(p, lt);
  if (!FS_IS_POS_FINITE(p)) {
    FS_KEEP_BRANCH();
    q = fs_phred_fix(p, q);
  }
  return q;
}
)
