// phred_src.h — fabs(-10 log10 p) as the drivers print GPP / FPP (/root/reference/src/file.cpp:696-745:
// +inf -> 99999), ONE definition for both places it runs: compiled into io_kernels.hip (FS_PHRED_DEF
// expands to the code) and carried as text into every generated kernel (FS_PHRED_DEF stringifies it),
// so the fused call path and the separate phred_call stage produce the same bits.
//
// Exponent and mantissa split, m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh((m - 1) / (m + 1)) by its series
// to s^23 (|s| < 0.172), quotient from v_rcp_f64 + one Newton step + one residual correction.  Relative
// error < 1e-15 in 40 instructions (ocml's log10: 111, for a last half ulp that the 6 printed digits
// never see).  0 -> 99999; NaN or negative -> NaN.  Everything but a positive finite p is rare: ONE class test
// (v_cmp_class_f64) and the special values behind a real branch (FS_KEEP_BRANCH keeps hipcc from turning it back
// into nine compare-and-select instructions per logarithm).
// Needs FS_RCP / FS_FREXP_MANT / FS_FREXP_EXP / FS_IS_POS_FINITE / FS_KEEP_BRANCH (device builtins; the host test shims them).
FS_PHRED_DEF(
static __device__ __forceinline__ double fs_phred(double p) {
  const double m0 = FS_FREXP_MANT(p);
  int e = FS_FREXP_EXP(p);
  const bool lo = m0 < 0.70710678118654757;
  const double m = lo ? m0 + m0 : m0;
  e -= lo ? 1 : 0;
  const double f = m - 1.0;
  const double d = m + 1.0;
  double r = FS_RCP(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = __builtin_fma(__builtin_fma(-s, d, f), r, s);
  const double z = s * s;
  double t = 2.0 / 23;
  t = __builtin_fma(t, z, 2.0 / 21);
  t = __builtin_fma(t, z, 2.0 / 19);
  t = __builtin_fma(t, z, 2.0 / 17);
  t = __builtin_fma(t, z, 2.0 / 15);
  t = __builtin_fma(t, z, 2.0 / 13);
  t = __builtin_fma(t, z, 2.0 / 11);
  t = __builtin_fma(t, z, 2.0 / 9);
  t = __builtin_fma(t, z, 2.0 / 7);
  t = __builtin_fma(t, z, 2.0 / 5);
  t = __builtin_fma(t, z, 2.0 / 3);
  const double ln = __builtin_fma(s * z, t, s + s);
  double q = __builtin_fabs(__builtin_fma((double)e, -3.0102999566398120, ln * -4.3429448190325175));
  if (!FS_IS_POS_FINITE(p)) {
    FS_KEEP_BRANCH();
    q = p == 0.0 ? 99999.0 : (p == __builtin_inf() ? p : __builtin_nan(""));
  }
  return q;
}
)
