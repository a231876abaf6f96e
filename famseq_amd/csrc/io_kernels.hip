// io_kernels.hip — the two memory-bound stages either side of the posterior kernels
// (SURVEY.md section 8(f), rows N2 and N4).
//
//  unpack_pl16   packed integer PLs -> fp64 likelihood rows.  Replaces, on the device, the
//                reference's per-field  lk = pow(10.0, -fabs(atof(field))/10.0)
//                (/root/reference/src/file.cpp:588-590, :821-828) and its "missing sample stays
//                {1,1,1}" rule (:565, :794-809).  The table is filled on the host with the same
//                libm pow call, so the values are identical; PL >= 3240 is exactly 0 either way.
//  phred_call    posterior rows -> what the reference prints per sequenced sample:
//                GPP/FPP = fabs(-10*log10(p)) with +inf -> 99999 (file.cpp:696-745) and
//                FGT = arg-max with strict '<' from -1 (family.cpp:636-665), gathered in VCF
//                column order (get_postProb(true), family.cpp:584-596).
// Both are one-element-per-lane streaming kernels (coalesced 8 B/lane), grid-stride.
#include <hip/hip_runtime.h>

#include "io_kernels.h"

namespace famseq {

namespace {

__global__ __launch_bounds__(256) void unpack_pl16_kernel(const uint16_t *__restrict__ pl, const int32_t *__restrict__ col_of_member,
                                                          const double *__restrict__ lut, int n_members, int n_seq,
                                                          long n_sites, double *__restrict__ lk) {
  const long total = n_sites * n_members * 3;
  const int w3 = 3 * n_members;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long s = e / w3;
    const int r = (int)(e - s * w3), i = r / 3, g = r - 3 * i;
    const int c = col_of_member[i];
    double v = 1.0;  // unsequenced member, or sequenced but missing at this site (all three PLs 0xFFFF)
    if (c >= 0) {
      const uint16_t *p = pl + (s * n_seq + c) * 3;
      const uint16_t a = p[0], b = p[1], d = p[2];
      if (!(a == kPlMissing && b == kPlMissing && d == kPlMissing)) {
        const uint16_t x = g == 0 ? a : (g == 1 ? b : d);
        v = x < kPlLutSize ? lut[x] : 0.0;
      }
    }
    lk[e] = v;
  }
}

__device__ __forceinline__ double phred(double p) {
  const double q = -10 * log10(p);
  return q == __builtin_inf() ? 99999.0 : fabs(q);
}

__global__ __launch_bounds__(256) void phred_call_kernel(const double *__restrict__ post, const double *__restrict__ single,
                                                         const uint8_t *__restrict__ status,
                                                         const int32_t *__restrict__ seq_members, int n_members, int n_seq,
                                                         long n_sites, double *__restrict__ gpp, double *__restrict__ fpp,
                                                         int8_t *__restrict__ fgt) {
  const long total = n_sites * n_seq * 3;
  const int w3 = 3 * n_seq;
  const double nan = __builtin_nan("");
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long s = e / w3;
    const int r = (int)(e - s * w3), k = r / 3, g = r - 3 * k;
    const long src = (s * n_members + seq_members[k]) * 3;
    const int st = status[s] & 3;
    gpp[e] = st == 1 ? nan : phred(single[src + g]);
    fpp[e] = st != 0 ? nan : phred(post[src + g]);
    if (g == 0) {
      int8_t pick = -1;
      if (st == 0) {
        double best = -1;
        for (int h = 0; h < 3; ++h)
          if (best < post[src + h]) {
            best = post[src + h];
            pick = (int8_t)h;
          }
      }
      fgt[s * n_seq + k] = pick;
    }
  }
}

int grid_for(long total) {
  const long blocks = (total + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

}  // namespace

hipError_t launch_unpack_pl16(const uint16_t *d_pl, const int32_t *d_col_of_member, const double *d_lut, int n_members,
                              int n_seq, int64_t n_sites, double *d_lk, hipStream_t stream) {
  if (n_sites <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_pl16_kernel, dim3(grid_for(n_sites * n_members * 3)), dim3(256), 0, stream, d_pl,
                     d_col_of_member, d_lut, n_members, n_seq, (long)n_sites, d_lk);
  return hipGetLastError();
}

hipError_t launch_phred_call(const double *d_post, const double *d_single, const uint8_t *d_status,
                             const int32_t *d_seq_members, int n_members, int n_seq, int64_t n_sites, double *d_gpp,
                             double *d_fpp, int8_t *d_fgt, hipStream_t stream) {
  if (n_sites <= 0 || n_seq <= 0) return hipSuccess;
  hipLaunchKernelGGL(phred_call_kernel, dim3(grid_for(n_sites * n_seq * 3)), dim3(256), 0, stream, d_post, d_single,
                     d_status, d_seq_members, n_members, n_seq, (long)n_sites, d_gpp, d_fpp, d_fgt);
  return hipGetLastError();
}

}  // namespace famseq
