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
// Both are one-element-per-lane streaming kernels (coalesced 8 B/lane).  A workgroup walks tiles of
// kTileSites sites; inside a tile the element index is a 16-bit number, so site / member / genotype
// come from two multiply-shift divisions (exact for n < 2^16, divisors <= 510: the launchers check
// members <= 170) instead of the 64-bit integer divisions a flat index over n_sites * W3 costs.
// phred_call is bound by its two fp64 logarithms per element (VALU; phred_src.h), unpack_pl16 by HBM.
// Batches served by the generated kernels do both inside the posterior kernel (elim_codegen.cpp kCallHelpers);
// these two remain for the compiled-in team kernel and the lanes-per-site mode.
#include <hip/hip_runtime.h>

#include "g6_core.h"
#include "io_kernels.h"

namespace famseq {

namespace {

constexpr int kTileSites = 128;  // 128 * 3 * kMaxIoMembers = 65280 elements per tile (must stay < 2^16)
constexpr int kMaxIoMembers = 170;  // r / 3 as (r * 171) >> 9 is exact for r < 512

// n / d for n < 2^16, 1 <= d <= 510: the high word of n * (2^32 / d + 1) (error < n / 2^32 < 1 / d)
__device__ __forceinline__ unsigned div_small(unsigned n, unsigned magic) { return __umulhi(n, magic); }
__host__ __device__ constexpr unsigned magic_for(unsigned d) { return 0xFFFFFFFFu / d + 1; }

__global__ __launch_bounds__(256) void unpack_pl16_kernel(const uint16_t *__restrict__ pl, const int32_t *__restrict__ col_of_member,
                                                          const double *__restrict__ lut, int n_members, int n_seq,
                                                          long n_sites, double *__restrict__ lk) {
  const unsigned w3 = 3 * n_members, mw3 = magic_for(w3);
  const long tiles = (n_sites + kTileSites - 1) / kTileSites;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long site0 = t * kTileSites;
    const unsigned ns = n_sites - site0 < kTileSites ? (unsigned)(n_sites - site0) : kTileSites;
    const unsigned nel = ns * w3;
    const uint16_t *plt = pl + site0 * n_seq * 3;
    double *lkt = lk + site0 * w3;
    for (unsigned e = threadIdx.x; e < nel; e += 256) {
      const unsigned s = div_small(e, mw3), r = e - s * w3, i = (r * 171) >> 9, g = r - 3 * i;
      const int c = col_of_member[i];
      double v = 1.0;  // unsequenced member, or sequenced but missing at this site (all three PLs 0xFFFF)
      if (c >= 0) {
        const uint16_t *p = plt + (s * n_seq + c) * 3;
        const uint16_t a = p[0], b = p[1], d = p[2];
        if (!(a == kPlMissing && b == kPlMissing && d == kPlMissing)) {
          const uint16_t x = g == 0 ? a : (g == 1 ? b : d);
          v = x < kPlLutSize ? lut[x] : 0.0;
        }
      }
      lkt[e] = v;
    }
  }
}

#define FS_RCP(x) __builtin_amdgcn_rcp(x)
#define FS_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define FS_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define FS_IS_POS_FINITE(x) __builtin_amdgcn_class(x, 0x180)
#define FS_KEEP_BRANCH() asm volatile("" ::: "memory")
#define FS_HI32(x) __double2hiint(x)
typedef double fs_v2d __attribute__((ext_vector_type(2)));
#define FS_PHRED_DEF(...) __VA_ARGS__
#include "phred_src.h"
#define FS_TAB_ROW(a, b) a, b,
__device__ const double fs_logtab[258] = {
#include "phred_tab.h"
};
#undef FS_TAB_ROW

__global__ __launch_bounds__(256) void phred_call_kernel(const double *__restrict__ post, const double *__restrict__ single,
                                                         const uint8_t *__restrict__ status,
                                                         const int32_t *__restrict__ seq_members, int n_members, int n_seq,
                                                         long n_sites, double *__restrict__ gpp, double *__restrict__ fpp,
                                                         int8_t *__restrict__ fgt) {
  const unsigned w3 = 3 * n_seq, mw3 = magic_for(w3);
  const double nan = __builtin_nan("");
  __shared__ __attribute__((aligned(16))) double s_lt[258];  // fs_phred's table: one 16-byte LDS read per logarithm
  for (int i = threadIdx.x; i < 258; i += 256) s_lt[i] = fs_logtab[i];
  __syncthreads();
  const long tiles = (n_sites + kTileSites - 1) / kTileSites;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long site0 = t * kTileSites;
    const unsigned ns = n_sites - site0 < kTileSites ? (unsigned)(n_sites - site0) : kTileSites;
    const unsigned nel = ns * w3;
    const double *postt = post + site0 * n_members * 3, *singlet = single + site0 * n_members * 3;
    for (unsigned e = threadIdx.x; e < nel; e += 256) {
      const unsigned s = div_small(e, mw3), r = e - s * w3, k = (r * 171) >> 9, g = r - 3 * k;
      const unsigned src = (s * n_members + seq_members[k]) * 3;
      const int st = status[site0 + s] & 3;
      gpp[site0 * w3 + e] = st == 1 ? nan : fs_phred(singlet[src + g], s_lt);
      fpp[site0 * w3 + e] = st != 0 ? nan : fs_phred(postt[src + g], s_lt);
      if (g == 0) {
        int8_t pick = -1;
        if (st == 0) {  // the lowest genotype within 1e-12 relative of the largest posterior (model.cpp famseq_call_genotypes)
          const double p0 = postt[src], p1 = postt[src + 1], p2 = postt[src + 2];
          double best = -1;
          if (best < p0) best = p0;
          if (best < p1) best = p1;
          if (best < p2) best = p2;
          const double thr = best * (1.0 - 1e-12);
          pick = best < 0 ? (int8_t)-1 : (p0 >= thr ? (int8_t)0 : (p1 >= thr ? (int8_t)1 : (int8_t)2));
        }
        fgt[(site0 + s) * n_seq + k] = pick;
      }
    }
  }
}

// Diagnostic: the posterior kernels' traffic shape (read one fp64 array, write two of the same size) as a
// bare elementwise kernel, 16 B per lane, non-temporal stores — what this device gives that shape today.
// Devices of this kind differ by 10-20 % in what their memory system sustains (0.74 of the 8 TB/s nominal
// peak on one box, 0.63 on another, same binary: tools/io_ceiling.hip); bench.py quotes the kernels against
// this figure, measured in the same run, next to the nominal peak.
typedef double v2d_ __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream_probe_kernel(const v2d_ *__restrict__ in, v2d_ *__restrict__ o1,
                                                           v2d_ *__restrict__ o2, size_t n2) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const v2d_ v = in[i];
    __builtin_nontemporal_store(v * 0.5, o1 + i);
    __builtin_nontemporal_store(v * 2.0, o2 + i);
  }
}

// text_call: one lane per (site, sample) pair.  The lane writes its record's characters one byte at a time into its LDS
// record (84-byte stride: 21 words, odd, so the 64 lanes of a wave writing "their" byte k hit 64 different banks); the
// workgroup then copies the tile's records out 16 bytes per lane, contiguous across the wave.  What the host would do
// with the six doubles — six %g conversions, 60 per ten-member site, the slowest stage of `FamSeq vcf` — is ~1,000
// integer instructions per pair here, 0.3 ms per million ten-member sites, and the host is left with one memcpy per pair.
constexpr int kRecWords = 21;  // LDS stride of a record in 4-byte words (kTextStride / 4 = 20 are copied out)
static_assert(kTextStride == 80, "the copy-out below moves five 16-byte pieces per record");

typedef unsigned fs_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void text_call_kernel(const double *__restrict__ gpp, const double *__restrict__ fpp,
                                                        const int8_t *__restrict__ fgt, long n_items, fs_v4u *__restrict__ text) {
  __shared__ uint32_t s_rec[256 * kRecWords];
  const long tiles = (n_items + 255) / 256;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long item0 = t * 256;
    const int ni = n_items - item0 < 256 ? (int)(n_items - item0) : 256;
    const int tid = threadIdx.x;
    if (tid < ni) {
      const long i = item0 + tid;
      const double g0 = gpp[i * 3], g1 = gpp[i * 3 + 1], g2 = gpp[i * 3 + 2];
      const double f0 = fpp[i * 3], f1 = fpp[i * 3 + 1], f2 = fpp[i * 3 + 2];
      const int gt = fgt[i];
      uint32_t *w = s_rec + tid * kRecWords;
#pragma unroll
      for (int k = 0; k < kTextStride / 4; ++k) w[k] = 0;
      unsigned char *rec = reinterpret_cast<unsigned char *>(w);
      int pos = 0;
      pos += famseq_g6::g6_phred(rec + pos, g0), rec[pos++] = ',';
      pos += famseq_g6::g6_phred(rec + pos, g1), rec[pos++] = ',';
      pos += famseq_g6::g6_phred(rec + pos, g2), rec[pos++] = ':';
      pos += famseq_g6::g6_phred(rec + pos, f0), rec[pos++] = ',';
      pos += famseq_g6::g6_phred(rec + pos, f1), rec[pos++] = ',';
      pos += famseq_g6::g6_phred(rec + pos, f2), rec[pos++] = ':';
      // get_postRlt's 0 / 1 / 2 as the drivers print it (file.cpp:733-745): "0/0", "0/1", anything else "1/1"
      rec[pos] = gt == 0 || gt == 1 ? '0' : '1';
      rec[pos + 1] = '/';
      rec[pos + 2] = gt == 0 ? '0' : '1';
      rec[pos + 3] = '\t';
      rec[kTextStride - 1] = (unsigned char)(pos + 4);  // <= 6 * 11 + 6 + 4 = 76
    }
    __syncthreads();
    fs_v4u *out = text + item0 * (kTextStride / 16);
    for (int q = tid; q < ni * (kTextStride / 16); q += 256) {
      const int r = q / 5, p = q - r * 5;
      const uint32_t *w = s_rec + r * kRecWords + p * 4;
      fs_v4u v;
      v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
      __builtin_nontemporal_store(v, out + q);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void g6_probe_kernel(const double *__restrict__ in, long n, fs_v4u *__restrict__ out) {
  __shared__ uint32_t s_rec[256 * 5];  // 20-byte stride: odd in words
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    uint32_t *w = s_rec + threadIdx.x * 5;
    w[0] = w[1] = w[2] = w[3] = 0;
    unsigned char *c = reinterpret_cast<unsigned char *>(w);
    c[15] = (unsigned char)famseq_g6::g6_phred(c, in[i]);
    fs_v4u v;
    v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
    out[i] = v;
  }
}

int grid_for(long n_sites) {
  const long tiles = (n_sites + kTileSites - 1) / kTileSites;
  return (int)(tiles < 1 ? 1 : (tiles > 16384 ? 16384 : tiles));
}

}  // namespace

hipError_t launch_unpack_pl16(const uint16_t *d_pl, const int32_t *d_col_of_member, const double *d_lut, int n_members,
                              int n_seq, int64_t n_sites, double *d_lk, hipStream_t stream) {
  if (n_sites <= 0) return hipSuccess;
  if (n_members < 1 || n_members > kMaxIoMembers) return hipErrorInvalidValue;  // div_small's range
  hipLaunchKernelGGL(unpack_pl16_kernel, dim3(grid_for(n_sites)), dim3(256), 0, stream, d_pl,
                     d_col_of_member, d_lut, n_members, n_seq, (long)n_sites, d_lk);
  return hipGetLastError();
}

hipError_t launch_stream_probe(const double *d_in, double *d_out1, double *d_out2, int64_t n_doubles, hipStream_t stream) {
  if (n_doubles < 2) return hipSuccess;
  hipLaunchKernelGGL(stream_probe_kernel, dim3(8192), dim3(256), 0, stream, reinterpret_cast<const v2d_ *>(d_in),
                     reinterpret_cast<v2d_ *>(d_out1), reinterpret_cast<v2d_ *>(d_out2), (size_t)(n_doubles / 2));
  return hipGetLastError();
}

hipError_t launch_phred_call(const double *d_post, const double *d_single, const uint8_t *d_status,
                             const int32_t *d_seq_members, int n_members, int n_seq, int64_t n_sites, double *d_gpp,
                             double *d_fpp, int8_t *d_fgt, hipStream_t stream) {
  if (n_sites <= 0 || n_seq <= 0) return hipSuccess;
  if (n_seq > kMaxIoMembers || n_members > kMaxIoMembers) return hipErrorInvalidValue;  // div_small's range
  hipLaunchKernelGGL(phred_call_kernel, dim3(grid_for(n_sites)), dim3(256), 0, stream, d_post, d_single,
                     d_status, d_seq_members, n_members, n_seq, (long)n_sites, d_gpp, d_fpp, d_fgt);
  return hipGetLastError();
}

hipError_t launch_text_call(const double *d_gpp, const double *d_fpp, const int8_t *d_fgt, int64_t n_items, char *d_text,
                            hipStream_t stream) {
  if (n_items <= 0) return hipSuccess;
  if (reinterpret_cast<uintptr_t>(d_text) & 15) return hipErrorInvalidValue;
  const long tiles = (n_items + 255) / 256;
  hipLaunchKernelGGL(text_call_kernel, dim3((unsigned)(tiles > 16384 ? 16384 : tiles)), dim3(256), 0, stream, d_gpp, d_fpp, d_fgt,
                     (long)n_items, reinterpret_cast<fs_v4u *>(d_text));
  return hipGetLastError();
}

hipError_t launch_g6_probe(const double *d_in, int64_t n, char *d_out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(g6_probe_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, d_in, (long)n,
                     reinterpret_cast<fs_v4u *>(d_out));
  return hipGetLastError();
}

}  // namespace famseq
