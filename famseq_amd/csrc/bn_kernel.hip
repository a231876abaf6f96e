// bn_kernel.hip — gfx950 (CDNA4, wave64) kernel for the pedigree BN posterior.
//
// Computes, for a batch of sites, what family::calPostProbBN computes per site
// (/root/reference/src/family.cpp:750-1124; CUDA twin family.cu:769-800, 929-971):
//   single posterior (family.cpp:1405-1499), the -LRC shortcut vote (:767-789), the 3^N
//   joint-genotype enumeration (:882-941 autosome, :990-1106 chrX) and the row
//   normalisation with its failure rule (:943-954).
// The reference CUDA kernel strides one site's 3^N configurations over 8x512 threads,
// decodes N base-3 digits per configuration, keeps 3N partial sums per thread in scratch
// and reduces 4096 partials on the host.  None of that is reused here.  This kernel:
//   * owns one site per TEAM of 3^A lanes (a workgroup holds one or more teams),
//   * never decodes a configuration: lane digits are baked into a per-lane offset table
//     (LDS), iterated digits into per-step scalar offsets, low digits into the unrolled
//     register loop (see plan.h),
//   * shares prefix products down the low-member tree (~4 fp64 ops per configuration
//     instead of 3N+1) and accumulates marginals hierarchically,
//   * reduces across lanes through LDS in a fixed order (bit-reproducible run to run).
// All arithmetic is fp64, FMA contraction off (the single posterior and the shortcut vote
// are bit-identical to the CPU reference; the enumeration differs only by summation order).
//
// LDS per workgroup (byte offsets in KParams):
//   tc       [4 flag combos][4 kinds][27]  factor tables: prior at [9g] for founders,
//                                           transmission table [9g+3gm+gf] for children
//   laneoff  [n_slots][3^A] u32             packed (lk index << 16 | table index)
//   lk       [teams][N][3]                  this pass's likelihood rows
//   flags    [teams][4] + member info [N]
//   red      [cols][block_threads]          per-lane results (column-major: conflict-free)
//   part     [teams][3N][parts], bins [teams][3N]
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "bn_kernel.h"

#pragma clang fp contract(off)

namespace famseq {

namespace {

constexpr int kTab = kIterTab;

// ---- the unrolled low-member tree ------------------------------------------------------
// Level K multiplies the running prefix by member K's three factors; the innermost level
// materialises the weight of each of the 3^L configurations.  Returns the subtree total;
// lb[K][g] accumulates member K's marginal.
// Instruction order matters here: left alone, hipcc hoists all 3^L products ahead of the
// additions and spills them to scratch.  PIN() is an empty asm that takes a value "in/out":
// volatile asms keep their program order, so the products of one leaf cannot be issued
// before the additions of the previous leaf have consumed theirs.  It emits no instruction.
#define FAMSEQ_PIN1(a) asm volatile("" : "+v"(a))
#define FAMSEQ_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <int K, int L>
__device__ __forceinline__ double low_tree(double prefix, const double (&v)[L][3], double (&lb)[L][3]) {
  if constexpr (K == L - 1) {
    FAMSEQ_PIN1(prefix);
    const double w0 = prefix * v[K][0];
    const double w1 = prefix * v[K][1];
    const double w2 = prefix * v[K][2];
    lb[K][0] += w0;
    lb[K][1] += w1;
    lb[K][2] += w2;
    double sub = (w0 + w1) + w2;
    FAMSEQ_PIN4(lb[K][0], lb[K][1], lb[K][2], sub);
    return sub;
  } else {
    const double s0 = low_tree<K + 1, L>(prefix * v[K][0], v, lb);
    const double s1 = low_tree<K + 1, L>(prefix * v[K][1], v, lb);
    const double s2 = low_tree<K + 1, L>(prefix * v[K][2], v, lb);
    lb[K][0] += s0;
    lb[K][1] += s1;
    lb[K][2] += s2;
    double sub = (s0 + s1) + s2;
    FAMSEQ_PIN4(lb[K][0], lb[K][1], lb[K][2], sub);
    return sub;
  }
}

__device__ __forceinline__ int ipow3(int e) {
  int r = 1;
  for (int i = 0; i < e; ++i) r *= 3;
  return r;
}

template <int L>
__global__ __launch_bounds__(1024) void bn_enum_kernel(const KParams P, const uint32_t *__restrict__ img,
                                                       const double *__restrict__ tc_g, const long n_sites,
                                                       const double *__restrict__ lk_g,
                                                       const uint8_t *__restrict__ flags_g, double *__restrict__ post_g,
                                                       double *__restrict__ single_g, uint8_t *__restrict__ status_g) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *s_tc = reinterpret_cast<double *>(smem + P.lds_tc);
  uint32_t *s_lo = reinterpret_cast<uint32_t *>(smem + P.lds_laneoff);
  double *s_lk = reinterpret_cast<double *>(smem + P.lds_lk);
  int *s_fl = reinterpret_cast<int *>(smem + P.lds_flags);  // per team: flags, full-BN vote, single fail, BN fail
  double *s_red = reinterpret_cast<double *>(smem + P.lds_red);
  double *s_part = reinterpret_cast<double *>(smem + P.lds_part);
  double *s_bins = reinterpret_cast<double *>(smem + P.lds_bins);

  const int tid = threadIdx.x;
  const int BT = blockDim.x;
  const int N = P.N, W3 = 3 * P.N, TL = P.team_lanes, tpb = P.tpb;
  int *s_minfo = s_fl + 4 * tpb;
  const uint32_t *__restrict__ joff = img + P.off_joff;
  const uint32_t *__restrict__ jdig = img + P.off_jdig;

  // ---- once per workgroup: constants into LDS
  for (int i = tid; i < 4 * 4 * 27; i += BT) s_tc[i] = tc_g[i];
  for (int i = tid; i < P.n_slots * TL; i += BT) s_lo[i] = img[i];
  for (int i = tid; i < N; i += BT) s_minfo[i] = (int)img[P.off_minfo + i];
  const int team = tid / TL;
  const int t = tid - team * TL;
  const double kNaN = __builtin_nan("");
  __syncthreads();

  for (long site0 = (long)blockIdx.x * tpb; site0 < n_sites; site0 += (long)gridDim.x * tpb) {
    const long left = n_sites - site0;
    const int nteam = left < tpb ? (int)left : tpb;
    const int nel = nteam * W3;
    const double *lk_in = lk_g + site0 * W3;

    // ---- phase 0: stage this pass's likelihood rows and flags (coalesced)
    for (int e = tid; e < nel; e += BT) s_lk[e] = lk_in[e];
    for (int e = tid; e < nteam; e += BT) {
      s_fl[4 * e + 0] = flags_g ? (flags_g[site0 + e] & 3) : 0;
      s_fl[4 * e + 1] = 0;
      s_fl[4 * e + 2] = 0;
      s_fl[4 * e + 3] = 0;
    }
    __syncthreads();

    // ---- phase 1: single posterior (family.cpp:1426-1445) and shortcut vote (:767-789)
    for (int e = tid; e < nel; e += BT) {
      const int tm = e / W3, r = e - tm * W3, i = r / 3, g = r - 3 * i;
      const int minfo = s_minfo[i];
      const double *pr = s_tc + s_fl[4 * tm] * 108 + (minfo & 1) * 27;  // founder table of this sex
      const double *row = s_lk + tm * W3 + 3 * i;
      const double p0 = row[0] * pr[0], p1 = row[1] * pr[9], p2 = row[2] * pr[18];
      const double s = (p0 + p1) + p2;
      if (s <= 0) s_fl[4 * tm + 2] = 1;
      s_bins[e] = (g == 0 ? p0 : (g == 1 ? p1 : p2)) / s;
      if (g == 0 && ((minfo >> 2) & 1)) {  // sequenced member: max/sum < lc  =>  full BN
        double big = 0;
        if (big < row[0]) big = row[0];
        if (big < row[1]) big = row[1];
        if (big < row[2]) big = row[2];
        const double sum = (row[0] + row[1]) + row[2];
        big = big / sum;
        if (big < P.lc) s_fl[4 * tm + 1] = 1;
      }
    }
    __syncthreads();

    // ---- phase 2: emit the single posterior; sites that do not enumerate are finished here
    for (int e = tid; e < nel; e += BT) {
      const int tm = e / W3;
      const bool fail = s_fl[4 * tm + 2] != 0, full = s_fl[4 * tm + 1] != 0;
      const double v = fail ? kNaN : s_bins[e];
      if (single_g) single_g[site0 * W3 + e] = v;
      if (fail || !full) post_g[site0 * W3 + e] = v;  // shortcut: same formula (family.cpp:793-878)
    }
    if (status_g)
      for (int e = tid; e < nteam; e += BT) {
        if (s_fl[4 * e + 2]) status_g[site0 + e] = FAMSEQ_ST_SINGLE_FAIL;
        else if (!s_fl[4 * e + 1]) status_g[site0 + e] = FAMSEQ_ST_OK | FAMSEQ_ST_SHORTCUT;
      }

    // ---- phase 3: enumeration.  Lane (team, t) covers the configurations whose fixed
    // digits spell t, for every iterated combination and all 3^L low combinations.
    const bool run = team < nteam && s_fl[4 * team + 1] != 0 && s_fl[4 * team + 2] == 0;
    if (run) {
      const double *lkT = s_lk + team * W3;
      const double *tcf = s_tc + s_fl[4 * team] * 108;
      const uint32_t *lo = s_lo + t;
      const int nAB = P.nA + P.nB;
      double pA = 10000000;  // family.cpp:911
      for (int s = 0; s < P.nA; ++s) {
        const uint32_t pk = lo[s * TL];
        pA = pA * (tcf[pk & 0xffffu] * lkT[pk >> 16]);
      }
      double lb[L][3];
#pragma unroll
      for (int k = 0; k < L; ++k) lb[k][0] = lb[k][1] = lb[k][2] = 0;
      double total = 0;
      for (int c = 3 * L; c < 3 * L + 3 * P.J; ++c) s_red[c * BT + tid] = 0;
      const bool deep = P.jlevels > 1;
      for (int j2 = 0; j2 < P.jn2; ++j2) {
        double sub2 = 0;
        for (int j1 = 0; j1 < P.jn1; ++j1) {
          double sub1 = 0;
          for (int j0 = 0; j0 < P.jn0; ++j0) {
            const uint32_t *jo0 = joff + j0;
            const uint32_t *jo1 = joff + (size_t)P.n_slots * kTab + j1;
            const uint32_t *jo2 = joff + (size_t)2 * P.n_slots * kTab + j2;
            double pj = pA;
            for (int s = P.nA; s < nAB; ++s) {
              uint32_t pk = lo[s * TL] + jo0[s * kTab];
              if (deep) pk += jo1[s * kTab] + jo2[s * kTab];
              pj = pj * (tcf[pk & 0xffffu] * lkT[pk >> 16]);
            }
            double v[L][3];
#pragma unroll
            for (int k = 0; k < L; ++k) {
              const int s = nAB + k;
              uint32_t pk = lo[s * TL] + jo0[s * kTab];
              if (deep) pk += jo1[s * kTab] + jo2[s * kTab];
              const double *tt = tcf + (pk & 0xffffu);
              const double *ll = lkT + (pk >> 16);
              v[k][0] = tt[0] * ll[0];
              v[k][1] = tt[9] * ll[1];
              v[k][2] = tt[18] * ll[2];
            }
            const double tot = low_tree<0, L>(pj, v, lb);
            sub1 += tot;
            const uint32_t dg = jdig[j0];
            for (int d = 0; d < P.jd0; ++d) {
              const int col = 3 * L + 3 * d + ((dg >> (2 * d)) & 3);
              s_red[col * BT + tid] += tot;
            }
          }
          sub2 += sub1;
          if (deep) {
            const uint32_t dg = jdig[kTab + j1];
            for (int d = 0; d < P.jd1; ++d) {
              const int col = 3 * L + 3 * (kIterDigitsPerLevel + d) + ((dg >> (2 * d)) & 3);
              s_red[col * BT + tid] += sub1;
            }
          }
        }
        total += sub2;
        if (deep) {
          const uint32_t dg = jdig[2 * kTab + j2];
          for (int d = 0; d < P.jd2; ++d) {
            const int col = 3 * L + 3 * (2 * kIterDigitsPerLevel + d) + ((dg >> (2 * d)) & 3);
            s_red[col * BT + tid] += sub2;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < L; ++k) {
        s_red[(3 * k + 0) * BT + tid] = lb[k][0];
        s_red[(3 * k + 1) * BT + tid] = lb[k][1];
        s_red[(3 * k + 2) * BT + tid] = lb[k][2];
      }
      s_red[(P.cols - 1) * BT + tid] = total;
    }
    __syncthreads();

    // ---- phase 4: cross-lane reduction, `parts` partial sums per marginal, fixed order
    {
      const int per_team = W3 * P.parts;
      for (int w = tid; w < nteam * per_team; w += BT) {
        const int tm = w / per_team, r = w - tm * per_team, b = r / P.parts, part = r - b * P.parts;
        if (s_fl[4 * tm + 1] == 0 || s_fl[4 * tm + 2] != 0) continue;
        const int i = b / 3, g = b - 3 * i;
        const int minfo = s_minfo[i], bk = (minfo >> 3) & 3, bi = minfo >> 5;
        const int base = tm * TL;
        double sum = 0;
        if (bk == 2) {  // fixed member: lanes whose digit bi equals g, their totals
          const int p3 = ipow3(bi);
          const double *col = s_red + (P.cols - 1) * BT + base;
          for (int u = part; u < TL / 3; u += P.parts) {
            const int hi = u / p3;
            sum += col[hi * 3 * p3 + g * p3 + (u - hi * p3)];
          }
        } else {
          const double *col = s_red + ((bk == 0 ? 3 * bi : 3 * L + 3 * bi) + g) * BT + base;
          for (int u = part; u < TL; u += P.parts) sum += col[u];
        }
        s_part[w] = sum;
      }
    }
    __syncthreads();

    // ---- phase 5: marginals -> row sums -> normalise (family.cpp:943-954)
    for (int e = tid; e < nel; e += BT) {
      const int tm = e / W3;
      if (s_fl[4 * tm + 1] == 0 || s_fl[4 * tm + 2] != 0) continue;
      const int r = e - tm * W3, i = r / 3, g = r - 3 * i;
      const double *pp = s_part + (size_t)(tm * W3 + 3 * i) * P.parts;
      double b[3];
      for (int h = 0; h < 3; ++h) {
        double acc = 0;
        for (int k = 0; k < P.parts; ++k) acc += pp[h * P.parts + k];
        b[h] = acc;
      }
      const double s = (b[0] + b[1]) + b[2];
      if (s <= 0) s_fl[4 * tm + 3] = 1;
      s_bins[e] = b[g] / s;
    }
    __syncthreads();
    for (int e = tid; e < nel; e += BT) {
      const int tm = e / W3;
      if (s_fl[4 * tm + 1] == 0 || s_fl[4 * tm + 2] != 0) continue;
      post_g[site0 * W3 + e] = s_fl[4 * tm + 3] ? kNaN : s_bins[e];
    }
    if (status_g)
      for (int e = tid; e < nteam; e += BT)
        if (s_fl[4 * e + 1] != 0 && s_fl[4 * e + 2] == 0)
          status_g[site0 + e] = s_fl[4 * e + 3] ? FAMSEQ_ST_BN_FAIL : FAMSEQ_ST_OK;
    __syncthreads();
  }
}

using KernelFn = void (*)(const KParams, const uint32_t *, const double *, const long, const double *,
                          const uint8_t *, double *, double *, uint8_t *);

KernelFn kernel_for(int L) {
  switch (L) {
    case 1: return bn_enum_kernel<1>;
    case 2: return bn_enum_kernel<2>;
    case 3: return bn_enum_kernel<3>;
    case 4: return bn_enum_kernel<4>;
    case 5: return bn_enum_kernel<5>;
  }
  return nullptr;
}

}  // namespace

KParams make_kparams(const Plan &p, double lc) {
  KParams k{};
  k.N = p.N; k.L = p.L; k.A = p.A; k.J = p.J;
  k.team_lanes = p.team_lanes; k.tpb = p.teams_per_block;
  k.nA = p.nA; k.nB = p.nB; k.n_slots = p.n_slots;
  k.jn0 = p.jn[0]; k.jn1 = p.jn[1]; k.jn2 = p.jn[2];
  k.jd0 = p.jd[0]; k.jd1 = p.jd[1]; k.jd2 = p.jd[2];
  k.jlevels = p.jlevels;
  k.cols = p.cols; k.parts = p.parts;
  k.off_joff = p.n_slots * p.team_lanes;
  k.off_jdig = k.off_joff + kIterLevels * p.n_slots * kIterTab;
  k.off_minfo = k.off_jdig + kIterLevels * kIterTab;
  const LdsLayout l = lds_layout(p);
  size_t o = 0;
  k.lds_tc = (int)o; o += l.tc;
  k.lds_laneoff = (int)o; o += l.laneoff;
  k.lds_lk = (int)o; o += l.lk;
  k.lds_flags = (int)o; o += l.flags;
  k.lds_red = (int)o; o += l.red;
  k.lds_part = (int)o; o += l.part;
  k.lds_bins = (int)o; o += l.bins;
  k.lc = lc;
  return k;
}

void build_factor_tables(const famseq_model &m, double *tc) {
  for (int i = 0; i < 4 * 4 * 27; ++i) tc[i] = 0;
  for (int fl = 0; fl < 4; ++fl) {
    const bool known = fl & FAMSEQ_FLAG_KNOWN, x = fl & FAMSEQ_FLAG_CHRX;
    const double *autos = known ? m.genoProbK : m.genoProbN;                  // family.cpp:885-893
    const double *male = x ? (known ? m.genoProbXK : m.genoProbXN) : autos;   // family.cpp:992-1009, :1052-1062
    double *blk = tc + fl * 108;
    for (int g = 0; g < 3; ++g) {
      blk[kFounderMale * 27 + 9 * g] = male[g];
      blk[kFounderFemale * 27 + 9 * g] = autos[g];
    }
    for (int c = 0; c < 27; ++c) {
      blk[kChildMale * 27 + c] = x ? m.pcp2Xm[c] : m.pcp2[c];    // family.cpp:1063-1073
      blk[kChildFemale * 27 + c] = x ? m.pcp2Xf[c] : m.pcp2[c];
    }
  }
}

int bn_enum_blocks_per_cu(const Plan &p, hipError_t *err) {
  KernelFn fn = kernel_for(p.L);
  if (!fn) {
    if (err) *err = hipErrorInvalidValue;
    return -1;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)p.lds_bytes);
  int nb = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(fn), p.block_threads,
                                                     p.lds_bytes);
  if (err) *err = e;
  return e == hipSuccess ? nb : -1;
}

hipError_t launch_bn_enum(const Plan &p, const KParams &kp, int grid_blocks, const uint32_t *d_img,
                          const double *d_tc, int64_t n_sites, const double *d_lk, const uint8_t *d_flags,
                          double *d_post, double *d_single, uint8_t *d_status, hipStream_t stream) {
  KernelFn fn = kernel_for(p.L);
  if (!fn) return hipErrorInvalidValue;
  if (n_sites <= 0) return hipSuccess;
  // host-side shape checks: the kernel indexes LDS with these and nothing else
  if (kp.tpb * kp.team_lanes > p.block_threads || p.lds_bytes > 160 * 1024 || grid_blocks < 1)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(fn, dim3(grid_blocks), dim3(p.block_threads), p.lds_bytes, stream, kp, d_img, d_tc,
                     (long)n_sites, d_lk, d_flags, d_post, d_single, d_status);
  return hipGetLastError();
}

}  // namespace famseq
