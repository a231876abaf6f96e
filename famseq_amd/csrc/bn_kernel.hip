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
//   * never decodes a configuration: lane digits are baked into a per-lane row of packed byte
//     offsets (LDS), iterated digits into per-step records fetched with scalar loads, low
//     digits into the unrolled register tree (see plan.h),
//   * materialises every one of the 3^N joint weights with exactly one fp64 FMA (the leaf
//     level of the tree) and gives every other member its marginal from block sums,
//   * reduces across lanes through LDS in a fixed order (bit-reproducible run to run).
// All arithmetic is fp64; FMA contraction is off and FMAs appear only where written (the
// single posterior and the shortcut vote are bit-identical to the CPU reference; the
// enumeration differs from it only by summation order, ~1e-15 relative).
//
// LDS per workgroup (byte offsets in KParams):
//   tc       [4 flag combos][4 kinds][27]  factor tables: prior at [9g] for founders,
//                                           transmission table [9g+3gm+gf] for children
//   laneoff  [3^A][row_stride] u32          lane t's packed offsets, one dword per plan entry
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

constexpr int kMaxList = FAMSEQ_MAX_MEMBERS;  // static bound of the A- and B-list loops

// ---- the unrolled low-member tree ------------------------------------------------------
// For fixed high digits the low members are independent 3-vectors v[k][.].  Level K owns a
// prefix P (product down to level K-1) and hands P*v[K][g] to level K+1.  The leaf level forms
// the weight of each of the 3^L configurations inside one FMA and adds it to the leaf member's
// marginal.  The other members' marginals need the sum of a whole subtree, which for
// independent members is P*v[K][g]*R[K] with R[K] = prod_{k>K} (v[k][0]+v[k][1]+v[k][2]).
// Cost: 3^L FMAs at the leaves + 2*(3^L-3)/2 ops above = ~2 fp64 instructions per configuration.
//
// Instruction order matters: left alone, hipcc hoists the products of the whole unrolled tree
// ahead of their uses and spills them to scratch.  PIN is an empty asm that takes a value
// "in/out"; volatile asms keep program order, so work of one leaf cannot be issued before the
// previous leaf has consumed its operands.  It emits no instruction.
#define FAMSEQ_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define FAMSEQ_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
template <int K, int L>
__device__ __forceinline__ void low_tree(double prefix, const double (&v)[L][3], const double (&R)[L],
                                         double (&lb)[L][3]) {
  if constexpr (K == L - 1) {
    lb[K][0] = __builtin_fma(prefix, v[K][0], lb[K][0]);
    lb[K][1] = __builtin_fma(prefix, v[K][1], lb[K][1]);
    lb[K][2] = __builtin_fma(prefix, v[K][2], lb[K][2]);
    FAMSEQ_PIN4(lb[K][0], lb[K][1], lb[K][2], prefix);  // prefix dies here: no copy is made
  } else {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      double pg = prefix * v[K][g];
      lb[K][g] = __builtin_fma(pg, R[K], lb[K][g]);
      FAMSEQ_PIN2(pg, lb[K][g]);  // both issued before the subtree below, which alone uses pg
      low_tree<K + 1, L>(pg, v, R, lb);
      FAMSEQ_PIN2(prefix, lb[K][g]);  // the next product waits for this subtree
    }
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, so each of
// the barriers of a pass would wait for the wave's global stores to land; no global data is
// handed between lanes here (staging loads are consumed by the ds_write that follows them).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int ipow3(int e) {
  int r = 1;
  for (int i = 0; i < e; ++i) r *= 3;
  return r;
}

__device__ __forceinline__ double lds_f64(const unsigned char *base, uint32_t byte_off) {
  return *reinterpret_cast<const double *>(base + byte_off);
}

// One step record = kStepSlots packed dwords, fetched with scalar loads (uniform address).
struct StepRec {
  uint32_t w[kStepSlots];
};
__device__ __forceinline__ void load_step(StepRec &r, const uint32_t *__restrict__ p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < kStepSlots / 4; ++i) {
    const uint4 x = q[i];
    r.w[4 * i + 0] = x.x;
    r.w[4 * i + 1] = x.y;
    r.w[4 * i + 2] = x.z;
    r.w[4 * i + 3] = x.w;
  }
}

// low factors v[k][g] = T[.. + 9g] * lk[k][g] for the lane's (and step's) parent digits
template <int L>
__device__ __forceinline__ void load_low(double (&v)[L][3], double (&R)[L], double &Sall, const uint32_t *rowL,
                                         const StepRec *rec, const unsigned char *tcf, const unsigned char *lkT) {
#pragma unroll
  for (int k = 0; k < L; ++k) {
    uint32_t pk = rowL[k];
    if (rec) pk += rec->w[k];
    const uint32_t to = pk & 0xffffu, lo = pk >> 16;
    v[k][0] = lds_f64(tcf, to) * lds_f64(lkT, lo);
    v[k][1] = lds_f64(tcf, to + 72) * lds_f64(lkT, lo + 8);
    v[k][2] = lds_f64(tcf, to + 144) * lds_f64(lkT, lo + 16);
  }
  double run = 1.0;
#pragma unroll
  for (int k = L - 1; k >= 0; --k) {
    R[k] = run;
    run = run * ((v[k][0] + v[k][1]) + v[k][2]);
  }
  Sall = run;
}

// __launch_bounds__(768): the largest team (3^6 = 729 lanes) needs a 768-thread workgroup.  For the usual
// 256-thread launch this is not a looser budget than it needs: 768 threads = 3 waves per SIMD = 168 VGPRs,
// exactly the occupancy the kernel runs at (3 workgroups of 256 per CU; LDS allows no more), i.e. the same
// register budget as __launch_bounds__(256, 3) would give.
template <int L, bool LOWINV>
__global__ __launch_bounds__(768) void bn_enum_kernel(const KParams P, const uint32_t *__restrict__ img,
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
  for (int i = tid; i < P.row_stride * TL; i += BT) s_lo[i] = img[i];
  for (int i = tid; i < N; i += BT) s_minfo[i] = (int)img[P.off_minfo + i];
  const int team = tid / TL;
  const int t = tid - team * TL;
  const double kNaN = __builtin_nan("");
  // this lane's row of packed offsets: A-list entries, then the L low members, then the B list
  const uint32_t *rowA = s_lo + (team < tpb ? t : 0) * P.row_stride;
  const uint32_t *rowL = rowA + P.nA;
  const uint32_t *rowB = rowL + L;
  __syncthreads();

  // Contiguous passes per workgroup: neighbouring sites share 128-byte lines, so keeping them
  // on one CU/XCD avoids fetching those lines into two L2s.
  const long passes = (n_sites + tpb - 1) / tpb;
  const long per_wg = (passes + gridDim.x - 1) / gridDim.x;
  const long pass_lo = (long)blockIdx.x * per_wg;
  const long pass_hi = pass_lo + per_wg < passes ? pass_lo + per_wg : passes;

  for (long pass = pass_lo; pass < pass_hi; ++pass) {
    const long site0 = pass * tpb;
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
    lds_barrier();

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
    lds_barrier();

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
      const unsigned char *lkT = reinterpret_cast<const unsigned char *>(s_lk + team * W3);
      const unsigned char *tcf = reinterpret_cast<const unsigned char *>(s_tc + s_fl[4 * team] * 108);
      double pA = 10000000;  // family.cpp:911
#pragma unroll
      for (int s = 0; s < kMaxList; ++s)
        if (s < P.nA) {
          const uint32_t pk = rowA[s];
          pA = pA * (lds_f64(tcf, pk & 0xffffu) * lds_f64(lkT, pk >> 16));
        }
      double lb[L][3], v[L][3], R[L], Sall;
#pragma unroll
      for (int k = 0; k < L; ++k) lb[k][0] = lb[k][1] = lb[k][2] = 0;
      if (LOWINV) load_low<L>(v, R, Sall, rowL, nullptr, tcf, lkT);
      double total = 0;
      for (int c = 3 * L; c < 3 * L + 3 * P.J; ++c) s_red[c * BT + tid] = 0;
      const bool deep = P.jlevels > 1;
      const bool stepped = P.J > 0;
      for (int j2 = 0; j2 < P.jn2; ++j2) {
        double sub2 = 0;
        for (int j1 = 0; j1 < P.jn1; ++j1) {
          double sub1 = 0;
          for (int j0 = 0; j0 < P.jn0; ++j0) {
            StepRec rec;
            double pj = pA;
            if (stepped) {
              load_step(rec, joff + (size_t)j0 * kStepSlots);
              if (deep) {
                StepRec r1, r2;
                load_step(r1, joff + ((size_t)kIterTab + j1) * kStepSlots);
                load_step(r2, joff + ((size_t)2 * kIterTab + j2) * kStepSlots);
#pragma unroll
                for (int i = 0; i < kStepSlots; ++i) rec.w[i] += r1.w[i] + r2.w[i];
              }
#pragma unroll
              for (int s = 0; s < kMaxList - 1; ++s)
                if (s < P.nB) {
                  const uint32_t pk = rowB[s] + rec.w[(L + s) % kStepSlots];
                  pj = pj * (lds_f64(tcf, pk & 0xffffu) * lds_f64(lkT, pk >> 16));
                }
              if (!LOWINV) load_low<L>(v, R, Sall, rowL, &rec, tcf, lkT);
            }
            low_tree<0, L>(pj, v, R, lb);
            const double tot = pj * Sall;
            sub1 += tot;
            if (stepped) {
              const uint32_t dg = jdig[j0];
              for (int d = 0; d < P.jd0; ++d) {
                const int col = 3 * L + 3 * d + ((dg >> (2 * d)) & 3);
                s_red[col * BT + tid] += tot;
              }
            }
          }
          sub2 += sub1;
          if (deep) {
            const uint32_t dg = jdig[kIterTab + j1];
            for (int d = 0; d < P.jd1; ++d) {
              const int col = 3 * L + 3 * (kIterDigitsPerLevel + d) + ((dg >> (2 * d)) & 3);
              s_red[col * BT + tid] += sub1;
            }
          }
        }
        total += sub2;
        if (deep) {
          const uint32_t dg = jdig[2 * kIterTab + j2];
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
    lds_barrier();

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
    lds_barrier();

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
    lds_barrier();
    for (int e = tid; e < nel; e += BT) {
      const int tm = e / W3;
      if (s_fl[4 * tm + 1] == 0 || s_fl[4 * tm + 2] != 0) continue;
      post_g[site0 * W3 + e] = s_fl[4 * tm + 3] ? kNaN : s_bins[e];
    }
    if (status_g)
      for (int e = tid; e < nteam; e += BT)
        if (s_fl[4 * e + 1] != 0 && s_fl[4 * e + 2] == 0)
          status_g[site0 + e] = s_fl[4 * e + 3] ? FAMSEQ_ST_BN_FAIL : FAMSEQ_ST_OK;
    lds_barrier();
  }
}

using KernelFn = void (*)(const KParams, const uint32_t *, const double *, const long, const double *,
                          const uint8_t *, double *, double *, uint8_t *);

KernelFn kernel_for(int L, bool lowinv) {
  switch (L) {
    case 1: return lowinv ? bn_enum_kernel<1, true> : bn_enum_kernel<1, false>;
    case 2: return lowinv ? bn_enum_kernel<2, true> : bn_enum_kernel<2, false>;
    case 3: return lowinv ? bn_enum_kernel<3, true> : bn_enum_kernel<3, false>;
    case 4: return lowinv ? bn_enum_kernel<4, true> : bn_enum_kernel<4, false>;
    case 5: return lowinv ? bn_enum_kernel<5, true> : bn_enum_kernel<5, false>;
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
  k.row_stride = p.row_stride;
  k.off_joff = (p.row_stride * p.team_lanes + 3) & ~3;  // 16-byte aligned: step records are read as uint4
  k.off_jdig = k.off_joff + kIterLevels * kIterTab * kStepSlots;
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

void build_factor_tables(const Model &m, double *tc) {
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
  KernelFn fn = kernel_for(p.L, p.low_invariant != 0);
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
  KernelFn fn = kernel_for(p.L, p.low_invariant != 0);
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
