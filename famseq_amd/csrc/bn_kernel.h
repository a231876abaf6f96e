// bn_kernel.h — launch interface between the C ABI (capi.cpp) and the gfx950 kernels.
#ifndef FAMSEQ_BN_KERNEL_H_
#define FAMSEQ_BN_KERNEL_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "plan.h"

namespace famseq {

// Scalars the kernel needs; passed by value (kernarg segment -> SGPRs).
struct KParams {
  int N, L, A, J;
  int team_lanes, tpb, nA, nB, n_slots, row_stride;
  int jn0, jn1, jn2, jd0, jd1, jd2, jlevels;
  int cols, parts;
  int off_joff, off_jdig, off_minfo;  // word offsets into the plan image
  int lds_tc, lds_laneoff, lds_lk, lds_flags, lds_red, lds_part, lds_bins;  // byte offsets
  double lc;
};

KParams make_kparams(const Plan &p, double lc);

// 4 flag combos x 4 member kinds x 27 doubles: the factor-table block `Tc` (see bn_kernel.hip)
void build_factor_tables(const Model &m, double *tc /* 432 */);

// Resident workgroups per CU for this plan (occupancy query), or <0 on error.
int bn_enum_blocks_per_cu(const Plan &p, hipError_t *err);

// Enqueue the enumeration kernel for n_sites sites on `stream`.
hipError_t launch_bn_enum(const Plan &p, const KParams &kp, int grid_blocks, const uint32_t *d_img,
                          const double *d_tc, int64_t n_sites, const double *d_lk, const uint8_t *d_flags,
                          double *d_post, double *d_single, uint8_t *d_status, hipStream_t stream);

}  // namespace famseq
#endif
