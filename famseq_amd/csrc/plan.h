// plan.h — host-side enumeration plan for the BN posterior kernel.
//
// The reference enumerates all 3^N joint genotypes with an odometer over members in
// PED order (family.cpp:894-941; member 0 fastest).  The sum it forms,
//     post[i][a] = sum over g with g_i = a of  1e7 * prod_m f_m(g_m | g_mother(m), g_father(m)),
// does not depend on the visiting order, so the plan re-tiles the same 3^N space for a
// 64-wide-wavefront machine.  Members are split into three groups:
//
//   low   (L <= 5)  childless members.  Their parents are never low, so for fixed high
//                   digits each low member contributes a 3-vector; a lane walks the 3^L
//                   low configurations in a fully unrolled register loop with shared
//                   prefix products.
//   fixed (A)       high members whose digit is the lane's own base-3 digit: a "team" of
//                   3^A lanes covers them.  Several teams (sites) share a workgroup when
//                   3^A is small.
//   iter  (J)       remaining high members; every lane loops over their 3^J digit
//                   combinations (up to three nested loops of <= 243 steps).
//
// L + A + J = N and every configuration is visited exactly once.
#ifndef FAMSEQ_PLAN_H_
#define FAMSEQ_PLAN_H_

#include <cstdint>
#include <string>
#include <vector>

#include "famseq_hip.h"
#include "model.h"

namespace famseq {

constexpr int kMaxLow = 5;
constexpr int kMaxFixed = 6;
constexpr int kIterDigitsPerLevel = 5;  // 3^5 = 243 steps per nested loop
constexpr int kIterLevels = 3;
constexpr int kIterTab = 243;
constexpr int kStepSlots = 20;  // dwords per step record (L + nB <= N <= 20)

// member kinds select the 27-entry factor table (see Tc in bn_kernel.hip)
enum Kind { kFounderMale = 0, kFounderFemale = 1, kChildMale = 2, kChildFemale = 3 };

struct PlanOptions {
  int fixed_digits = -1;   // A, -1 = auto
  int low_members = -1;    // L, -1 = auto
  int block_threads = -1;  // -1 = auto
};

struct Plan {
  int N = 0, L = 0, A = 0, J = 0;
  int team_lanes = 1;       // 3^A
  int block_threads = 256;
  int teams_per_block = 1;  // sites per workgroup pass
  int nA = 0, nB = 0;       // high members evaluated once per site / once per iter step
  int n_slots = 0;          // nA + L + nB, in that order (A list, low members, B list)
  int row_stride = 1;       // dwords per lane row of `laneoff` (odd: conflict-free ds_read_b32)
  int low_invariant = 1;    // no low member has an iterated parent: low factors hoisted out of the steps
  int jlevels = 0;          // nested iter loops in use (0..3)
  int jn[kIterLevels] = {1, 1, 1};   // steps per level (3^digits), level 0 innermost
  int jd[kIterLevels] = {0, 0, 0};   // digits per level
  int cols = 0;             // per-lane reduction columns: 3L low bins, 3J iter bins, 1 total
  int parts = 1;            // partial sums per bin in the cross-lane reduction
  std::vector<int> low_member, fixed_member, iter_member;  // digit position -> member
  std::vector<int> slot_member;                            // n_slots
  std::vector<int> kind;                                   // N
  // packed BYTE offsets: low 16 bits = offset into the 108-double factor-table block of the
  // site's flag combo (8*(kind*27 + 9g + 3gm + gf), minus the parts supplied elsewhere), high
  // 16 bits = offset into the site's lk[N][3] (8*(3m + g)).  lane part + step part = address.
  std::vector<uint32_t> laneoff;  // [team_lanes][row_stride]: entry e of lane t
  std::vector<uint32_t> joff;     // [kIterLevels][kIterTab][kStepSlots]: k < L low member k, L + s B-list entry s
  std::vector<uint16_t> jdigits;  // [kIterLevels][kIterTab]: 2 bits per digit of the level
  // cross-lane reduction: bin b = 3*member + g
  std::vector<int> bin_kind;      // N: 0 low, 1 iter, 2 fixed
  std::vector<int> bin_index;     // N: k / q / p
  size_t lds_bytes = 0;

  std::string json() const;
  // flat 32-bit image uploaded to the device (layout documented in bn_kernel.hip)
  std::vector<uint32_t> device_image() const;
};

// Throws std::runtime_error on an invalid model/options.
Plan build_plan(const Model &m, const PlanOptions &opt);

// LDS carve-up shared by the plan (for occupancy estimates) and the kernel.
struct LdsLayout {
  size_t tc, laneoff, lk, flags, red, part, bins, total;
};
LdsLayout lds_layout(const Plan &p);

}  // namespace famseq
#endif
