// enum_codegen.h — source generator of the lane-per-site 3^N enumeration kernel.
#ifndef FAMSEQ_ENUM_CODEGEN_H_
#define FAMSEQ_ENUM_CODEGEN_H_

#include <string>

#include "famseq_hip.h"
#include "model.h"

namespace famseq {

// HIP source of `extern "C" __global__ famseq_enum_lane(lk, flags, post, single, status, n_sites, tc, lc)`.
// Variants of the one-lane-per-site kernel, tried in this order by jit_pick_variant (the first that does not
// spill wins, else the one that spills least): 0, 1 = a 7-member unrolled block (2187 configurations per outer
// step: fewer table rebuilds; measured +1.9 % at 15 members, where it does not spill; at 10 members it spills
// more than the 6-member block and loses the pick), 2, 3 = a 6-member block; odd = the members of the single
// posterior fenced one from the other (fewer registers).  The lanes-per-site forms always use the 6-member block.
constexpr int kEnumVariants = 4;
// group_digits = d > 0: lanes-per-site mode for small batches — 3^d consecutive lanes share a site, each
// taking one combination of the d outermost looped members' digits (d <= enumgen_max_group_digits)
constexpr int kEnumMaxGroupDigits = 4;
// call_ct_out (call path only): the stage-out walk with the row width as a constant (see kElimCallVariants)
std::string enumgen_source(const Model &m, int variant, int group_digits = 0, bool call_mode = false, bool call_ct_out = true);
int enumgen_max_group_digits(const Model &m);
// true when the call-path form of the one-lane-per-site kernel re-reads some members' likelihoods from the
// fp64 rows in global memory inside its loops (wide pedigrees whose LDS row cannot hold them): such a kernel
// must be given fp64 input (lk_g non-null), never packed PLs
bool enumgen_reads_global_rows(const Model &m, int variant);
int enumgen_sites_per_chunk(const Model &m, int group_digits);  // sites a workgroup handles per chunk
int enumgen_block_threads(const Model &m, int group_digits = 0);
// One-line description of the lane kernel's tiling (which members are looped / unrolled).
std::string enumgen_describe(const Model &m, int variant = -1);  // shape of the one-lane-per-site kernel of that variant

}  // namespace famseq
#endif
