// enum_codegen.h — source generator of the lane-per-site 3^N enumeration kernel.
#ifndef FAMSEQ_ENUM_CODEGEN_H_
#define FAMSEQ_ENUM_CODEGEN_H_

#include <string>

#include "famseq_hip.h"

namespace famseq {

// HIP source of `extern "C" __global__ famseq_enum_lane(lk, flags, post, single, status, n_sites, tc, lc)`.
constexpr int kEnumVariants = 2;  // see elim_codegen.h
std::string enumgen_source(const famseq_model &m, int variant);
int enumgen_block_threads(const famseq_model &m);
// One-line description of the lane kernel's tiling (which members are looped / unrolled).
std::string enumgen_describe(const famseq_model &m);

}  // namespace famseq
#endif
