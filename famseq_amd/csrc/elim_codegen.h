// elim_codegen.h — source generator of the exact sum-product ("elimination") engine.
#ifndef FAMSEQ_ELIM_CODEGEN_H_
#define FAMSEQ_ELIM_CODEGEN_H_

#include <string>

#include "famseq_hip.h"
#include "model.h"

namespace famseq {

// Loop-free pedigree (member/nuclear-family graph is a forest)?  `why` receives the reason if not.
bool elim_supported(const Model &m, std::string *why);
// members conditioned on to cut the pedigree's loops (0 for a loop-free pedigree), -1 if unsupported
int elim_conditioned_members(const Model &m);
// HIP source of `extern "C" __global__ famseq_elim(lk, flags, post, single, status, n_sites, tc, lc)`
// specialised for the model's topology, sexes and sequenced set.  Throws if unsupported.
// variant 0..kElimVariants-1: decreasing instruction-level parallelism / register pressure
// (jit_pick_variant takes the first that does not spill)
constexpr int kElimVariants = 4;
// call_mode: the fused call path's form (packed PLs or fp64 rows in; GPP / FPP / FGT / status out)
std::string elim_source(const Model &m, int variant, bool call_mode = false);
int elim_block_threads(const Model &m, bool call_mode = false);
int elim_first_variant(const Model &m, bool call_mode = false);  // where jit_pick_variant starts (see elim_block_threads)

// Shared shell of the generated kernels (see elim_codegen.cpp).
extern const std::string kCallHelpers;  // fused call path: fs_phred, STAGE_IN_PL, STAGE_OUT_CALL, STAGE_FGT
extern const char kCallArgs[];     // ... and the kernel arguments that go with them
std::string single_posterior_statements(const Model &m, bool flags_pass, bool store, bool fence_single);
std::string kernel_shell(const Model &m, const std::string &entry, const std::string &comment,
                         const std::string &body, int bt, int min_waves, bool regs_l, bool fence_single,
                         bool chrx_loop = false, int row_doubles = 0, bool call_mode = false, bool lane_body = false);

}  // namespace famseq
#endif
