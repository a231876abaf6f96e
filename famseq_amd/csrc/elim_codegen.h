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
// variant = 4 * r + f.  f = 0..3: decreasing instruction-level parallelism / register pressure (0 no compiler fences,
// 1 a fence per family->member message and the transmission tables through scalar loads, 2 fences after every block,
// 3 also between the members of the single posterior).  r: where the likelihoods live during the message passing —
// 0 re-read from the lane's LDS row at each use (the marginals wait in registers until the row is free), 1 read once
// into registers, the row being the output stage from then on (a tenth of the LDS reads, issued together; round 3:
// equal for trios and quads, +2...17 % from five members on with one exception in ten pedigrees, x1.4-2 beyond twenty
// members: profiles/r03a/exp_elim_registers_first*.txt).  jit_pick_variant takes the first that does not spill, from
// elim_first_variant on.  The call-path form has the r = 0 family only.
constexpr int kElimVariants = 12;  // 8..11 (r = 2): no LDS staging at all, rows straight from and to global memory (the widest pedigrees)
// The call-path form: fence level f = v & 3 of the r = 0 family, and (v & 4) how its stage-out walks the rows — 0: the row width a
// constant (no per-step bound to test, every load of the walk in flight together; more registers), 4: read from the arguments.
constexpr int kElimCallVariants = 8;
// call_mode: the fused call path's form (packed PLs or fp64 rows in; GPP / FPP / FGT / status out)
std::string elim_source(const Model &m, int variant, bool call_mode = false);
int elim_block_threads(const Model &m, bool call_mode = false);
int elim_first_variant(const Model &m, bool call_mode = false);  // where jit_pick_variant starts (see elim_block_threads)

// Shared shell of the generated kernels (see elim_codegen.cpp).
extern const std::string kCallHelpers;  // fused call path: fs_phred, STAGE_IN_PL, STAGE_OUT_CALL, STAGE_FGT
extern const char kCallArgs[];     // ... and the kernel arguments that go with them
extern const char kDiv3Text[];     // FS_DIV_OK / FS_DIV3_FAST: what single_posterior_statements' text needs defined
std::string single_posterior_statements(const Model &m, bool flags_pass, bool store, bool fence_single, const char *dst = "row");
std::string kernel_shell(const Model &m, const std::string &entry, const std::string &comment,
                         const std::string &body, int bt, int min_waves, bool regs_l, bool fence_single,
                         bool chrx_loop = false, int row_doubles = 0, bool call_mode = false, bool lane_body = false,
                         bool call_ct_out = true);

}  // namespace famseq
#endif
