// capi.cpp — the C ABI of libfamseq_hip.so (see include/famseq_hip.h).
//
// Replaces the host driver the reference wraps around its kernel
// (/root/reference/src/family.cu:1106-1705: per SITE 6 cudaMalloc, 5 H2D copies, one
// launch, a 4096x3N D2H copy, a host reduction and 6 cudaFree).  Here the context owns
// device memory, streams and pinned staging for its lifetime, a call moves a whole batch,
// and nothing is reduced on the host.  There is no CPU compute path in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "bn_kernel.h"
#include "elim_codegen.h"
#include "enum_codegen.h"
#include "jit.h"
#include "famseq_hip.h"
#include "io_kernels.h"
#include "plan.h"

using namespace famseq;

// The fused call path's side of a generated kernel's argument list (kCallArgs): packed PLs in, GPP / FPP /
// FGT out.  All null on the plain path.
struct CallIO {  // = struct fs_call_args of the generated source (elim_codegen.cpp kCallHelpers)
  const uint16_t *pl = nullptr;
  const double *lut = nullptr;
  const int32_t *col = nullptr, *slot = nullptr;
  double *gpp = nullptr, *fpp = nullptr;
  int8_t *fgt = nullptr;
  int32_t n_seq = 0;
  uint32_t magic_w = 0, magic_n = 0;
  unsigned long long *phase_clk = nullptr;  // FAMSEQ_PHASE_CLOCK (measuring aid): cycles per phase of the call-path kernel, summed over waves
};
constexpr int kPhases = 8;

struct famseq_ctx {
  Model model;
  bool big = false;  // more than FAMSEQ_MAX_MEMBERS members: no enumeration plan, sum-product engine only
  PlanOptions opt{};
  Plan plan{};
  KParams kp{};
  bool plan_dirty = true;
  int device = -1;
  int n_cus = 0;
  int blocks_per_cu = 0;
  int64_t grid_override = 0;
  int64_t chunk_sites = 0;
  int engine = FAMSEQ_ENGINE_ENUM;
  JitKernel elim{};      // generated sum-product kernel (engine = FAMSEQ_ENGINE_ELIM)
  JitKernel elim_call{}, lane_call{};  // their fused call-path forms (famseq_bn_call_batch), built on first use
  int elim_call_blocks_per_cu = 0, lane_call_blocks_per_cu = 0;
  int elim_blocks_per_cu = 0;
  int elim_variant = -1, lane_variant = -1;  // which generator variant jit_pick_variant took
  // enumeration engine: the team-per-site kernel is compiled into the library; the lane-per-site
  // kernel is generated per pedigree.  enum_impl: -1 auto (lane for large batches), 0 team, 1 lane
  int enum_impl = -1;
  JitKernel lane{};  // one lane per site (large batches)
  int lane_blocks_per_cu = 0;
  bool lane_failed = false;  // the plain one-lane-per-site kernel could not be built (no compiler at run time, ...)
  std::string lane_error;
  // every other generated kernel keeps its own verdict: a call-path form that does not build must not take the plain
  // kernels down with it, nor the other way round
  bool grp_failed[kEnumMaxGroupDigits + 1] = {};
  bool call_failed[2] = {false, false};  // [0] lane call-path form, [1] sum-product call-path form
  std::string call_error[2];
  // lanes-per-site mode of the same generator for batches too small to give every lane of the chip a
  // site: grp[d] lets 3^d lanes share a site (d = 1..4 of the outermost looped members' digits on
  // lanes).  Compiled on first use of each d.  group_digits: -1 auto (by batch size), 0..4 forced.
  JitKernel grp[kEnumMaxGroupDigits + 1]{};
  int grp_blocks_per_cu[kEnumMaxGroupDigits + 1] = {};
  int group_digits = -1, last_group_digits = 0;
  int lane_reads_rows = -1;  // does the lane call-path kernel re-read fp64 rows from global memory (unknown until asked)
  int lane_call_variant = -1;  // the variant jit_pick_variant took for it (the answer depends on the variant)
  int64_t lane_min_sites = 256;  // below this the compiled-in team kernel answers at once (no per-pedigree compile for tiny calls) ...
  // ... unless the generated kernel for that batch is loaded or on disk already (a pre-built pedigree, or one this user has run
  // before): then nothing has to be waited for and it serves every batch size (team kernel: 0.0237 ms per 256 ten-member
  // sites, three lanes ... 81 lanes per site: 0.0159).  -1 unknown, 0 would have to be compiled, 1 ready.
  int grp_ready[kEnumMaxGroupDigits + 1] = {-1, -1, -1, -1, -1};
  // device constants
  uint32_t *d_img = nullptr;
  double *d_tc = nullptr;
  // staging for the host-buffer entry point: a three-stage pipeline (copy in / compute / copy out,
  // one stream each, so both directions of the host link run at once) over two buffer slots
  static constexpr int kSlots = 2;
  static constexpr int kStages = 3;
  hipStream_t stream[kStages] = {nullptr, nullptr, nullptr};  // 0 copy in, 1 compute, 2 copy out
  hipEvent_t ev_in[kSlots] = {}, ev_done[kSlots] = {}, ev_out[kSlots] = {};
  int64_t slot_sites = 0;
  double *d_lk[kSlots] = {}, *d_post[kSlots] = {}, *d_single[kSlots] = {};
  uint8_t *d_flags[kSlots] = {}, *d_status[kSlots] = {};
  // packed input / called output stages (famseq_bn_call_batch*)
  int slot_seq = 0;
  uint16_t *d_pl[kSlots] = {};
  double *d_gpp[kSlots] = {}, *d_fpp[kSlots] = {};
  int8_t *d_fgt[kSlots] = {};
  char *d_text[kSlots] = {};  // text records of the called outputs (famseq_bn_call_text_batch)
  double *d_lut = nullptr;
  int32_t *d_seq = nullptr, *d_col = nullptr, *d_slot = nullptr;  // column -> member; member -> column or -1; member -> output slot
  CallIO *d_call[kSlots] = {};  // the generated kernels' call-path arguments, one per slot
  unsigned long long *d_phase = nullptr;  // FAMSEQ_PHASE_CLOCK: kPhases counters
  // device-resident call path (famseq_bn_call_batch_device): its own argument block, what it holds, and scratch rows for
  // batches the fused kernels do not serve (separate unpack / posterior / Phred stages)
  CallIO *d_call_dev = nullptr;
  CallIO call_dev_host{};
  bool call_dev_valid = false;
  double *dev_tmp[5] = {};  // lk, post, single, gpp, fpp
  int8_t *dev_tmp_fgt = nullptr;
  uint8_t *dev_tmp_status = nullptr;
  int64_t dev_tmp_sites = 0;
  int dev_tmp_seq = 0;
  std::vector<int32_t> seq_members;
  std::string tune_report;  // what famseq_set_option "tune" measured (famseq_plan_json "tune")
  std::string err, json;
};

namespace {

void set_err(char *err, size_t n, const std::string &msg) {
  if (err && n) {
    std::snprintf(err, n, "%s", msg.c_str());
  }
}

int fail(famseq_ctx *c, int code, const std::string &msg) {
  c->err = msg;
  return code;
}

#define HIP_TRY(c, call)                                                                              \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail((c), FAMSEQ_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
  } while (0)

void validate_model(const Model &m) {
  const int n = m.n_members;
  if (n < 1) throw std::runtime_error("n_members must be >= 1 (and the member arrays given)");
  if ((int)m.mother.size() != n || (int)m.father.size() != n || (int)m.gender.size() != n || (int)m.sequenced.size() != n)
    throw std::runtime_error("member arrays do not match n_members");
  for (int i = 0; i < n; ++i) {
    const int mo = m.mother[i], fa = m.father[i];
    if ((mo < 0) != (fa < 0)) throw std::runtime_error("member with exactly one known parent");
    if (mo >= n || fa >= n) throw std::runtime_error("parent index out of range");
  }
  // nobody may be their own ancestor: generation numbers must settle within n rounds
  std::vector<int> depth(n, 0);
  for (int pass = 0; pass <= n; ++pass) {
    bool moved = false;
    for (int i = 0; i < n; ++i)
      if (m.mother[i] >= 0) {
        const int d = 1 + std::max(depth[m.mother[i]], depth[m.father[i]]);
        if (d > depth[i]) {
          depth[i] = d;
          moved = true;
        }
      }
    if (!moved) return;
  }
  throw std::runtime_error("pedigree has a member who is their own ancestor");
}

void free_slots(famseq_ctx *c) {
  for (int s = 0; s < famseq_ctx::kSlots; ++s) {
    if (c->d_lk[s]) (void)hipFree(c->d_lk[s]);
    if (c->d_post[s]) (void)hipFree(c->d_post[s]);
    if (c->d_single[s]) (void)hipFree(c->d_single[s]);
    if (c->d_flags[s]) (void)hipFree(c->d_flags[s]);
    if (c->d_status[s]) (void)hipFree(c->d_status[s]);
    if (c->d_pl[s]) (void)hipFree(c->d_pl[s]);
    if (c->d_gpp[s]) (void)hipFree(c->d_gpp[s]);
    if (c->d_fpp[s]) (void)hipFree(c->d_fpp[s]);
    if (c->d_fgt[s]) (void)hipFree(c->d_fgt[s]);
    if (c->d_text[s]) (void)hipFree(c->d_text[s]);
    c->d_lk[s] = c->d_post[s] = c->d_single[s] = nullptr;
    c->d_flags[s] = c->d_status[s] = nullptr;
    c->d_pl[s] = nullptr;
    c->d_gpp[s] = c->d_fpp[s] = nullptr;
    c->d_fgt[s] = nullptr;
    c->d_text[s] = nullptr;
  }
  c->slot_sites = 0;
  c->slot_seq = 0;
}

// (Re)build the plan and, on a device ctx, upload its image and the factor tables.
int refresh_plan(famseq_ctx *c) {
  if (!c->plan_dirty) return 0;
  if (!c->big) {
    try {
      c->plan = build_plan(c->model, c->opt);
    } catch (const std::exception &e) {
      return fail(c, FAMSEQ_E_ARG, std::string("plan: ") + e.what());
    }
    c->kp = make_kparams(c->plan, c->model.lc);
  }
  if (c->device >= 0) {
    HIP_TRY(c, hipSetDevice(c->device));
    double tc[4 * 4 * 27];
    build_factor_tables(c->model, tc);
    if (!c->d_tc) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_tc), sizeof tc));
    HIP_TRY(c, hipMemcpy(c->d_tc, tc, sizeof tc, hipMemcpyHostToDevice));
  }
  if (c->device >= 0 && !c->big) {
    std::vector<uint32_t> img = c->plan.device_image();
    for (int i = 0; i < c->model.n_members; ++i)
      img[c->kp.off_minfo + i] |= uint32_t(c->model.sequenced[i] ? 1 : 0) << 2;
    if (c->d_img) (void)hipFree(c->d_img);
    c->d_img = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_img), img.size() * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(c->d_img, img.data(), img.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipError_t e = hipSuccess;
    c->blocks_per_cu = bn_enum_blocks_per_cu(c->plan, &e);
    if (c->blocks_per_cu < 1)
      return fail(c, FAMSEQ_E_HIP, std::string("occupancy query: ") + hipGetErrorString(e));
  }
  c->plan_dirty = false;
  return 0;
}

int grid_for(const famseq_ctx *c, int64_t n_sites) {
  const int64_t passes = (n_sites + c->plan.teams_per_block - 1) / c->plan.teams_per_block;
  int64_t resident = c->grid_override > 0 ? c->grid_override : int64_t(c->n_cus) * c->blocks_per_cu;
  return (int)std::max<int64_t>(1, std::min(passes, resident));
}

// Generate, compile (or fetch) and load the elimination kernel for this model.
int load_elim(famseq_ctx *c) {
  if (c->elim.fn || (c->device < 0 && !c->elim.path.empty())) return 0;
  std::string why;
  if (!elim_supported(c->model, &why)) return fail(c, FAMSEQ_E_ARG, "elimination engine: " + why);
  try {
    const Model &mdl = c->model;
    // a measured pick (the autotuner's note, or the table build() ships) is loaded as it is: the spill contest below is
    // the static rule for pedigrees nobody has measured, and must not move off a measurement
    const int pick = jit_read_pick(elim_source(mdl, 0));
    std::string src;
    if (pick >= 0 && pick < kElimVariants) {
      src = elim_source(mdl, pick);
      c->elim_variant = pick;
    } else {
      src = jit_pick_variant([&mdl](int v) { return elim_source(mdl, v); }, kElimVariants, &c->elim_variant, elim_first_variant(mdl));
    }
    if (c->device < 0) {  // plan-only ctx: generate and compile into the cache (this is how build() pre-builds)
      c->elim.path = jit_compile(src);
      return 0;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    c->elim = jit_load(src, "famseq_elim");
  } catch (const std::exception &e) {
    return fail(c, FAMSEQ_E_HIP, e.what());
  }
  int nb = 0;
  HIP_TRY(c, hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, c->elim.fn, elim_block_threads(c->model), 0));
  c->elim_blocks_per_cu = nb > 0 ? nb : 1;
  return 0;
}

// Generate/compile/load the lane-per-site enumeration kernel (once).  Returns false when it is
// unavailable (no compiler at run time, ...): the compiled-in team kernel then serves all batches.
// d > 0: the lanes-per-site form with 3^d lanes per site.
bool load_lane(famseq_ctx *c, int d = 0) {
  JitKernel &k = d == 0 ? c->lane : c->grp[d];
  if (k.fn) return true;
  if (d == 0 ? c->lane_failed : c->grp_failed[d]) return false;
  if (c->device < 0 && !k.path.empty()) return true;
  try {
    const Model &mdl = c->model;
    int variant = -1;
    std::string src;
    const int pick = d == 0 ? jit_read_pick(enumgen_source(mdl, 0, 0)) : -1;  // the autotuner's note (7- or 6-member block)
    if (pick >= 0 && pick < kEnumVariants) {  // measured: exactly that variant (see load_elim)
      src = enumgen_source(mdl, pick, d);
      variant = pick;
    } else {
      src = jit_pick_variant([&mdl, d](int v) { return enumgen_source(mdl, v, d); }, kEnumVariants, &variant, 0);
    }
    if (d == 0) c->lane_variant = variant;
    if (c->device < 0) {
      k.path = jit_compile(src);
      return true;
    }
    if (hipSetDevice(c->device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    k = jit_load(src, "famseq_enum_lane");
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.fn, enumgen_block_threads(c->model, d), 0) != hipSuccess)
      nb = 1;
    (d == 0 ? c->lane_blocks_per_cu : c->grp_blocks_per_cu[d]) = nb > 0 ? nb : 1;
    return true;
  } catch (const std::exception &e) {
    if (d == 0) c->lane_failed = true;
    else c->grp_failed[d] = true;
    if (d == 0 || c->lane_error.empty()) c->lane_error = e.what();
    // said once per ctx and kernel, where a user of the CLI or of the library sees it (also in famseq_plan_json)
    if (!std::getenv("FAMSEQ_QUIET"))
      std::fprintf(stderr, "famseq: the per-pedigree enumeration kernel%s is unavailable (%s); %s\n", d ? " (lanes-per-site form)" : "",
                   std::string(e.what()).substr(0, 300).c_str(),
                   d ? "such batches take the one-lane-per-site or the compiled-in kernel"
                     : "large batches fall back to the compiled-in team-per-site kernel (about 4x slower)");
    return false;
  }
}

// How many of the outermost looped members' digits go on lanes for a batch of n_sites.  Cost model
// (measured on the 10-member benchmark pedigree, tools/small_batch_rates.py): a lane's work is its
// share of the enumeration, 3^N / 3^d configurations, plus what every lane of a group repeats (single
// posterior, tables of the fixed levels, its columns of the reduction: about 250 N configuration
// times); lanes run at full speed while there is at most one wave per SIMD (n_cus * 256 lanes), beyond
// that the time grows with the lane count.  More lanes per site pay while the batch leaves SIMDs idle.
int pick_group_digits(const famseq_ctx *c, int64_t n_sites) {
  const int dmax = enumgen_max_group_digits(c->model);
  if (c->group_digits >= 0) return std::min(c->group_digits, dmax);
  if (c->enum_impl == 1) return 0;  // an explicit choice of the lane-per-site kernel is exactly that kernel
  const double full_speed_lanes = double(std::max(1, c->n_cus)) * 256.0;
  const double configs = std::pow(3.0, c->model.n_members), per_lane = 250.0 * c->model.n_members;
  int best = 0;
  double best_t = 0;
  for (int d = 0, g = 1; d <= dmax; ++d, g *= 3) {
    const double t = std::max(1.0, double(n_sites) * g / full_speed_lanes) * (configs / g + (d ? per_lane : 0.0));
    if (d == 0 || t < best_t * 0.9) {
      best = d;
      best_t = t;
    }
  }
  return best;
}

hipError_t launch_generated(famseq_ctx *c, hipFunction_t fn, int bt, int blocks_per_cu, int64_t n_sites,
                            const double *d_lk, const uint8_t *d_flags, double *d_post, double *d_single,
                            uint8_t *d_status, hipStream_t stream, int sites_per_chunk = 0, const CallIO *d_call = nullptr) {
  const int spc = sites_per_chunk > 0 ? sites_per_chunk : bt;
  const int64_t chunks = (n_sites + spc - 1) / spc;
  int64_t resident = c->grid_override > 0 ? c->grid_override : int64_t(c->n_cus) * blocks_per_cu;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min(chunks, resident));
  long ns = (long)n_sites;
  double lc = c->model.lc;
  const double *tc = c->d_tc;
  void *args[] = {&d_lk, &d_flags, &d_post, &d_single, &d_status, &ns, &tc, &lc, &d_call};  // the plain forms take the first eight
  return hipModuleLaunchKernel(fn, grid, 1, 1, (unsigned)bt, 1, 1, 0, stream, args, nullptr);
}

hipError_t launch_elim(famseq_ctx *c, int64_t n_sites, const double *d_lk, const uint8_t *d_flags, double *d_post,
                       double *d_single, uint8_t *d_status, hipStream_t stream) {
  return launch_generated(c, c->elim.fn, elim_block_threads(c->model), c->elim_blocks_per_cu, n_sites, d_lk, d_flags,
                          d_post, d_single, d_status, stream);
}

// Can the generated kernel with 3^d lanes per site run without a compilation (loaded, or its code object on disk)?
bool generated_ready(famseq_ctx *c, int d) {
  if ((d == 0 ? c->lane : c->grp[d]).fn) return true;
  if (d == 0 ? c->lane_failed : c->grp_failed[d]) return false;
  if (c->grp_ready[d] < 0) {
    try {
      const int pick = d == 0 ? jit_read_pick(enumgen_source(c->model, 0, 0)) : -1;
      // (the variant the loader would take first; a spilling first variant sends it on to others, which may need the compiler:
      // then this says "not ready" and the tiny batch stays on the compiled-in kernel, which is always right)
      c->grp_ready[d] = jit_cached(enumgen_source(c->model, pick >= 0 && pick < kEnumVariants ? pick : 0, d)) ? 1 : 0;
    } catch (const std::exception &) {
      c->grp_ready[d] = 0;
    }
  }
  return c->grp_ready[d] == 1;
}

hipError_t launch_engine(famseq_ctx *c, int64_t n_sites, const double *d_lk, const uint8_t *d_flags, double *d_post,
                         double *d_single, uint8_t *d_status, hipStream_t stream) {
  if (c->engine == FAMSEQ_ENGINE_ELIM) return launch_elim(c, n_sites, d_lk, d_flags, d_post, d_single, d_status, stream);
  const bool want_lane = c->enum_impl == 1 || (c->enum_impl < 0 && (n_sites >= c->lane_min_sites ||
                                                                    generated_ready(c, pick_group_digits(c, n_sites))));
  if (want_lane) {
    int d = pick_group_digits(c, n_sites);
    if (d > 0 && !load_lane(c, d)) d = 0;  // that group size does not build: one lane per site before the compiled-in kernel
    if (load_lane(c, d)) {
      c->last_group_digits = d;
      if (d > 0)
        return launch_generated(c, c->grp[d].fn, enumgen_block_threads(c->model, d), c->grp_blocks_per_cu[d], n_sites, d_lk,
                                d_flags, d_post, d_single, d_status, stream, enumgen_sites_per_chunk(c->model, d));
      return launch_generated(c, c->lane.fn, enumgen_block_threads(c->model), c->lane_blocks_per_cu, n_sites, d_lk, d_flags,
                              d_post, d_single, d_status, stream);
    }
  }
  return launch_bn_enum(c->plan, c->kp, grid_for(c, n_sites), c->d_img, c->d_tc, n_sites, d_lk, d_flags, d_post,
                        d_single, d_status, stream);
}

// The fused call path: one launch does PL -> likelihood, posterior, Phred scaling and the genotype call
// (famseq_bn_call_batch).  Served by the generated kernels in their one-lane-per-site form; returns
// false when this batch goes through the separate stages instead (team kernel, lanes-per-site mode, or
// a lane kernel that re-reads fp64 rows from global memory while the input is packed).
// Load (once) the call-path form of a generated kernel.  False when it cannot be built.
bool load_call_kernel(famseq_ctx *c, bool elim) {
  JitKernel &k = elim ? c->elim_call : c->lane_call;
  if (k.fn) return true;
  if (c->call_failed[elim]) return false;
  if (c->device < 0 && !k.path.empty()) return true;
  try {
    const Model &mdl = c->model;
    std::string src;
    if (elim) {
      src = jit_pick_variant([&mdl](int v) { return elim_source(mdl, v, true); }, kElimCallVariants, nullptr, elim_first_variant(mdl, true));
    } else {
      // the same block size as the plain lane kernel runs with (variants 0-1 / 2-3: kEnumVariants), so that a batch
      // gives the same bits whether it goes through the fused kernel or through the separate stages
      if (!load_lane(c, 0)) throw std::runtime_error("the plain lane kernel is unavailable: " + c->lane_error);
      const int base = c->lane_variant >= 0 ? (c->lane_variant & ~1) : 0;
      int pick = 0;
      // v & 1: the single posterior fenced member by member; v & 2: the leaner stage-out (see kElimCallVariants)
      src = jit_pick_variant([&mdl, base](int v) { return enumgen_source(mdl, base + (v & 1), 0, true, !(v & 2)); }, 4, &pick);
      c->lane_call_variant = base + (pick & 1);
      c->lane_reads_rows = enumgen_reads_global_rows(mdl, c->lane_call_variant) ? 1 : 0;
    }
    if (c->device < 0) {
      k.path = jit_compile(src);
      return true;
    }
    if (hipSetDevice(c->device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    k = jit_load(src, elim ? "famseq_elim" : "famseq_enum_lane");
    int nb = 0;
    const int bt = elim ? elim_block_threads(c->model, true) : enumgen_block_threads(c->model);
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.fn, bt, 0) != hipSuccess) nb = 1;
    (elim ? c->elim_call_blocks_per_cu : c->lane_call_blocks_per_cu) = nb > 0 ? nb : 1;
    return true;
  } catch (const std::exception &e) {
    c->call_failed[elim] = true;
    c->call_error[elim] = e.what();
    if (!std::getenv("FAMSEQ_QUIET"))
      std::fprintf(stderr, "famseq: the fused call-path form of the %s kernel is unavailable (%s); famseq_bn_call_batch runs the "
                           "separate unpack / posterior / Phred stages instead (same results)\n",
                   elim ? "sum-product" : "enumeration", c->call_error[elim].substr(0, 300).c_str());
    return false;
  }
}

// The fused call path: one launch does PL -> likelihood, posterior, Phred scaling and the genotype call
// (famseq_bn_call_batch).  Served by the call-path forms of the generated kernels (one lane per site);
// returns false when this batch goes through the separate stages instead (team kernel, lanes-per-site
// mode, or a lane kernel that re-reads fp64 rows from global memory while the input is packed).
// Does this batch go through a generated kernel's call-path form (loading it on first use)?  The one place that decides.
bool call_fuses(famseq_ctx *c, int64_t n_sites, bool packed_in) {
  const bool elim = c->engine == FAMSEQ_ENGINE_ELIM;
  if (c->big) return false;  // no call-path form of the wide-pedigree kernel: separate stages
  if (!elim) {
    const bool want_lane = c->enum_impl == 1 || (c->enum_impl < 0 && n_sites >= c->lane_min_sites);
    if (!want_lane || pick_group_digits(c, n_sites) != 0) return false;
  }
  if (!load_call_kernel(c, elim)) return false;
  if (!elim && packed_in && c->lane_reads_rows != 0) return false;  // (set by load_call_kernel for the variant it took)
  return true;
}

bool launch_engine_fused(famseq_ctx *c, int64_t n_sites, const double *d_lk, const uint8_t *d_flags, uint8_t *d_status,
                         bool packed_in, const CallIO *d_io, hipStream_t stream, hipError_t *err) {
  const bool elim = c->engine == FAMSEQ_ENGINE_ELIM;
  if (!call_fuses(c, n_sites, packed_in)) return false;
  if (!elim) c->last_group_digits = 0;
  *err = launch_generated(c, elim ? c->elim_call.fn : c->lane_call.fn, elim ? elim_block_threads(c->model, true) : enumgen_block_threads(c->model),
                          elim ? c->elim_call_blocks_per_cu : c->lane_call_blocks_per_cu, n_sites, d_lk, d_flags, nullptr, nullptr,
                          d_status, stream, 0, d_io);
  return true;
}

}  // namespace

extern "C" int famseq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {
famseq_ctx *create_ctx(const Model &model, int device_id, char *err, size_t errlen);
}

extern "C" famseq_ctx *famseq_create(const famseq_model *model, int device_id, char *err, size_t errlen) {
  if (!model) {
    set_err(err, errlen, "model is NULL");
    return nullptr;
  }
  if (model->n_members < 1 || model->n_members > FAMSEQ_MAX_MEMBERS) {
    set_err(err, errlen, "n_members must be 1..20 (famseq_model; famseq_pedigree has no limit)");
    return nullptr;
  }
  return create_ctx(Model(*model), device_id, err, errlen);
}

extern "C" famseq_ctx *famseq_create_pedigree(const famseq_pedigree *pedigree, int device_id, char *err, size_t errlen) {
  if (!pedigree) {
    set_err(err, errlen, "pedigree is NULL");
    return nullptr;
  }
  return create_ctx(Model(*pedigree), device_id, err, errlen);
}

namespace {
famseq_ctx *create_ctx(const Model &model, int device_id, char *err, size_t errlen) {
  famseq_ctx *c = new famseq_ctx;
  c->model = model;
  c->big = model.n_members > FAMSEQ_MAX_MEMBERS;
  try {
    validate_model(c->model);
  } catch (const std::exception &e) {
    set_err(err, errlen, e.what());
    delete c;
    return nullptr;
  }
  if (device_id >= 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device_id >= n) {
      set_err(err, errlen, "no usable HIP device " + std::to_string(device_id) + " (" +
                               (e != hipSuccess ? hipGetErrorString(e) : "device_id out of range") + ")");
      delete c;
      return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
      set_err(err, errlen, std::string("hipSetDevice: ") + hipGetErrorString(e));
      delete c;
      return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      set_err(err, errlen, std::string("device is ") + prop.gcnArchName + ", this library only carries gfx950 code");
      delete c;
      return nullptr;
    }
    c->device = device_id;
    c->n_cus = prop.multiProcessorCount;
    for (int s = 0; s < famseq_ctx::kStages; ++s)
      if ((e = hipStreamCreateWithFlags(&c->stream[s], hipStreamNonBlocking)) != hipSuccess) {
        set_err(err, errlen, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        famseq_destroy(c);
        return nullptr;
      }
    for (int s = 0; s < famseq_ctx::kSlots; ++s)
      for (hipEvent_t *ev : {&c->ev_in[s], &c->ev_done[s], &c->ev_out[s]})
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) {
          set_err(err, errlen, std::string("hipEventCreate: ") + hipGetErrorString(e));
          famseq_destroy(c);
          return nullptr;
        }
  }
  if (refresh_plan(c) != 0) {
    set_err(err, errlen, c->err);
    famseq_destroy(c);
    return nullptr;
  }
  if (c->big) {  // beyond the enumeration's reach: the sum-product engine or nothing
    if (load_elim(c) != 0) {
      set_err(err, errlen, "a pedigree of " + std::to_string(model.n_members) + " members is served by the sum-product engine only: " + c->err);
      famseq_destroy(c);
      return nullptr;
    }
    c->engine = FAMSEQ_ENGINE_ELIM;
  }
  return c;
}
}  // namespace

extern "C" void famseq_destroy(famseq_ctx *c) {
  if (!c) return;
  if (c->device >= 0) {
    (void)hipSetDevice(c->device);
    free_slots(c);
    if (c->d_img) (void)hipFree(c->d_img);
    if (c->d_tc) (void)hipFree(c->d_tc);
    jit_unload(c->elim);
    jit_unload(c->lane);
    jit_unload(c->elim_call);
    jit_unload(c->lane_call);
    for (JitKernel &k : c->grp) jit_unload(k);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_seq) (void)hipFree(c->d_seq);
    if (c->d_col) (void)hipFree(c->d_col);
    if (c->d_slot) (void)hipFree(c->d_slot);
    for (int s = 0; s < famseq_ctx::kSlots; ++s)
      if (c->d_call[s]) (void)hipFree(c->d_call[s]);
    if (c->d_call_dev) (void)hipFree(c->d_call_dev);
    if (c->d_phase) (void)hipFree(c->d_phase);
    for (double *q : c->dev_tmp)
      if (q) (void)hipFree(q);
    if (c->dev_tmp_fgt) (void)hipFree(c->dev_tmp_fgt);
    if (c->dev_tmp_status) (void)hipFree(c->dev_tmp_status);
    for (int s = 0; s < famseq_ctx::kStages; ++s)
      if (c->stream[s]) (void)hipStreamDestroy(c->stream[s]);
    for (int s = 0; s < famseq_ctx::kSlots; ++s)
      for (hipEvent_t ev : {c->ev_in[s], c->ev_done[s], c->ev_out[s]})
        if (ev) (void)hipEventDestroy(ev);
  }
  delete c;
}

extern "C" const char *famseq_last_error(famseq_ctx *c) { return c ? c->err.c_str() : "ctx is NULL"; }

namespace {

// famseq_set_option(ctx, "tune", 1): where static rules pick a generated kernel's variant (the sum-product kernel's
// fence variant by pedigree size, the enumeration kernel's 7- or 6-member block), time the candidates on THIS device
// and pedigree — synthetic rows, a few milliseconds each — and leave the winner's index as a note in the kernel
// cache; every later context for the pedigree starts from it.  Opt-in: it compiles every candidate.
int tune(famseq_ctx *c) {
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "tuning times kernels: it needs a device");
  HIP_TRY(c, hipSetDevice(c->device));
  const Model &mdl = c->model;
  const int N = mdl.n_members;
  // about 10 ms of enumeration per launch, 64 K - 2 M sites; the sum-product kernel, whose time does not grow with
  // 3^N, always gets 8 M (at 64 K sites its launch is most of what a timer sees)
  // ... in whole ROUNDS of the chip: a candidate at one wave per SIMD takes 64 K sites at a time, one at two waves 128 K;
  // a batch of 2.3 rounds times the tail, not the kernel (a thirteen-member pedigree's two blocks came out 25 % apart that
  // way).  Up to eight rounds while a launch stays under a quarter of a second.
  const double configs = std::pow(3.0, N), t_round = 65536.0 * configs / 2.4e13;
  int64_t rounds = std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)(0.25 / t_round)));
  if (rounds > 1) rounds &= ~int64_t(1);
  const int64_t by_time = (int64_t)(0.01 * 2.4e13 / configs) / 131072 * 131072;
  const int64_t n_enum = std::min<int64_t>(int64_t(1) << 21, std::max<int64_t>(by_time, 65536 * rounds));
  const int64_t n_elim = int64_t(1) << 23, n_max = std::max(n_enum, n_elim);  // 8 M: it has to stream from HBM (2 M sites half fit the Infinity Cache)
  int64_t n = n_enum;  // sites of the launches being timed
  const size_t w = size_t(n_max) * 3 * N;
  double *d_lk = nullptr, *d_post = nullptr, *d_single = nullptr;
  uint8_t *d_status = nullptr;
  auto release = [&] {
    for (void *q : {(void *)d_lk, (void *)d_post, (void *)d_single, (void *)d_status})
      if (q) (void)hipFree(q);
  };
  if (hipMalloc(reinterpret_cast<void **>(&d_lk), w * 8) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d_post), w * 8) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&d_single), w * 8) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d_status), size_t(n_max)) != hipSuccess) {
    release();
    return fail(c, FAMSEQ_E_HIP, "tune: device buffers");
  }
  {
    // PL-shaped rows — one genotype at 1, the others 10^-(k/10) — for the first 64 K sites, doubled on the device from there
    const size_t w0 = std::min(w, size_t(1 << 16) * 3 * N);
    std::vector<double> h(w0);
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < w0; i += 3) {
      z = z * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned a = unsigned(z >> 33) % 3, p1 = 3 + unsigned(z >> 40) % 88, p2 = p1 + unsigned(z >> 50) % 160;
      h[i + a] = 1.0, h[i + (a + 1) % 3] = std::pow(10.0, -0.1 * p1), h[i + (a + 2) % 3] = std::pow(10.0, -0.1 * p2);
    }
    bool up = hipMemcpy(d_lk, h.data(), w0 * 8, hipMemcpyHostToDevice) == hipSuccess;
    for (size_t have = w0; up && have < w; have *= 2)
      up = hipMemcpy(d_lk + have, d_lk, std::min(have, w - have) * 8, hipMemcpyDeviceToDevice) == hipSuccess;
    if (!up) {
      release();
      return fail(c, FAMSEQ_E_HIP, "tune: upload");
    }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::string report;
  // best of three launches of one candidate, ms (< 0: it could not be built)
  auto time_one = [&](const std::string &src, const char *entry, int bt) {
    JitKernel k;
    try {
      k = jit_load(src, entry);
    } catch (const std::exception &) {
      return -1.0;
    }
    int nb = 1;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.fn, bt, 0) != hipSuccess || nb < 1) nb = 1;
    double best = -1;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0, c->stream[1]);
      const hipError_t e = launch_generated(c, k.fn, bt, nb, n, d_lk, nullptr, d_post, d_single, d_status, c->stream[1]);
      (void)hipEventRecord(e1, c->stream[1]);
      if (e != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
        best = -1;
        break;
      }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && (best < 0 || ms < best)) best = ms;  // the first launch warms up
    }
    jit_unload(k);
    return best;
  };
  auto race = [&](const char *what, const std::vector<int> &cands, const std::function<std::string(int)> &gen, const char *entry, int bt) {
    int win = -1;
    double win_ms = 0;
    report += std::string(report.empty() ? "" : "; ") + what + ":";
    for (int v : cands) {
      const double ms = time_one(gen(v), entry, bt);
      char buf[64];
      std::snprintf(buf, sizeof buf, " v%d %.4f ms", v, ms);
      report += buf;
      if (ms > 0 && (win < 0 || ms < win_ms * 0.97)) win = v, win_ms = ms;  // a later candidate has to win by 3 % (two runs of one
                                                                            // table disagreed on 17 of 78 pedigrees at 1 %: all within 2 %)
    }
    if (win >= 0) {
      jit_write_pick(gen(0), win);
      report += " -> v" + std::to_string(win);
    }
    return win;
  };
  try {
    if (enumgen_describe(mdl, 0) != enumgen_describe(mdl, 2))
      (void)race("enumeration (7- / 6-member block)", {0, 2}, [&mdl](int v) { return enumgen_source(mdl, v, 0); }, "famseq_enum_lane",
                 enumgen_block_threads(mdl, 0));
    else
      report += "enumeration: one block shape, nothing to choose";
    n = n_elim;
    if (elim_supported(mdl, nullptr))
      (void)race("sum-product (likelihoods re-read from LDS: fence-free, fenced; in registers: fence-free, fenced)", {0, 1, 4, 5},
                 [&mdl](int v) { return elim_source(mdl, v); }, "famseq_elim", elim_block_threads(mdl));
  } catch (const std::exception &e) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    release();
    return fail(c, FAMSEQ_E_HIP, std::string("tune: ") + e.what());
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  release();
  c->tune_report = report + " (synthetic sites per launch: " + std::to_string(n_enum) + " enumeration, " + std::to_string(n_elim) + " sum-product)";
  // the kernels this context holds may have lost: drop them, the next use loads the picks
  const bool had_lane = c->lane.fn != nullptr, had_elim = c->elim.fn != nullptr, had_lc = c->lane_call.fn != nullptr;
  jit_unload(c->lane), jit_unload(c->lane_call), jit_unload(c->elim);
  c->lane.path.clear(), c->lane_call.path.clear(), c->elim.path.clear();
  c->lane_variant = c->lane_call_variant = c->elim_variant = -1;
  c->lane_reads_rows = -1;
  if (had_lane && !load_lane(c)) return fail(c, FAMSEQ_E_HIP, "lane kernel unavailable after tuning: " + c->lane_error);
  if (had_lc && !load_call_kernel(c, false)) return fail(c, FAMSEQ_E_HIP, "call-path kernel unavailable after tuning: " + c->call_error[0]);
  if (had_elim || c->engine == FAMSEQ_ENGINE_ELIM) {
    const int rc = load_elim(c);
    if (rc != 0) return rc;
  }
  if (c->lane_variant >= 0 || c->elim_variant >= 0)  // what this context runs from here on (the notes are honoured as written)
    c->tune_report += "; loaded:" + (c->lane_variant >= 0 ? " enumeration v" + std::to_string(c->lane_variant) : std::string()) +
                      (c->elim_variant >= 0 ? " sum-product v" + std::to_string(c->elim_variant) : std::string());
  return 0;
}

}  // namespace

namespace {
// FAMSEQ_PHASE_CLOCK on the plain kernels: their counters sit in a module global; print and clear them.
void report_phase_clock(famseq_ctx *c) {
  for (JitKernel *k : {&c->elim, &c->lane}) {
    if (!k->module) continue;
    hipDeviceptr_t p = nullptr;
    size_t bytes = 0;
    if (hipModuleGetGlobal(&p, &bytes, k->module, "fs_phase_clk") != hipSuccess || bytes < 8 * sizeof(unsigned long long)) continue;
    unsigned long long ph[8] = {};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(ph, p, sizeof ph, hipMemcpyDeviceToHost) != hipSuccess) continue;
    unsigned long long tot = 0;
    for (unsigned long long v : ph) tot += v;
    std::fprintf(stderr, "famseq phase clock, %s kernel (wave cycles):", k == &c->elim ? "sum-product" : "enumeration");
    for (int i = 0; i < 8; ++i) std::fprintf(stderr, " [%d] %.1f%%", i, tot ? 100.0 * double(ph[i]) / double(tot) : 0.0);
    std::fprintf(stderr, "  total %llu\n", tot);
    (void)hipMemset(p, 0, sizeof ph);
  }
}
}  // namespace

extern "C" int famseq_set_option(famseq_ctx *c, const char *key, int64_t value) {
  if (!c || !key) return FAMSEQ_E_ARG;
  const std::string k(key);
  if (c->big) {  // no enumeration here: its knobs have nothing to act on
    for (const char *e : {"fixed_digits", "low_members", "block_threads", "enum_impl", "group_digits", "pick_lane", "tune", "lane_min_sites"})
      if (k == e) return fail(c, FAMSEQ_E_ARG, "option " + k + " belongs to the enumeration engine, which serves up to 20 members");
    if (k == "engine" && value == FAMSEQ_ENGINE_ENUM)
      return fail(c, FAMSEQ_E_ARG, "the 3^N enumeration serves up to 20 members; this pedigree has " + std::to_string(c->model.n_members));
    if (k == "call_kernels") return 0;  // (the call path of such a pedigree runs as separate stages)
  }
  PlanOptions saved = c->opt;
  if (k == "fixed_digits") c->opt.fixed_digits = (int)value;
  else if (k == "low_members") c->opt.low_members = (int)value;
  else if (k == "block_threads") c->opt.block_threads = (int)value;
  else if (k == "grid_blocks") { c->grid_override = value; return 0; }
  else if (k == "enum_impl") {
    if (value < -1 || value > 1) return fail(c, FAMSEQ_E_ARG, "enum_impl must be -1 (auto), 0 (team kernel) or 1 (lane kernel)");
    if (value == 1 && !load_lane(c)) return fail(c, FAMSEQ_E_HIP, "lane-per-site kernel unavailable: " + c->lane_error);
    c->enum_impl = (int)value;
    return 0;
  }
  else if (k == "lane_min_sites") { c->lane_min_sites = value; return 0; }
  else if (k == "tune") {
    if (value != 1) return fail(c, FAMSEQ_E_ARG, "tune takes 1");
    return tune(c);
  }
  else if (k == "pick_lane" || k == "pick_elim") {  // a variant measured elsewhere (the table build() ships): the note tune() would leave
    const bool lane = k == "pick_lane";
    if (value < 0 || value >= (lane ? kEnumVariants : kElimVariants)) return fail(c, FAMSEQ_E_ARG, k + " takes a variant index");
    if (!lane && !elim_supported(c->model, nullptr)) return fail(c, FAMSEQ_E_ARG, "pick_elim: the sum-product engine does not serve this pedigree");
    try {
      jit_write_pick(lane ? enumgen_source(c->model, 0, 0) : elim_source(c->model, 0), (int)value);
    } catch (const std::exception &e) {
      return fail(c, FAMSEQ_E_HIP, e.what());
    }
    if (lane) {  // whatever this context holds of that kernel is dropped; the next use starts from the note
      jit_unload(c->lane), jit_unload(c->lane_call);
      c->lane.path.clear(), c->lane_call.path.clear();
      c->lane_variant = c->lane_call_variant = -1;
      c->lane_reads_rows = -1;
    } else {
      jit_unload(c->elim);
      c->elim.path.clear();
      c->elim_variant = -1;
      if (c->engine == FAMSEQ_ENGINE_ELIM) return load_elim(c);
    }
    return 0;
  }
  else if (k == "prebuild_lane" || k == "prebuild_elim") {  // compile one given variant into the cache (what the tuner will race): no load
    const bool lane = k == "prebuild_lane";
    if (value < 0 || value >= (lane ? kEnumVariants : kElimVariants)) return fail(c, FAMSEQ_E_ARG, k + " takes a variant index");
    try {
      (void)jit_compile(lane ? enumgen_source(c->model, (int)value, 0) : elim_source(c->model, (int)value));
    } catch (const std::exception &e) {
      return fail(c, FAMSEQ_E_HIP, e.what());
    }
    return 0;
  }
  else if (k == "call_kernels") {  // build (and on a device ctx load) the fused call-path forms now rather than on first use
    if (value != 1) return fail(c, FAMSEQ_E_ARG, "call_kernels takes 1");
    if (!load_call_kernel(c, false)) return fail(c, FAMSEQ_E_HIP, "call-path kernel unavailable: " + c->call_error[0]);
    if (elim_supported(c->model, nullptr) && !load_call_kernel(c, true))
      return fail(c, FAMSEQ_E_HIP, "call-path kernel unavailable: " + c->call_error[1]);
    return 0;
  }
  else if (k == "group_digits") {
    if (value < -1 || value > enumgen_max_group_digits(c->model))
      return fail(c, FAMSEQ_E_ARG, "group_digits must be -1 (auto) or 0.." + std::to_string(enumgen_max_group_digits(c->model)) +
                                       " (looped members of this pedigree's enumeration)");
    if (value >= 0 && !load_lane(c, (int)value)) return fail(c, FAMSEQ_E_HIP, "lane kernel unavailable: " + c->lane_error);
    c->group_digits = (int)value;
    return 0;
  }
  else if (k == "engine") {
    if (value != FAMSEQ_ENGINE_ENUM && value != FAMSEQ_ENGINE_ELIM) return fail(c, FAMSEQ_E_ARG, "engine must be 0 (enum) or 1 (elim)");
    if (value == FAMSEQ_ENGINE_ELIM) {
      const int rc = load_elim(c);
      if (rc != 0) return rc;
    }
    c->engine = (int)value;
    return 0;
  }
  else if (k == "phase_clock_report") {  // measuring aid: see report_phase_clock
    if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "phase_clock_report needs a device");
    HIP_TRY(c, hipSetDevice(c->device));
    report_phase_clock(c);
    return 0;
  }
  else if (k == "chunk_sites") {
    if (c->device >= 0) HIP_TRY(c, hipSetDevice(c->device));
    c->chunk_sites = value;
    free_slots(c);
    return 0;
  }
  else return fail(c, FAMSEQ_E_ARG, "unknown option " + k);
  c->plan_dirty = true;
  const int rc = refresh_plan(c);
  if (rc != 0) {  // keep the previous, working plan
    c->opt = saved;
    c->plan_dirty = true;
    const std::string why = c->err;
    (void)refresh_plan(c);
    c->err = why;
  }
  return rc;
}

namespace {
std::string json_str(const std::string &v) {  // paths may hold quotes or backslashes
  std::string o;
  for (char ch : v) {
    if (ch == '"' || ch == '\\') o += '\\';
    if ((unsigned char)ch < 0x20) continue;
    o += ch;
  }
  return o;
}
}  // namespace

extern "C" const char *famseq_plan_json(famseq_ctx *c) {
  if (!c) return "{}";
  if (c->big) {
    c->json = "{\"N\":" + std::to_string(c->model.n_members) + ",\"engine\":" + std::to_string(c->engine) + ",\"elim_supported\":1,\"elim_code_object\":\"" +
              json_str(c->elim.path) + "\",\"elim_variant\":" + std::to_string(c->elim_variant) + ",\"elim_blocks_per_cu\":" +
              std::to_string(c->elim_blocks_per_cu) + ",\"elim_conditioned_members\":" + std::to_string(elim_conditioned_members(c->model)) +
              ",\"enum_supported\":0,\"device\":" + std::to_string(c->device) + ",\"cus\":" + std::to_string(c->n_cus) + "}";
    return c->json.c_str();
  }
  c->json = c->plan.json();
  c->json.pop_back();
  c->json += ",\"engine\":" + std::to_string(c->engine) + ",\"elim_supported\":" +
             std::string(elim_supported(c->model, nullptr) ? "1" : "0") + ",\"elim_code_object\":\"" + json_str(c->elim.path) +
             "\",\"enum_lane_shape\":\"" + enumgen_describe(c->model, c->lane_variant) + "\",\"enum_impl\":" + std::to_string(c->enum_impl) + ",\"enum_lane_code_object\":\"" + json_str(c->lane.path) +
             "\",\"enum_lane_failed\":" + std::string(c->lane_failed ? "1" : "0") + ",\"enum_lane_error\":\"" + json_str(c->lane_error.substr(0, 400)) + "\"" + ",\"device\":" + std::to_string(c->device) + ",\"cus\":" + std::to_string(c->n_cus) +
             ",\"blocks_per_cu\":" + std::to_string(c->blocks_per_cu) + ",\"elim_variant\":" + std::to_string(c->elim_variant) +
             ",\"elim_conditioned_members\":" + std::to_string(elim_conditioned_members(c->model)) +
             ",\"elim_blocks_per_cu\":" + std::to_string(c->elim_blocks_per_cu) + ",\"enum_lane_variant\":" +
             std::to_string(c->lane_variant) + ",\"enum_lane_blocks_per_cu\":" + std::to_string(c->lane_blocks_per_cu) +
             ",\"enum_group_digits\":" + std::to_string(c->group_digits) + ",\"enum_group_digits_max\":" +
             std::to_string(enumgen_max_group_digits(c->model)) + ",\"enum_group_digits_last\":" +
             std::to_string(c->last_group_digits) + ",\"enum_group_code_objects\":[";
  for (int d = 1; d <= kEnumMaxGroupDigits; ++d) c->json += std::string(d > 1 ? "," : "") + "\"" + json_str(c->grp[d].path) + "\"";
  c->json += "],\"enum_lane_call_code_object\":\"" + json_str(c->lane_call.path) + "\",\"elim_call_code_object\":\"" +
             json_str(c->elim_call.path) + "\",\"enum_lane_call_error\":\"" + json_str(c->call_error[0].substr(0, 300)) + "\",\"elim_call_error\":\"" +
             json_str(c->call_error[1].substr(0, 300)) + "\",\"enum_lane_call_reads_rows\":" + std::to_string(c->lane_reads_rows) + ",\"tune\":\"" +
             json_str(c->tune_report) + "\"}";
  return c->json.c_str();
}

extern "C" int famseq_bn_batch_device(famseq_ctx *c, int64_t n_sites, const double *d_lk, const uint8_t *d_flags,
                                      double *d_post, double *d_single, uint8_t *d_status, void *stream) {
  if (!c) return FAMSEQ_E_ARG;
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_sites < 0 || (n_sites > 0 && (!d_lk || !d_post))) return fail(c, FAMSEQ_E_ARG, "bad batch arguments");
  if (n_sites == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_engine(c, n_sites, d_lk, d_flags, d_post, d_single, d_status, static_cast<hipStream_t>(stream)));
  return 0;
}

namespace {

struct HostIO {
  const double *lk = nullptr;      // exactly one of lk / pl16
  const uint16_t *pl16 = nullptr;  // [n_sites][n_seq][3]
  const uint8_t *flags = nullptr;
  double *post = nullptr, *single = nullptr;  // raw outputs [n_sites][N][3]
  uint8_t *status = nullptr;
  double *gpp = nullptr, *fpp = nullptr;  // called outputs [n_sites][n_seq][3]
  int8_t *fgt = nullptr;                  // [n_sites][n_seq]
  char *text = nullptr;                   // the same three as printed: [n_sites][n_seq][FAMSEQ_TEXT_STRIDE]
};

// Upload the sequenced-member list (VCF column order) and its inverse when it changes.
int set_sequenced(famseq_ctx *c, const int32_t *seq, int n_seq) {
  if (n_seq < 0 || n_seq > c->model.n_members || (n_seq > 0 && !seq)) return fail(c, FAMSEQ_E_ARG, "bad sequenced-member list");
  std::vector<int32_t> v(seq, seq + n_seq), col(c->model.n_members, -1);
  for (int k = 0; k < n_seq; ++k) {
    if (v[k] < 0 || v[k] >= c->model.n_members || col[v[k]] >= 0) return fail(c, FAMSEQ_E_ARG, "bad sequenced-member list");
    col[v[k]] = k;
  }
  if (c->d_seq && v == c->seq_members) return 0;
  if (!c->d_seq) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_seq), c->model.n_members * sizeof(int32_t)));
  if (!c->d_col) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_col), c->model.n_members * sizeof(int32_t)));
  if (!c->d_slot) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_slot), c->model.n_members * sizeof(int32_t)));
  // where a member's printed values go in the call-path kernels' output rows: its column; members without one fill the slots behind
  std::vector<int32_t> slot(col);
  for (int p = 0, next = n_seq; p < c->model.n_members; ++p)
    if (slot[p] < 0) slot[p] = next++;
  HIP_TRY(c, hipMemcpy(c->d_slot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (n_seq) HIP_TRY(c, hipMemcpy(c->d_seq, v.data(), n_seq * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_col, col.data(), col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  c->seq_members = v;
  return 0;
}

int run_chunks(famseq_ctx *c, int64_t n_sites, const HostIO &io, int n_seq, int64_t chunk);

// Chunked host pipeline shared by every host-buffer entry point: per chunk H2D -> [unpack] ->
// posterior kernel -> [phred/call] -> D2H, each stage on its own stream and chained by events, so
// that chunk k+1 is copied in while chunk k is copied out (two chunk-sized streams running the
// whole sequence each fell into lockstep and used one direction of the link at a time).
int run_host(famseq_ctx *c, int64_t n_sites, const HostIO &io, int n_seq) {
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_sites == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const int N = c->model.n_members;
  const size_t row = size_t(3) * N * sizeof(double);
  const bool called = io.gpp || io.fpp || io.fgt || io.text;
  // default chunk: at most 64 MiB per array, at least four chunks per call so that the stages overlap,
  // but not below the batch size the lane-per-site kernel needs to fill the chip
  int64_t chunk = c->chunk_sites;
  if (chunk <= 0) {
    chunk = std::max<int64_t>(1, (int64_t(64) << 20) / int64_t(row));
    chunk = std::min(chunk, std::max<int64_t>(c->lane_min_sites, (n_sites + 3) / 4));
  }
  chunk = std::min(chunk, n_sites);
  const int want_seq = (io.pl16 || called) ? std::max(n_seq, 1) : 0;
  if (c->slot_sites < chunk || c->slot_seq < want_seq) {
    const int64_t cap = std::max(chunk, c->slot_sites);  // slots only grow: varying batch sizes do not thrash
    const int seqcap = std::max(want_seq, c->slot_seq);
    free_slots(c);
    for (int s = 0; s < famseq_ctx::kSlots; ++s) {
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_lk[s]), cap * row));
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_post[s]), cap * row));
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_single[s]), cap * row));
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_flags[s]), cap));
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_status[s]), cap));
      if (seqcap) {
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_pl[s]), cap * seqcap * 3 * sizeof(uint16_t)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_gpp[s]), cap * seqcap * 3 * sizeof(double)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_fpp[s]), cap * seqcap * 3 * sizeof(double)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_fgt[s]), cap * seqcap));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_text[s]), cap * seqcap * size_t(kTextStride)));
        if (!c->d_call[s]) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_call[s]), sizeof(CallIO)));
      }
    }
    c->slot_sites = cap;
    c->slot_seq = seqcap;
  }
  if (io.pl16 && !c->d_lut) {  // pow(10,-k/10) through the host's libm, as file.cpp:589 computes it
    std::vector<double> lut(kPlLutSize);
    for (int k = 0; k < kPlLutSize; ++k) lut[k] = std::pow(10.0, -std::fabs(double(k)) / 10.0);
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_lut), lut.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_lut, lut.data(), lut.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (called) {
    // The generated kernels' call-path arguments (struct fs_call_args) depend on the slot only, not on the
    // chunk: written once per call, synchronously (nothing of an earlier call is in flight any more), so
    // that no asynchronous copy ever reads host memory that has gone out of scope.
    for (int s = 0; s < famseq_ctx::kSlots; ++s) {
      CallIO cio;
      cio.pl = io.pl16 ? c->d_pl[s] : nullptr;
      cio.lut = c->d_lut;
      cio.col = c->d_col;
      cio.slot = c->d_slot;
      cio.gpp = io.gpp || io.text ? c->d_gpp[s] : nullptr;
      cio.fpp = io.fpp || io.text ? c->d_fpp[s] : nullptr;
      cio.fgt = io.fgt || io.text ? c->d_fgt[s] : nullptr;
      cio.n_seq = n_seq;
      // e / d for e < 2^16, d <= 60 as the high word of e * (2^32 / d + 1): exact (io_kernels.hip)
      cio.magic_w = 0xFFFFFFFFu / uint32_t(3 * n_seq) + 1;
      cio.magic_n = n_seq > 1 ? 0xFFFFFFFFu / uint32_t(n_seq) + 1 : 0;  // one column: the kernel divides by 1 itself
      if (std::getenv("FAMSEQ_PHASE_CLOCK")) {
        if (!c->d_phase) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_phase), kPhases * sizeof(unsigned long long)));
        if (s == 0) HIP_TRY(c, hipMemset(c->d_phase, 0, kPhases * sizeof(unsigned long long)));
        cio.phase_clk = c->d_phase;
      }
      HIP_TRY(c, hipMemcpy(c->d_call[s], &cio, sizeof cio, hipMemcpyHostToDevice));
    }
  }
  // From here on copies into the caller's buffers may be in flight: an error must not return
  // before both streams have drained.
  const int rc = run_chunks(c, n_sites, io, n_seq, chunk);
  for (int s = 0; s < famseq_ctx::kStages; ++s) {
    const hipError_t e = hipStreamSynchronize(c->stream[s]);
    if (e != hipSuccess && rc == 0) return fail(c, FAMSEQ_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  }
  if (called && c->d_phase && rc == 0) {  // measuring aid: where the call-path kernel's waves spend their cycles
    unsigned long long ph[kPhases];
    HIP_TRY(c, hipMemcpy(ph, c->d_phase, sizeof ph, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (int i = 0; i < kPhases; ++i) tot += ph[i];
    std::fprintf(stderr, "famseq phase clock (wave cycles, %lld sites):", (long long)n_sites);
    for (int i = 0; i < kPhases; ++i) std::fprintf(stderr, " [%d] %.1f%%", i, tot ? 100.0 * double(ph[i]) / double(tot) : 0.0);
    std::fprintf(stderr, "  total %llu\n", tot);
  }
  return rc;
}

int run_chunks(famseq_ctx *c, int64_t n_sites, const HostIO &io, int n_seq, int64_t chunk) {
  const int N = c->model.n_members;
  const size_t row = size_t(3) * N * sizeof(double);
  const bool called = io.gpp || io.fpp || io.fgt || io.text;
  hipStream_t s_in = c->stream[0], s_k = c->stream[1], s_out = c->stream[2];
  int k = 0;
  for (int64_t lo = 0; lo < n_sites; lo += chunk, ++k) {
    const int s = k % famseq_ctx::kSlots;
    const int64_t n = std::min(chunk, n_sites - lo);
    // copy in: the slot is free once the chunk that used it last has been copied out
    if (k >= famseq_ctx::kSlots) HIP_TRY(c, hipStreamWaitEvent(s_in, c->ev_out[s], 0));
    if (io.pl16) {
      HIP_TRY(c, hipMemcpyAsync(c->d_pl[s], io.pl16 + lo * n_seq * 3, n * n_seq * 3 * sizeof(uint16_t),
                                hipMemcpyHostToDevice, s_in));
    } else {
      HIP_TRY(c, hipMemcpyAsync(c->d_lk[s], io.lk + lo * 3 * N, n * row, hipMemcpyHostToDevice, s_in));
    }
    if (io.flags) HIP_TRY(c, hipMemcpyAsync(c->d_flags[s], io.flags + lo, n, hipMemcpyHostToDevice, s_in));
    HIP_TRY(c, hipEventRecord(c->ev_in[s], s_in));
    // compute
    HIP_TRY(c, hipStreamWaitEvent(s_k, c->ev_in[s], 0));
    const bool need_single = io.single || io.gpp || io.text;
    const bool need_status = io.status || called;
    bool fused = false;
    if (called && !io.post && !io.single) {
      hipError_t e = hipSuccess;
      fused = launch_engine_fused(c, n, io.pl16 ? nullptr : c->d_lk[s], io.flags ? c->d_flags[s] : nullptr,
                                  need_status ? c->d_status[s] : nullptr, io.pl16 != nullptr, c->d_call[s], s_k, &e);
      if (fused) HIP_TRY(c, e);
    }
    if (!fused) {
      if (io.pl16) HIP_TRY(c, launch_unpack_pl16(c->d_pl[s], c->d_col, c->d_lut, N, n_seq, n, c->d_lk[s], s_k));
      HIP_TRY(c, launch_engine(c, n, c->d_lk[s], io.flags ? c->d_flags[s] : nullptr, c->d_post[s],
                               need_single ? c->d_single[s] : nullptr, need_status ? c->d_status[s] : nullptr, s_k));
      if (called)
        HIP_TRY(c, launch_phred_call(c->d_post[s], c->d_single[s], c->d_status[s], c->d_seq, N, n_seq, n, c->d_gpp[s],
                                     c->d_fpp[s], c->d_fgt[s], s_k));
    }
    // the called outputs as text, while they are in HBM anyway: one record per (site, sample) pair
    if (io.text) HIP_TRY(c, launch_text_call(c->d_gpp[s], c->d_fpp[s], c->d_fgt[s], n * n_seq, c->d_text[s], s_k));
    HIP_TRY(c, hipEventRecord(c->ev_done[s], s_k));
    // copy out
    HIP_TRY(c, hipStreamWaitEvent(s_out, c->ev_done[s], 0));
    if (called) {
      const size_t cr = size_t(3) * n_seq * sizeof(double);
      if (io.gpp) HIP_TRY(c, hipMemcpyAsync(io.gpp + lo * 3 * n_seq, c->d_gpp[s], n * cr, hipMemcpyDeviceToHost, s_out));
      if (io.fpp) HIP_TRY(c, hipMemcpyAsync(io.fpp + lo * 3 * n_seq, c->d_fpp[s], n * cr, hipMemcpyDeviceToHost, s_out));
      if (io.fgt) HIP_TRY(c, hipMemcpyAsync(io.fgt + lo * n_seq, c->d_fgt[s], n * n_seq, hipMemcpyDeviceToHost, s_out));
      if (io.text)
        HIP_TRY(c, hipMemcpyAsync(io.text + lo * n_seq * kTextStride, c->d_text[s], size_t(n) * n_seq * kTextStride, hipMemcpyDeviceToHost, s_out));
    }
    if (io.post) HIP_TRY(c, hipMemcpyAsync(io.post + lo * 3 * N, c->d_post[s], n * row, hipMemcpyDeviceToHost, s_out));
    if (io.single) HIP_TRY(c, hipMemcpyAsync(io.single + lo * 3 * N, c->d_single[s], n * row, hipMemcpyDeviceToHost, s_out));
    if (io.status) HIP_TRY(c, hipMemcpyAsync(io.status + lo, c->d_status[s], n, hipMemcpyDeviceToHost, s_out));
    HIP_TRY(c, hipEventRecord(c->ev_out[s], s_out));
  }
  return 0;
}

}  // namespace

extern "C" int famseq_bn_batch(famseq_ctx *c, int64_t n_sites, const double *lk, const uint8_t *flags, double *post,
                               double *post_single, uint8_t *status) {
  if (!c) return FAMSEQ_E_ARG;
  if (n_sites < 0 || (n_sites > 0 && (!lk || !post))) return fail(c, FAMSEQ_E_ARG, "bad batch arguments");
  HostIO io;
  io.lk = lk; io.flags = flags; io.post = post; io.single = post_single; io.status = status;
  return run_host(c, n_sites, io, 0);
}

extern "C" int famseq_bn_batch_sharded(famseq_ctx *const *ctxs, int n_ctx, int64_t n_sites, const double *lk,
                                       const uint8_t *flags, double *post, double *post_single, uint8_t *status) {
  if (!ctxs || n_ctx < 1) return FAMSEQ_E_ARG;
  for (int g = 0; g < n_ctx; ++g)
    if (!ctxs[g] || ctxs[g]->model.n_members != ctxs[0]->model.n_members) return FAMSEQ_E_ARG;
  if (n_sites < 0 || (n_sites > 0 && (!lk || !post))) return fail(ctxs[0], FAMSEQ_E_ARG, "bad batch arguments");
  const int64_t w = int64_t(3) * ctxs[0]->model.n_members;
  std::vector<int> rc(n_ctx, 0);
  std::vector<std::thread> pool;
  for (int g = 0; g < n_ctx; ++g)
    pool.emplace_back([&, g] {
      const int64_t lo = n_sites * g / n_ctx, hi = n_sites * (g + 1) / n_ctx;
      rc[g] = famseq_bn_batch(ctxs[g], hi - lo, lk + lo * w, flags ? flags + lo : nullptr, post + lo * w,
                              post_single ? post_single + lo * w : nullptr, status ? status + lo : nullptr);
    });
  for (std::thread &t : pool) t.join();
  for (int g = 0; g < n_ctx; ++g)
    if (rc[g] != 0) return rc[g];
  return 0;
}

extern "C" int famseq_bn_batch_device_sharded(famseq_ctx *const *ctxs, int n_ctx, const int64_t *n_sites,
                                              const double *const *d_lk, const uint8_t *const *d_flags,
                                              double *const *d_post, double *const *d_single, uint8_t *const *d_status) {
  if (!ctxs || n_ctx < 1 || !n_sites || !d_lk || !d_post) return FAMSEQ_E_ARG;
  for (int g = 0; g < n_ctx; ++g) {
    if (!ctxs[g] || ctxs[g]->model.n_members != ctxs[0]->model.n_members) return FAMSEQ_E_ARG;
    if (ctxs[g]->device < 0) return fail(ctxs[g], FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
    if (n_sites[g] < 0 || (n_sites[g] > 0 && (!d_lk[g] || !d_post[g]))) return fail(ctxs[g], FAMSEQ_E_ARG, "bad batch arguments");
  }
  std::vector<int> rc(n_ctx, 0);
  std::vector<std::thread> pool;
  for (int g = 0; g < n_ctx; ++g)
    pool.emplace_back([&, g] {
      famseq_ctx *c = ctxs[g];
      rc[g] = [&]() -> int {
        if (n_sites[g] == 0) return 0;
        HIP_TRY(c, hipSetDevice(c->device));  // the device binding is per host thread
        HIP_TRY(c, launch_engine(c, n_sites[g], d_lk[g], d_flags ? d_flags[g] : nullptr, d_post[g],
                                 d_single ? d_single[g] : nullptr, d_status ? d_status[g] : nullptr, c->stream[1]));
        HIP_TRY(c, hipStreamSynchronize(c->stream[1]));
        return 0;
      }();
    });
  for (std::thread &t : pool) t.join();
  for (int g = 0; g < n_ctx; ++g)
    if (rc[g] != 0) return rc[g];
  return 0;
}

extern "C" int famseq_stream_probe(famseq_ctx *c, int64_t n_doubles, const double *d_in, double *d_out1, double *d_out2,
                                   void *stream) {
  if (!c) return FAMSEQ_E_ARG;
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_doubles < 0 || (n_doubles > 0 && (!d_in || !d_out1 || !d_out2)) || ((uintptr_t)d_in | (uintptr_t)d_out1 | (uintptr_t)d_out2) & 15)
    return fail(c, FAMSEQ_E_ARG, "bad probe arguments (arrays must be 16-byte aligned)");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_stream_probe(d_in, d_out1, d_out2, n_doubles, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" void *famseq_alloc_pinned(size_t bytes) {
  void *p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

extern "C" void famseq_free_pinned(void *p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int famseq_bn_call_batch(famseq_ctx *c, int64_t n_sites, const double *lk, const uint16_t *pl16,
                                    const uint8_t *flags, const int32_t *seq_members, int32_t n_seq, double *gpp,
                                    double *fpp, int8_t *fgt, uint8_t *status) {
  if (!c) return FAMSEQ_E_ARG;
  if (n_sites < 0 || (n_sites > 0 && ((lk == nullptr) == (pl16 == nullptr))))
    return fail(c, FAMSEQ_E_ARG, "exactly one of lk / pl16 must be given");
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_seq < 1) return fail(c, FAMSEQ_E_ARG, "n_seq must be >= 1");
  HIP_TRY(c, hipSetDevice(c->device));
  const int rc = set_sequenced(c, seq_members, n_seq);
  if (rc != 0) return rc;
  HostIO io;
  io.lk = lk; io.pl16 = pl16; io.flags = flags; io.gpp = gpp; io.fpp = fpp; io.fgt = fgt; io.status = status;
  return run_host(c, n_sites, io, n_seq);
}

static_assert(kTextStride == FAMSEQ_TEXT_STRIDE, "the header's record size is the text kernel's");

extern "C" int famseq_bn_call_text_batch(famseq_ctx *c, int64_t n_sites, const double *lk, const uint16_t *pl16,
                                         const uint8_t *flags, const int32_t *seq_members, int32_t n_seq, char *text,
                                         uint8_t *status) {
  if (!c) return FAMSEQ_E_ARG;
  if (n_sites < 0 || (n_sites > 0 && ((lk == nullptr) == (pl16 == nullptr))))
    return fail(c, FAMSEQ_E_ARG, "exactly one of lk / pl16 must be given");
  if (n_sites > 0 && !text) return fail(c, FAMSEQ_E_ARG, "text must be given");
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_seq < 1) return fail(c, FAMSEQ_E_ARG, "n_seq must be >= 1");
  HIP_TRY(c, hipSetDevice(c->device));
  const int rc = set_sequenced(c, seq_members, n_seq);
  if (rc != 0) return rc;
  HostIO io;
  io.lk = lk; io.pl16 = pl16; io.flags = flags; io.text = text; io.status = status;
  return run_host(c, n_sites, io, n_seq);
}

extern "C" int famseq_format_probe(famseq_ctx *c, int64_t n, const double *values, char *out) {
  if (!c) return FAMSEQ_E_ARG;
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n < 0 || (n > 0 && (!values || !out))) return fail(c, FAMSEQ_E_ARG, "bad probe arguments");
  if (n == 0) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  double *d_in = nullptr;
  char *d_out = nullptr;
  HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&d_in), size_t(n) * sizeof(double)));
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), size_t(n) * 16);
  if (e == hipSuccess) e = hipMemcpy(d_in, values, size_t(n) * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_g6_probe(d_in, n, d_out, c->stream[1]);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream[1]);
  if (e == hipSuccess) e = hipMemcpy(out, d_out, size_t(n) * 16, hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, FAMSEQ_E_HIP, std::string("famseq_format_probe: ") + hipGetErrorString(e));
  return 0;
}

// The call path on buffers that are resident already (a pipeline that keeps its packed PLs, or its results, in HBM; bench.py's
// `call_path` line): the same kernels as famseq_bn_call_batch, enqueued on the caller's stream, nothing copied.
extern "C" int famseq_bn_call_batch_device(famseq_ctx *c, int64_t n_sites, const double *d_lk, const uint16_t *d_pl16,
                                           const uint8_t *d_flags, const int32_t *seq_members, int32_t n_seq, double *d_gpp,
                                           double *d_fpp, int8_t *d_fgt, uint8_t *d_status, char *d_text, void *stream_) {
  if (!c) return FAMSEQ_E_ARG;
  if (n_sites < 0 || (n_sites > 0 && ((d_lk == nullptr) == (d_pl16 == nullptr))))
    return fail(c, FAMSEQ_E_ARG, "exactly one of d_lk / d_pl16 must be given");
  if (c->device < 0) return fail(c, FAMSEQ_E_NODEVICE, "context was created without a device; there is no CPU path");
  if (n_seq < 1) return fail(c, FAMSEQ_E_ARG, "n_seq must be >= 1");
  if (d_text && (reinterpret_cast<uintptr_t>(d_text) & 15)) return fail(c, FAMSEQ_E_ARG, "d_text must be 16-byte aligned");
  HIP_TRY(c, hipSetDevice(c->device));
  {
    const int rc = set_sequenced(c, seq_members, n_seq);
    if (rc != 0) return rc;
  }
  if (n_sites == 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int N = c->model.n_members;
  const bool want_text = d_text != nullptr;
  // what the stages write: the caller's arrays, or scratch of this context's for those the caller does not ask for but a later
  // stage reads (the text kernel reads all three) or a separate-stages batch passes through (fp64 rows in and out).  Scratch is
  // allocated for what this call needs only: a fused batch with every output given needs none.
  const size_t row = size_t(3) * N * sizeof(double);
  const bool will_fuse = call_fuses(c, n_sites, d_pl16 != nullptr);
  const bool need_rows = !will_fuse, need_called = !will_fuse || (want_text && !(d_gpp && d_fpp && d_fgt)), need_status = !d_status;
  if ((need_rows && (!c->dev_tmp[0] || c->dev_tmp_sites < n_sites)) || (need_called && (!c->dev_tmp[3] || c->dev_tmp_sites < n_sites || c->dev_tmp_seq < n_seq)) ||
      (need_status && (!c->dev_tmp_status || c->dev_tmp_sites < n_sites))) {
    HIP_TRY(c, hipStreamSynchronize(stream));  // nothing of an earlier call may still use what is freed here
    const int64_t cap = std::max(n_sites, c->dev_tmp_sites);
    const int seqcap = std::max<int>(n_seq, c->dev_tmp_seq);
    const bool had_rows = c->dev_tmp[0] != nullptr, had_called = c->dev_tmp[3] != nullptr;
    for (double *&q : c->dev_tmp) {
      if (q) (void)hipFree(q);
      q = nullptr;
    }
    if (c->dev_tmp_fgt) (void)hipFree(c->dev_tmp_fgt);
    if (c->dev_tmp_status) (void)hipFree(c->dev_tmp_status);
    c->dev_tmp_fgt = nullptr, c->dev_tmp_status = nullptr, c->dev_tmp_sites = 0;
    if (need_rows || had_rows)
      for (int i = 0; i < 3; ++i) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->dev_tmp[i]), size_t(cap) * row));
    if (need_called || had_called) {
      for (int i = 3; i < 5; ++i) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->dev_tmp[i]), size_t(cap) * 3 * seqcap * sizeof(double)));
      HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->dev_tmp_fgt), size_t(cap) * seqcap));
    }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->dev_tmp_status), size_t(cap)));
    c->dev_tmp_sites = cap, c->dev_tmp_seq = seqcap;
  }
  double *gpp = d_gpp ? d_gpp : (want_text ? c->dev_tmp[3] : nullptr), *fpp = d_fpp ? d_fpp : (want_text ? c->dev_tmp[4] : nullptr);
  int8_t *fgt = d_fgt ? d_fgt : (want_text ? c->dev_tmp_fgt : nullptr);
  uint8_t *status = d_status ? d_status : c->dev_tmp_status;
  if (d_pl16 && !c->d_lut) {  // pow(10,-k/10) through the host's libm, as file.cpp:589 computes it
    std::vector<double> lut(kPlLutSize);
    for (int k = 0; k < kPlLutSize; ++k) lut[k] = std::pow(10.0, -std::fabs(double(k)) / 10.0);
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_lut), lut.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_lut, lut.data(), lut.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  CallIO cio;
  cio.pl = d_pl16;
  cio.lut = c->d_lut;
  cio.col = c->d_col;
  cio.slot = c->d_slot;
  cio.gpp = gpp, cio.fpp = fpp, cio.fgt = fgt;
  cio.n_seq = n_seq;
  cio.magic_w = 0xFFFFFFFFu / uint32_t(3 * n_seq) + 1;
  cio.magic_n = n_seq > 1 ? 0xFFFFFFFFu / uint32_t(n_seq) + 1 : 0;
  if (!c->d_call_dev) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_call_dev), sizeof(CallIO)));
  if (!c->call_dev_valid || std::memcmp(&cio, &c->call_dev_host, sizeof cio) != 0) {
    // the argument block changes only when the caller's pointers do: written synchronously, after whatever of this stream
    // may still read the old one
    HIP_TRY(c, hipStreamSynchronize(stream));
    HIP_TRY(c, hipMemcpy(c->d_call_dev, &cio, sizeof cio, hipMemcpyHostToDevice));
    c->call_dev_host = cio, c->call_dev_valid = true;
  }
  hipError_t e = hipSuccess;
  const bool fused = launch_engine_fused(c, n_sites, d_lk, d_flags, status, d_pl16 != nullptr, c->d_call_dev, stream, &e);
  if (fused) HIP_TRY(c, e);
  if (!fused) {
    if (!c->dev_tmp[0] || !c->dev_tmp[3]) return fail(c, FAMSEQ_E_HIP, "call path: the batch left the fused kernel without scratch rows (internal)");
    const double *lk = d_lk;
    if (d_pl16) {
      HIP_TRY(c, launch_unpack_pl16(d_pl16, c->d_col, c->d_lut, N, n_seq, n_sites, c->dev_tmp[0], stream));
      lk = c->dev_tmp[0];
    }
    HIP_TRY(c, launch_engine(c, n_sites, lk, d_flags, c->dev_tmp[1], c->dev_tmp[2], status, stream));
    HIP_TRY(c, launch_phred_call(c->dev_tmp[1], c->dev_tmp[2], status, c->d_seq, N, n_seq, n_sites, gpp ? gpp : c->dev_tmp[3],
                                 fpp ? fpp : c->dev_tmp[4], fgt ? fgt : c->dev_tmp_fgt, stream));
  }
  if (want_text)
    HIP_TRY(c, launch_text_call(gpp ? gpp : c->dev_tmp[3], fpp ? fpp : c->dev_tmp[4], fgt ? fgt : c->dev_tmp_fgt, n_sites * n_seq, d_text, stream));
  return 0;
}
