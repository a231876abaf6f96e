// jit.h — per-pedigree kernel specialisation: HIP source -> code object -> hipFunction.
//
// Some engines of this library are generated as straight-line HIP for one pedigree (its
// topology fixes every index, so all per-site state lives in registers).  The source is
// compiled for gfx950 — in-process through libhiprtc; by `hipcc --genco` as a child process on a host
// without it — and cached on disk by content hash:
//   $FAMSEQ_KERNEL_CACHE when set; otherwise <dir of libfamseq_hip.so>/kernels/<hash>.hsaco
//   (in-tree: prebuilt objects travel with the library), then /tmp/famseq_kernels_<uid>.
// Fallback compiler: $FAMSEQ_HIPCC or /opt/rocm/bin/hipcc ($FAMSEQ_NO_HIPRTC=1, a test aid, goes straight to it).
#ifndef FAMSEQ_JIT_H_
#define FAMSEQ_JIT_H_

#include <hip/hip_runtime_api.h>

#include <functional>
#include <string>

namespace famseq {

struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  std::string path;  // code object on disk
};

// Compile (or fetch from the cache) and load on the current device.  Throws std::runtime_error.
JitKernel jit_load(const std::string &source, const std::string &entry);
// Compile into the cache without loading (no GPU needed); returns the code-object path.
// *scratch_bytes (optional) receives the kernel's scratch (register spill) bytes per lane as
// hipcc reported them, kept in <hash>.res next to the object; -1 when unknown.
// note_suffices: when the object is not there but its resource note is (a variant that lost an earlier
// jit_pick_variant), return "" with *scratch_bytes set instead of compiling again.
std::string jit_compile(const std::string &source, int *scratch_bytes = nullptr, bool note_suffices = false);
// Is the code object of `source` on disk already (shipped next to the library or in the cache)?  Loading it then takes
// no compilation: what decides whether a tiny batch may be served by a generated kernel.
bool jit_cached(const std::string &source);
// The generators can trade instruction-level parallelism against register pressure
// (`variant` 0 = most parallel).  Compiles variants in order and returns the source of the
// first one that does not spill (more than 16 bytes per lane), or of the one that spills least.  *picked = its index.
std::string jit_pick_variant(const std::function<std::string(int)> &generate, int n_variants, int *picked = nullptr, int first = 0);
// The autotuner's note (famseq_set_option "tune"): which variant of the kernel whose variant-0 source is
// `key_source` ran fastest on this machine — "<hash>.pick" next to the code objects.  -1: none.
int jit_read_pick(const std::string &key_source);
void jit_write_pick(const std::string &key_source, int variant);
void jit_unload(JitKernel &k);

}  // namespace famseq
#endif
