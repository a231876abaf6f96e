// jit.cpp — see jit.h
#include "jit.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace famseq {

namespace {

uint64_t fnv1a(const std::string &s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}

// a code object that is there AND has content (a zero-length file — an interrupted writer, a test placeholder —
// is no cache hit)
bool exists(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && st.st_size > 0;
}

bool writable_dir(const std::string &d) {
  ::mkdir(d.c_str(), 0755);
  return ::access(d.c_str(), W_OK | X_OK) == 0;
}

// The per-user fallback under /tmp is shared ground: code objects are loaded from it without
// recompiling, so it must be a real directory (not a link) of this user's that nobody else can
// write to — otherwise another local user could plant <hash>.hsaco there.
bool private_dir(const std::string &d) {
  ::mkdir(d.c_str(), 0700);
  struct stat st;
  if (::lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  if (st.st_uid != ::getuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;
  return ::access(d.c_str(), W_OK | X_OK) == 0;
}

// A profiler that preloads itself into this process (rocprofv3, roctracer) is inherited by every
// child: spawning sh -> hipcc -> clang from here would exec GPU-initialised processes, which the
// GPU pool forbids and which distorts the profile anyway.
const char *profiler_env() {
  for (const char *k : {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"})
    if (const char *v = std::getenv(k))
      if (*v) return k;
  if (const char *v = std::getenv("LD_PRELOAD"))
    if (std::strstr(v, "rocprof") || std::strstr(v, "roctracer")) return "LD_PRELOAD";
  return nullptr;
}

std::string lib_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<void *>(&jit_compile), &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t k = p.rfind('/');
    if (k != std::string::npos) return p.substr(0, k);
  }
  return ".";
}

std::string cache_dir() {
  if (const char *e = std::getenv("FAMSEQ_KERNEL_CACHE"))
    if (writable_dir(e)) return e;
  const std::string in_tree = lib_dir() + "/kernels";
  // ($FAMSEQ_LIBDIR_READONLY: treat the library's directory as read-only — what a packaged installation is; a test aid,
  // because root, who runs the CPU suite, can write everywhere)
  if (!std::getenv("FAMSEQ_LIBDIR_READONLY") && writable_dir(in_tree)) return in_tree;
  const std::string tmp = "/tmp/famseq_kernels_" + std::to_string((long)getuid());
  if (private_dir(tmp)) return tmp;
  throw std::runtime_error("no usable kernel cache directory (" + in_tree + " is not writable and " + tmp +
                           " is not a private directory of this user)");
}

const char kCompilerTag[] = "hipcc gfx950 -O3 -ffp-contract=off v2";

}  // namespace

namespace {

// "ScratchSize [bytes/lane]: N" of hipcc's -Rpass-analysis=kernel-resource-usage remarks
int parse_scratch(const std::string &log) {
  const char key[] = "ScratchSize [bytes/lane]: ";
  const size_t k = log.find(key);
  if (k == std::string::npos) return -1;
  return std::atoi(log.c_str() + k + sizeof key - 1);
}

int read_res(const std::string &obj) {
  std::ifstream f((obj.substr(0, obj.size() - 6) + ".res").c_str());  // <hash>.hsaco -> <hash>.res
  int v = -1;
  if (f >> v) return v;
  return -1;
}

}  // namespace

namespace {

// The in-process compiler: libhiprtc, which ships with the HIP runtime itself.  No hipcc on the host, no child
// process (so it also works where a process that has initialised the GPU must not exec: under rocprofv3), and the
// same code: for the ten-member enumeration kernel hiprtc and `hipcc --genco` emit identical instruction streams
// (4,835 instructions compared).  Loaded on first use with dlopen, so a host without it still has the cache and hipcc.
struct Rtc {
  typedef struct _hiprtcProgram *Program;
  int (*create)(Program *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*compile)(Program, int, const char **) = nullptr;
  int (*log_size)(Program, size_t *) = nullptr;
  int (*log)(Program, char *) = nullptr;
  int (*code_size)(Program, size_t *) = nullptr;
  int (*code)(Program, char *) = nullptr;
  int (*destroy)(Program *) = nullptr;
  bool ok = false;
};

const Rtc &rtc() {
  static const Rtc r = [] {
    Rtc x;
    void *h = nullptr;
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return x;
    auto sym = [&](const char *n) { return dlsym(h, n); };
    x.create = reinterpret_cast<decltype(x.create)>(sym("hiprtcCreateProgram"));
    x.compile = reinterpret_cast<decltype(x.compile)>(sym("hiprtcCompileProgram"));
    x.log_size = reinterpret_cast<decltype(x.log_size)>(sym("hiprtcGetProgramLogSize"));
    x.log = reinterpret_cast<decltype(x.log)>(sym("hiprtcGetProgramLog"));
    x.code_size = reinterpret_cast<decltype(x.code_size)>(sym("hiprtcGetCodeSize"));
    x.code = reinterpret_cast<decltype(x.code)>(sym("hiprtcGetCode"));
    x.destroy = reinterpret_cast<decltype(x.destroy)>(sym("hiprtcDestroyProgram"));
    x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
    return x;
  }();
  return r;
}

// true: `code` holds the code object and `log` the compiler's remarks.  false with `log` empty: no in-process
// compiler here (the caller may try hipcc); false with `log` set: the compiler rejected the source.
bool rtc_compile(const std::string &source, std::string &code, std::string &log) {
  code.clear(), log.clear();
  if (std::getenv("FAMSEQ_NO_HIPRTC")) return false;  // test aid: the hipcc route
  const Rtc &r = rtc();
  if (!r.ok) return false;
  Rtc::Program prog = nullptr;
  if (r.create(&prog, source.c_str(), "famseq_kernel.hip", 0, nullptr, nullptr) != 0) return false;
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"};
  const int rc = r.compile(prog, 5, opts);
  size_t n = 0;
  if (r.log_size(prog, &n) == 0 && n > 1) {
    log.resize(n);
    (void)r.log(prog, &log[0]);
  }
  bool good = rc == 0 && r.code_size(prog, &n) == 0 && n > 0;
  if (good) {
    code.resize(n);
    good = r.code(prog, &code[0]) == 0;
  }
  (void)r.destroy(&prog);
  if (!good && log.empty()) log = "hiprtcCompileProgram failed (" + std::to_string(rc) + ")";
  return good;
}

}  // namespace

std::string jit_compile(const std::string &source, int *scratch_bytes, bool note_suffices) {
  // One compilation at a time per process: famseq_bn_batch*_sharded runs a host thread per ctx, and
  // with a cold cache every one of them arrives here with the same source.  The first compiles,
  // the others find the object.  (Scratch names are unique per call as well, so that processes
  // sharing a cache directory — one rank per GPU — never touch each other's files.)
  static std::mutex mu;
  static std::atomic<unsigned> serial{0};
  std::lock_guard<std::mutex> lock(mu);
  char name[40];
  std::snprintf(name, sizeof name, "%016llx", (unsigned long long)fnv1a(source + kCompilerTag));
  // a prebuilt object next to the library wins even when that directory is read-only
  // (an explicit $FAMSEQ_KERNEL_CACHE is taken at its word: only that directory is consulted)
  const std::string shipped = lib_dir() + "/kernels/" + name + ".hsaco";
  if (!std::getenv("FAMSEQ_KERNEL_CACHE") && exists(shipped)) {
    if (scratch_bytes) *scratch_bytes = read_res(shipped);
    return shipped;
  }
  const std::string dir = cache_dir();
  const std::string obj = dir + "/" + name + ".hsaco";
  if (exists(obj)) {
    if (scratch_bytes) *scratch_bytes = read_res(obj);
    return obj;
  }
  if (note_suffices && scratch_bytes) {  // a variant that was tried before and lost: its note answers the question
    for (const std::string &o : {std::getenv("FAMSEQ_KERNEL_CACHE") ? obj : shipped, obj}) {  // (an explicit cache: that directory only)
      const int v = read_res(o);
      if (v >= 0) {
        *scratch_bytes = v;
        return "";
      }
    }
  }
  if (std::getenv("FAMSEQ_JIT_SOURCE_ONLY") && std::getenv("FAMSEQ_KERNEL_CACHE")) {
    // test aid (tests/test_generated_host.py compiles the generated SOURCE for the host), honoured only together with
    // an explicit scratch cache: keep the source and a resource note, do not run the compiler.  No code object is
    // written (not even an empty one: nothing a later process could take for a cache hit); the path returned names
    // where it would be, and loading it fails loudly, so this cannot turn into a silent product path.
    std::ofstream((dir + "/" + name + ".hip").c_str()) << source;
    std::ofstream((dir + "/" + name + ".res").c_str()) << 0 << "\n";
    if (scratch_bytes) *scratch_bytes = 0;
    return obj;
  }
  const std::string uniq = std::to_string((long)getpid()) + "_" + std::to_string(serial.fetch_add(1));
  const std::string tmp = obj + "." + uniq + ".tmp";
  auto publish = [&](int scratch) {  // resource note first, then the object: whoever sees the object also finds the note
    const std::string res_tmp = tmp + ".res";
    std::ofstream rf(res_tmp.c_str());
    rf << scratch << "\n";
    rf.close();
    ::rename(res_tmp.c_str(), (dir + "/" + name + ".res").c_str());
    if (scratch_bytes) *scratch_bytes = scratch;
    ::rename(tmp.c_str(), obj.c_str());  // atomic publish: concurrent ranks may compile the same kernel
    if (std::getenv("FAMSEQ_KEEP_SRC")) std::ofstream((dir + "/" + name + ".hip").c_str()) << source;  // debugging aid
  };
  {
    std::string code, log;
    if (rtc_compile(source, code, log)) {
      std::ofstream f(tmp.c_str(), std::ios::binary);
      f.write(code.data(), (std::streamsize)code.size());
      f.close();
      if (!f) throw std::runtime_error("cannot write " + tmp);
      publish(parse_scratch(log));
      return obj;
    }
    if (!log.empty()) throw std::runtime_error("kernel compilation failed (hiprtc): " + log.substr(0, 2000));
  }
  // no in-process compiler on this host: hipcc as a child process
  if (const char *why = profiler_env())
    throw std::runtime_error(std::string("kernel ") + name + " is not in the cache (" + dir + ") and a profiler is attached ($" +
                             why + "), this host has no libhiprtc to compile it in-process: build it outside the profiler first "
                             "(a plan-only ctx with the same options, or __graft_entry__.build())");
  const std::string src = dir + "/" + name + "." + uniq + ".hip";
  {
    std::ofstream f(src.c_str());
    f << source;
    if (!f) throw std::runtime_error("cannot write " + src);
  }
  const char *cc = std::getenv("FAMSEQ_HIPCC");
  // the command line goes through /bin/sh with the paths in single quotes
  if (dir.find('\'') != std::string::npos || (cc && std::string(cc).find_first_of("'\";|&$`\n") != std::string::npos)) {
    ::unlink(src.c_str());
    throw std::runtime_error("kernel cache directory or FAMSEQ_HIPCC holds a shell metacharacter");
  }
  const std::string cmd = std::string(cc ? cc : "/opt/rocm/bin/hipcc") +
                          " --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --genco"
                          " -Rpass-analysis=kernel-resource-usage -o '" + tmp + "' '" + src +
                          "' > '" + src + ".log' 2>&1";
  const int rc = std::system(cmd.c_str());
  std::string log;
  {
    std::ifstream lf((src + ".log").c_str());
    std::getline(lf, log, '\0');
  }
  if (rc != 0 || !exists(tmp)) {
    throw std::runtime_error("kernel compilation failed (" + cmd + "): " + log.substr(0, 2000));
  }
  publish(parse_scratch(log));
  ::unlink(src.c_str());
  ::unlink((src + ".log").c_str());
  return obj;
}

bool jit_cached(const std::string &source) {
  char name[40];
  std::snprintf(name, sizeof name, "%016llx", (unsigned long long)fnv1a(source + kCompilerTag));
  if (!std::getenv("FAMSEQ_KERNEL_CACHE") && exists(lib_dir() + "/kernels/" + name + ".hsaco")) return true;
  try {
    return exists(cache_dir() + "/" + name + ".hsaco");
  } catch (const std::exception &) {
    return false;
  }
}

// A couple of spilled registers cost less than the next variant's lost overlap (measured: 12 B of
// scratch on the fence-free 5-member sum-product kernel, still 7 % faster than the fenced one).
constexpr int kSpillTolerance = 16;  // bytes per lane

std::string jit_pick_variant(const std::function<std::string(int)> &generate, int n_variants, int *picked, int first) {
  std::string best;
  int best_scratch = -1, best_i = 0;
  first = std::max(0, std::min(first, n_variants - 1));
  if (const char *e = std::getenv("FAMSEQ_VARIANT_MIN")) first = std::max(0, std::min(std::atoi(e), n_variants - 1));  // tuning aid
  std::string best_obj;
  for (int v = first; v < n_variants; ++v) {
    std::string src = generate(v);
    int scratch = -1;
    const std::string obj = jit_compile(src, &scratch, /*note_suffices=*/true);
    if (scratch < 0) scratch = 0;  // an object without a note (older cache): take it as it is
    if (best.empty() || scratch < best_scratch) {
      best.swap(src);
      best_obj = obj;
      best_scratch = scratch;
      best_i = v;
    }
    if (best_scratch <= kSpillTolerance) break;
  }
  // (The objects of variants that lost stay where they are: another process or rank sharing the cache may just have
  // been handed one of those paths by jit_compile, and deleting it under that process turned into a failed
  // hipModuleLoad and a context stuck on the fallback kernel.  Their resource notes spare later contests the compilations.)
  if (picked) *picked = best_i;
  return best;
}

namespace {
std::string pick_name(const std::string &key_source) {
  char name[40];
  std::snprintf(name, sizeof name, "%016llx", (unsigned long long)fnv1a(key_source + kCompilerTag));
  return std::string(name) + ".pick";
}
}  // namespace

int jit_read_pick(const std::string &key_source) {
  const std::string name = pick_name(key_source);
  // where jit_write_pick writes first — a local "tune" overrides the picks shipped next to a read-only library —,
  // then the shipped directory (an explicit cache: that directory only)
  std::vector<std::string> dirs;
  try {
    dirs.push_back(cache_dir());
  } catch (const std::exception &) {
  }
  if (!std::getenv("FAMSEQ_KERNEL_CACHE") && (dirs.empty() || dirs[0] != lib_dir() + "/kernels")) dirs.push_back(lib_dir() + "/kernels");
  for (const std::string &d : dirs) {
    std::ifstream f((d + "/" + name).c_str());
    int v = -1;
    if (f >> v) return v;
  }
  return -1;
}

void jit_write_pick(const std::string &key_source, int variant) {
  const std::string path = cache_dir() + "/" + pick_name(key_source);
  const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
  {
    std::ofstream f(tmp.c_str());
    f << variant << "\n";
  }
  ::rename(tmp.c_str(), path.c_str());
}

JitKernel jit_load(const std::string &source, const std::string &entry) {
  JitKernel k;
  k.path = jit_compile(source);
  hipError_t e = hipModuleLoad(&k.module, k.path.c_str());
  if (e != hipSuccess && !std::getenv("FAMSEQ_JIT_SOURCE_ONLY")) {
    // a cached object that does not load (truncated by a crash, removed under us, written by another compiler): drop it
    // and build it once more — in a writable cache; a shipped read-only one is left alone and reported
    (void)hipGetLastError();
    if (::access(k.path.c_str(), F_OK) != 0 || ::unlink(k.path.c_str()) == 0) {
      k.path = jit_compile(source);
      e = hipModuleLoad(&k.module, k.path.c_str());
    }
  }
  if (e != hipSuccess) throw std::runtime_error("hipModuleLoad(" + k.path + "): " + hipGetErrorString(e));
  e = hipModuleGetFunction(&k.fn, k.module, entry.c_str());
  if (e != hipSuccess) {
    (void)hipModuleUnload(k.module);
    throw std::runtime_error("hipModuleGetFunction(" + entry + "): " + hipGetErrorString(e));
  }
  return k;
}

void jit_unload(JitKernel &k) {
  if (k.module) (void)hipModuleUnload(k.module);
  k.module = nullptr;
  k.fn = nullptr;
}

}  // namespace famseq
