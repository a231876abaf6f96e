// jit.cpp — see jit.h
#include "jit.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
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

bool exists(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}

bool writable_dir(const std::string &d) {
  ::mkdir(d.c_str(), 0755);
  return ::access(d.c_str(), W_OK | X_OK) == 0;
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
  if (writable_dir(in_tree)) return in_tree;
  const std::string tmp = "/tmp/famseq_kernels_" + std::to_string((long)getuid());
  if (writable_dir(tmp)) return tmp;
  throw std::runtime_error("no writable kernel cache directory");
}

const char kCompilerTag[] = "hipcc gfx950 -O3 -ffp-contract=off v1";

}  // namespace

std::string jit_compile(const std::string &source) {
  char name[40];
  std::snprintf(name, sizeof name, "%016llx", (unsigned long long)fnv1a(source + kCompilerTag));
  // a prebuilt object next to the library wins even when that directory is read-only
  const std::string shipped = lib_dir() + "/kernels/" + name + ".hsaco";
  if (exists(shipped)) return shipped;
  const std::string dir = cache_dir();
  const std::string obj = dir + "/" + name + ".hsaco";
  if (exists(obj)) return obj;
  const std::string src = dir + "/" + name + "." + std::to_string((long)getpid()) + ".hip";
  const std::string tmp = obj + "." + std::to_string((long)getpid()) + ".tmp";
  {
    std::ofstream f(src.c_str());
    f << source;
    if (!f) throw std::runtime_error("cannot write " + src);
  }
  const char *cc = std::getenv("FAMSEQ_HIPCC");
  const std::string cmd = std::string(cc ? cc : "/opt/rocm/bin/hipcc") +
                          " --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --genco -o '" + tmp + "' '" + src +
                          "' > '" + src + ".log' 2>&1";
  const int rc = std::system(cmd.c_str());
  if (rc != 0 || !exists(tmp)) {
    std::string log;
    std::ifstream lf((src + ".log").c_str());
    std::getline(lf, log, '\0');
    throw std::runtime_error("kernel compilation failed (" + cmd + "): " + log.substr(0, 2000));
  }
  ::rename(tmp.c_str(), obj.c_str());  // atomic publish: concurrent ranks may compile the same kernel
  if (std::getenv("FAMSEQ_KEEP_SRC")) {  // debugging aid: keep the generated source next to the object
    ::rename(src.c_str(), (dir + "/" + name + ".hip").c_str());
  } else {
    ::unlink(src.c_str());
  }
  ::unlink((src + ".log").c_str());
  return obj;
}

JitKernel jit_load(const std::string &source, const std::string &entry) {
  JitKernel k;
  k.path = jit_compile(source);
  hipError_t e = hipModuleLoad(&k.module, k.path.c_str());
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
