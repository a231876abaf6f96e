// tests/asan_host_check.cpp — the host side of libfamseq_hip.so under AddressSanitizer + UBSan
// (`make asan`; CPU only, never on the GPU box).  Links the library's own host translation units
// (model, plan, both kernel generators, jit, the C ABI), compiled by g++ with
// -fsanitize=address,undefined, and drives them the way the ABI's callers do on plan-only
// contexts: model setup and its rejections, plan options (valid and invalid), every variant of
// both generators, the JSON description, the error paths of the compute entry points without a
// device, and the JIT's failure path (no compiler).  Any sanitizer report aborts with a non-zero
// exit status; the program prints one line per pedigree.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "elim_codegen.h"
#include "enum_codegen.h"
#include "famseq_hip.h"

namespace {

struct Ped {
  const char *name;
  std::vector<int32_t> id, mo, fa, sex;
  std::vector<uint8_t> seq;
};

Ped chain(int n) {  // founder couple, then each generation marries in a founder: n members, depth ~n/2
  Ped p;
  p.name = "chain";
  for (int i = 1; i <= n; ++i) {
    p.id.push_back(i);
    const bool child = i >= 3 && i % 2 == 1;
    p.mo.push_back(child ? (i - 1) : 0);
    p.fa.push_back(child ? (i - 2) : 0);
    p.sex.push_back(i % 2 == 1 ? 1 : 2);
    p.seq.push_back(i % 5 != 0);
  }
  return p;
}

int fails = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
      ++fails;                                                           \
    }                                                                    \
  } while (0)

void drive(const Ped &p) {
  famseq_model m;
  const int n = (int)p.id.size();
  const int rc = famseq_model_init(&m, n, p.id.data(), p.mo.data(), p.fa.data(), p.sex.data(), p.seq.empty() ? nullptr : p.seq.data(),
                                   1e-7, 1.0);
  CHECK(rc == 0);
  if (rc != 0) return;
  char err[256] = "";
  famseq_ctx *c = famseq_create(&m, -1, err, sizeof err);
  CHECK(c != nullptr);
  if (!c) {
    std::fprintf(stderr, "%s: %s\n", p.name, err);
    return;
  }
  size_t bytes = std::strlen(famseq_plan_json(c));
  for (int a = 0; a <= 7; ++a) (void)famseq_set_option(c, "fixed_digits", a);  // 7 is out of range: must be refused cleanly
  for (int l = 0; l <= 6; ++l) (void)famseq_set_option(c, "low_members", l);
  for (int bt : {64, 256, 768, 100, 4096}) (void)famseq_set_option(c, "block_threads", bt);
  CHECK(famseq_set_option(c, "no_such_option", 1) == FAMSEQ_E_ARG);
  CHECK(famseq_set_option(c, "engine", 7) == FAMSEQ_E_ARG);
  bytes += std::strlen(famseq_plan_json(c));
  for (int v = 0; v < famseq::kEnumVariants; ++v) bytes += famseq::enumgen_source(m, v).size();
  bytes += famseq::enumgen_describe(m).size();
  std::string why;
  const bool elim = famseq::elim_supported(m, &why);
  if (elim)
    for (int v = 0; v < famseq::kElimVariants; ++v) bytes += famseq::elim_source(m, v).size();
  // no compiler: the JIT's failure path (message, no leak, the ctx stays usable)
  CHECK(famseq_set_option(c, "enum_impl", 1) != 0);
  if (elim) CHECK(famseq_set_option(c, "engine", FAMSEQ_ENGINE_ELIM) != 0);
  CHECK(std::strlen(famseq_last_error(c)) > 0);
  // compute entry points on a ctx without a device
  std::vector<double> lk(3 * n * 4, 0.25), post(3 * n * 4);
  std::vector<uint8_t> st(4);
  CHECK(famseq_bn_batch(c, 4, lk.data(), nullptr, post.data(), nullptr, st.data()) == FAMSEQ_E_NODEVICE);
  CHECK(famseq_bn_batch_device(c, 4, lk.data(), nullptr, post.data(), nullptr, nullptr, nullptr) == FAMSEQ_E_NODEVICE);
  CHECK(famseq_bn_batch(c, -1, lk.data(), nullptr, post.data(), nullptr, nullptr) == FAMSEQ_E_ARG);
  famseq_ctx *cs[1] = {c};
  const int64_t ns[1] = {4};
  const double *dl[1] = {lk.data()};
  double *dp[1] = {post.data()};
  CHECK(famseq_bn_batch_device_sharded(cs, 1, ns, dl, nullptr, dp, nullptr, nullptr) == FAMSEQ_E_NODEVICE);
  CHECK(famseq_bn_batch_sharded(cs, 1, 4, lk.data(), nullptr, post.data(), nullptr, nullptr) == FAMSEQ_E_NODEVICE);
  std::vector<int8_t> g(n * 4);
  famseq_call_genotypes(lk.data(), n * 4, g.data());
  famseq_destroy(c);
  std::printf("%-8s N=%2d elim=%d  %zu bytes of plan JSON + generated source\n", p.name, n, (int)elim, bytes);
}

}  // namespace

int main() {
  setenv("FAMSEQ_HIPCC", "/bin/false", 1);  // the generators run; nothing is compiled
  setenv("FAMSEQ_NO_HIPRTC", "1", 1);       // (neither in-process nor by hipcc)
  setenv("FAMSEQ_QUIET", "1", 1);
  char tmpl[] = "/tmp/famseq_asan_XXXXXX";
  const char *dir = mkdtemp(tmpl);
  if (!dir) return 2;
  setenv("FAMSEQ_KERNEL_CACHE", dir, 1);

  std::vector<Ped> peds;
  peds.push_back({"single", {1}, {0}, {0}, {1}, {}});
  peds.push_back({"pair", {1, 2}, {0, 0}, {0, 0}, {1, 2}, {}});
  peds.push_back({"trio", {1, 2, 3}, {0, 0, 2}, {0, 0, 1}, {1, 2, 1}, {0, 0, 1}});
  peds.push_back({"ped5", {1, 2, 3, 4, 5}, {0, 0, 2, 2, 2}, {0, 0, 1, 1, 1}, {1, 2, 1, 2, 1}, {}});
  peds.push_back({"ped10", {1, 2, 3, 4, 5, 6, 7, 8, 9, 10}, {0, 0, 2, 2, 0, 0, 5, 5, 4, 4}, {0, 0, 1, 1, 0, 0, 3, 3, 6, 6},
                  {1, 2, 1, 2, 2, 1, 1, 2, 1, 2}, {}});
  peds.push_back({"ped15", {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {0, 0, 2, 2, 2, 0, 0, 0, 6, 6, 4, 4, 8, 8, 8},
                  {0, 0, 1, 1, 1, 0, 0, 0, 3, 3, 7, 7, 5, 5, 5}, {1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1}, {}});
  // first-cousin marriage: a loop, cut by conditioning
  peds.push_back({"cousins", {1, 2, 3, 4, 5, 6, 7, 8, 9}, {0, 0, 2, 2, 0, 0, 5, 4, 8}, {0, 0, 1, 1, 0, 0, 3, 6, 7},
                  {1, 2, 1, 2, 2, 1, 1, 2, 1}, {}});
  Ped c20 = chain(20);
  c20.name = "chain20";
  peds.push_back(c20);
  for (const Ped &p : peds) drive(p);

  // rejections
  famseq_model m;
  const int32_t id3[] = {1, 2, 3}, half_mo[] = {0, 0, 2}, half_fa[] = {0, 0, 0}, sex3[] = {1, 2, 1};
  CHECK(famseq_model_init(&m, 3, id3, half_mo, half_fa, sex3, nullptr, 1e-7, 1.0) == FAMSEQ_E_PED_HALF);
  const int32_t mo3[] = {0, 0, 1}, fa3[] = {0, 0, 2};
  CHECK(famseq_model_init(&m, 3, id3, mo3, fa3, sex3, nullptr, 1e-7, 1.0) == FAMSEQ_E_PED_SEX);
  CHECK(famseq_model_init(&m, 0, id3, mo3, fa3, sex3, nullptr, 1e-7, 1.0) != 0);
  CHECK(famseq_model_init(&m, 21, id3, mo3, fa3, sex3, nullptr, 1e-7, 1.0) != 0);
  {  // a member who is their own ancestor: refused at famseq_create
    const int32_t good_mo[] = {0, 0, 2}, good_fa[] = {0, 0, 1};
    CHECK(famseq_model_init(&m, 3, id3, good_mo, good_fa, sex3, nullptr, 1e-7, 1.0) == 0);
    m.mother[1] = 2;  // 2's mother is her own child 3 ...
    m.father[1] = 0;
    char err[128] = "";
    CHECK(famseq_create(&m, -1, err, sizeof err) == nullptr && std::strlen(err) > 0);
    CHECK(famseq_create(nullptr, -1, err, sizeof err) == nullptr);
  }
  {  // the size-independent model: a 40-member chain through famseq_pedigree (the generators run, nothing compiles);
     // creation fails for want of a compiler — cleanly, with the reason — and the argument checks hold
    const Ped w = chain(40);
    const int n = (int)w.id.size();
    std::vector<int32_t> mo(n), fa(n);
    famseq_pedigree p;
    CHECK(famseq_pedigree_init(&p, n, w.id.data(), w.mo.data(), w.fa.data(), w.sex.data(), nullptr, 1e-7, 1.0, mo.data(), fa.data()) == 0);
    CHECK(p.n_members == n && p.mother == mo.data() && p.sequenced == nullptr);
    famseq::Model model(p);
    size_t bytes = 0;
    for (int v = 0; v < famseq::kElimVariants; ++v) bytes += famseq::elim_source(model, v).size();
    CHECK(bytes > 100000);
    char err[256] = "";
    CHECK(famseq_create_pedigree(&p, -1, err, sizeof err) == nullptr && std::strstr(err, "sum-product engine only") != nullptr);
    CHECK(famseq_create_pedigree(nullptr, -1, err, sizeof err) == nullptr);
    CHECK(famseq_pedigree_init(&p, n, w.id.data(), w.mo.data(), w.fa.data(), w.sex.data(), nullptr, 1e-7, 1.0, nullptr, fa.data()) == FAMSEQ_E_ARG);
    p.mother = nullptr;  // a struct without its arrays is refused, not dereferenced
    CHECK(famseq_create_pedigree(&p, -1, err, sizeof err) == nullptr);
    std::printf("chain40  N=40 through famseq_pedigree: %zu bytes of generated source\n", bytes);
  }
  double t0[27], t1[27], t2[27];
  for (double mu : {0.0, 1e-7, 1e-3, 0.5}) famseq_transmission_tables(mu, t0, t1, t2);
  CHECK(famseq_last_error(nullptr) != nullptr);
  famseq_destroy(nullptr);
  std::string cmd = std::string("rm -rf '") + dir + "'";
  if (std::system(cmd.c_str()) != 0) ++fails;
  std::printf("asan_host_check: %s\n", fails ? "FAILED" : "ok");
  return fails ? 1 : 0;
}
