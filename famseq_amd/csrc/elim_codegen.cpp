// elim_codegen.cpp — generates the exact sum-product ("elimination") kernel for one pedigree.
//
// Same marginals as the 3^N enumeration of family::calPostProbBN
// (/root/reference/src/family.cpp:882-954, :990-1120), computed by message passing on the
// pedigree's factor graph instead of visiting every joint genotype:
//   variable nodes = members, factor nodes = nuclear families (mother, father, children) with
//   Phi_F = prod_children T_c[g_c | g_m, g_f]; member-local factors are prior*lk for founders
//   and lk for the others (exactly the indProb terms of family.cpp:899-909).
// On a loop-free pedigree (the domain of the reference's own Elston-Stewart -method 2,
// family.cpp:1126-1403) the factor graph is a forest and two messages per edge give every
// marginal exactly: O(27 N) flops per site instead of 2*3^N, which makes the path HBM-bound.
// A pedigree with loops (consanguinity, marriage loops) is made a forest by conditioning on the
// smallest set of members that cuts every cycle (up to three): their genotypes are enumerated in a
// loop around the message passing, 3^|cut| passes per site (Emitter::conditioned_body).
//
// The kernel is emitted as straight-line HIP for this one topology: every index is a literal,
// so a lane keeps a whole site (likelihoods, messages) in registers; one lane = one site.
// Arithmetic: fp64; sums of products written as explicit FMA chains; the single posterior, the
// shortcut vote and the failure rules are the same statements as in bn_kernel.hip (bit-identical
// to the CPU reference).  One member per connected component carries the reference's 1e7 scale.
#include "elim_codegen.h"

#include <cstdlib>
#include <functional>
#include <map>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <vector>

namespace famseq {

namespace {

struct Family {
  int mo, fa;
  std::vector<int> kids;
};

struct Graph {
  int N = 0;
  std::vector<Family> fam;
  std::vector<std::vector<int>> nb;  // member -> adjacent families
  std::vector<int> scaled;           // loop-free pedigrees: one member per connected component carries 1e7
  // Pedigrees with loops: the members of `cut` are conditioned on (their genotypes are enumerated
  // around the message passing), which leaves a forest.  comp[p] = component of the non-cut member p
  // in that forest, rep[c] = one member of component c.
  std::vector<int> cut, comp, rep;
  bool is_cut(int p) const {
    for (int c : cut)
      if (c == p) return true;
    return false;
  }
};

// Union-find pass over the bipartite member/family graph without the members in `cut`.
// Returns false on a cycle.  comp (optional) receives a component id per member (-1 for cut ones).
bool forest_without(const Graph &g, const std::vector<int> &cut, std::vector<int> *comp, std::vector<int> *rep) {
  const int nf = (int)g.fam.size();
  std::vector<int> parent(g.N + nf);
  std::iota(parent.begin(), parent.end(), 0);
  std::function<int(int)> find = [&](int x) { return parent[x] == x ? x : parent[x] = find(parent[x]); };
  auto cutm = [&](int p) {
    for (int c : cut)
      if (c == p) return true;
    return false;
  };
  for (int f = 0; f < nf; ++f) {
    std::vector<int> mem = {g.fam[f].mo, g.fam[f].fa};
    mem.insert(mem.end(), g.fam[f].kids.begin(), g.fam[f].kids.end());
    int live = 0;
    for (int p : mem) {
      if (cutm(p)) continue;
      ++live;
      const int a = find(p), b = find(g.N + f);
      if (a == b) return false;
      parent[a] = b;
    }
    if (!cut.empty() && live == 0) return false;  // a family of conditioned members only: not handled
  }
  if (comp) {
    comp->assign(g.N, -1);
    rep->clear();
    std::map<int, int> id;
    for (int p = 0; p < g.N; ++p) {
      if (cutm(p)) continue;
      const int r = find(p);
      if (!id.count(r)) {
        id[r] = (int)rep->size();
        rep->push_back(p);
      }
      (*comp)[p] = id[r];
    }
  }
  return true;
}

bool build_graph(const Model &m, Graph &g, std::string *why) {
  g.N = m.n_members;
  std::map<std::pair<int, int>, int> idx;
  g.nb.assign(g.N, {});
  for (int i = 0; i < g.N; ++i) {
    if (m.mother[i] < 0) continue;
    const auto key = std::make_pair(m.mother[i], m.father[i]);
    auto it = idx.find(key);
    if (it == idx.end()) {
      it = idx.emplace(key, (int)g.fam.size()).first;
      g.fam.push_back({key.first, key.second, {}});
      if (key.first == key.second) {
        if (why) *why = "a member's mother and father are the same individual";
        return false;
      }
      g.nb[key.first].push_back(it->second);
      g.nb[key.second].push_back(it->second);
    }
    g.fam[it->second].kids.push_back(i);
    g.nb[i].push_back(it->second);
  }
  if (forest_without(g, {}, &g.comp, &g.rep)) {
    g.scaled = g.rep;  // one member per connected component
    return true;
  }
  // A loop (consanguinity, marriage loop): the smallest set of members whose conditioning breaks
  // every cycle, up to three of them (3^|cut| passes of message passing per site).
  for (int size = 1; size <= 3; ++size) {
    std::vector<int> pick(size);
    std::function<bool(int, int)> search = [&](int k, int from) {
      if (k == size) return forest_without(g, pick, &g.comp, &g.rep);
      for (int p = from; p < g.N; ++p) {
        pick[k] = p;
        if (search(k + 1, p + 1)) return true;
      }
      return false;
    };
    if (search(0, 0)) {
      g.cut = pick;
      return true;
    }
  }
  if (why) *why = "the pedigree's loops need more than three conditioning members; use the enumeration engine";
  return false;
}

class Emitter {
 public:
  // out: where the normalised marginals go — "q" (registers: the shell's compute-first flow) or "row" (the lane's LDS row,
  // free once the likelihoods sit in registers: the shell's registers-first flow)
  Emitter(const Model &m, const Graph &g, int fences, bool scalar_t, const char *out = "q", bool lean = false, int lean_from = 1 << 30)
      : m_(m), g_(g), fences_(fences), scalar_t_(scalar_t), out_(out), lean_(lean), lean_from_(lean_from) {}

  std::string body() {
    if (g_.cut.empty()) {
      for (int p = 0; p < g_.N; ++p) {
        marginal(p);
        normalise(p, "m" + num(p));
      }
      return o_.str();
    }
    return conditioned_body();
  }

  // Pedigree with loops: enumerate the genotypes of the cut members around the message passing.
  // For one assignment a of the cut members every family sees an indicator in their place, the rest
  // is a forest, and with  Z_c = total weight of component c,  L = 1e7 * prod_cut local(a):
  //   member p of component c :  acc[p][g] += m_p[g] * L * prod_{c' != c} Z_c'
  //   cut member k            :  acc[k][a_k] += L * prod_c Z_c
  // (unnormalised marginals: the normalisation happens once, after the last assignment).
  std::string conditioned_body() {
    const int nc = (int)g_.cut.size(), ncomp = (int)g_.rep.size();
    int total = 1;
    for (int k = 0; k < nc; ++k) total *= 3;
    std::ostringstream head;
    for (int p = 0; p < g_.N; ++p) head << "      double acc" << p << "_0 = 0, acc" << p << "_1 = 0, acc" << p << "_2 = 0;\n";
    head << "#pragma unroll 1\n      for (int as_ = 0; as_ < " << total << "; ++as_) {\n";
    int div = 1;
    for (int k = 0; k < nc; ++k) {
      head << "      const int a" << g_.cut[k] << " = (as_ / " << div << ") % 3;\n";
      div *= 3;
    }
    // local factors of the cut members at their assigned genotype, and the reference's 1e7
    std::string lam = "10000000.0";
    for (int k : g_.cut) {
      const std::string c = loc(k);
      o_ << "      const double lam" << k << " = a" << k << " == 0 ? " << c << "_0 : (a" << k << " == 1 ? " << c << "_1 : " << c
         << "_2);\n";
      lam = "(" + lam + " * lam" + num(k) + ")";
    }
    o_ << "      const double Lam = " << lam << ";\n";
    for (int c = 0; c < ncomp; ++c) {
      marginal(g_.rep[c]);
      o_ << "      const double Zc" << c << " = (m" << g_.rep[c] << "_0 + m" << g_.rep[c] << "_1) + m" << g_.rep[c] << "_2;\n";
    }
    for (int c = 0; c < ncomp; ++c) {  // weight of everything outside component c
      std::string w = "Lam";
      for (int c2 = 0; c2 < ncomp; ++c2)
        if (c2 != c) w = "(" + w + " * Zc" + num(c2) + ")";
      o_ << "      const double Wc" << c << " = " << w << ";\n";
    }
    std::string all = "Lam";
    for (int c = 0; c < ncomp; ++c) all = "(" + all + " * Zc" + num(c) + ")";
    o_ << "      const double Wall = " << all << ";\n";
    for (int p = 0; p < g_.N; ++p) {
      if (g_.is_cut(p)) {
        for (int g = 0; g < 3; ++g) o_ << "      acc" << p << "_" << g << " += a" << p << " == " << g << " ? Wall : 0.0;\n";
        continue;
      }
      marginal(p);
      for (int g = 0; g < 3; ++g)
        o_ << "      acc" << p << "_" << g << " = __builtin_fma(m" << p << "_" << g << ", Wc" << g_.comp[p] << ", acc" << p << "_" << g
           << ");\n";
      fence(1);
    }
    std::string out = head.str() + o_.str() + "      }\n";
    o_.str("");
    for (int p = 0; p < g_.N; ++p) normalise(p, "acc" + num(p));
    return out + o_.str();
  }

 private:
  const Model &m_;
  const Graph &g_;
  const int fences_;  // 0 none, 1 after every family->member message, 2 also after local factors and child sums
  const bool scalar_t_;  // transmission entries from tcx[] (uniform pointer: scalar loads) instead of the lane's LDS table
  const std::string out_;
  const bool lean_;  // local factors re-formed at each use (see loc)
  const int lean_from_;  // ... for members lean_from_ and above only (their likelihoods sit in the lane's LDS row: direct_shell)
  std::ostringstream o_;
  std::map<std::string, bool> done_;
  int uid_ = 0;

  static std::string num(int x) { return std::to_string(x); }
  // Compiler fence: LDS reads (table entries, likelihoods) may not be hoisted above it.  Without
  // it hipcc front-loads the reads of the whole straight-line program and spills to scratch,
  // which costs real HBM traffic in a kernel that is otherwise memory-bound.  Small pedigrees fit
  // without (and run faster: the blocks overlap), so the fences are a variant (elim_source).
  void fence(int level) { if (fences_ >= level) o_ << "      asm volatile(\"\" ::: \"memory\");\n"; }
  bool once(const std::string &key) {
    if (done_.count(key)) return false;
    done_[key] = true;
    return true;
  }
  int kind(int p) const {
    const bool male = m_.gender[p] == 1;
    return m_.mother[p] < 0 ? (male ? 0 : 1) : (male ? 2 : 3);
  }
  std::string T(int child, int gc, int gm, int gf) const {
    return std::string(scalar_t_ ? "tcx[" : "tcf[") + num(kind(child) * 27 + 9 * gc + 3 * gm + gf) + "]";
  }

  // member-local factor c{p}_g
  std::string loc(int p) {
    const std::string n = "c" + num(p);
    if (once(n)) {
      bool scale = false;
      for (int s : g_.scaled) scale |= s == p;
      for (int g = 0; g < 3; ++g) {
        std::string e = "l" + num(p) + "_" + num(g);
        if (m_.mother[p] < 0) e = "(tcf[" + num(kind(p) * 27 + 9 * g) + "] * " + e + ")";
        if (scale) e = "(10000000.0 * " + e + ")";
        // lean: not a variable but a macro — the factor is formed again at each of its two or three uses, from a fresh read of
        // the likelihood, instead of living in a register from the first use to the last (3N doubles: the widest pedigrees'
        // register wall)
        if (lean_ || p >= lean_from_) o_ << "#define " << n << "_" << g << " " << e << "\n";
        else o_ << "      const double " << n << "_" << g << " = " << e << ";\n";
      }
      if (!(lean_ || p >= lean_from_)) fence(2);
    }
    return n;
  }

  // member -> family message v{p}f{F}_g = local * prod of the other families' messages
  std::string var2fac(int p, int F) {
    const std::string n = "v" + num(p) + "f" + num(F);
    if (g_.is_cut(p)) {  // a conditioned member: every family sees its assigned genotype, nothing flows through
      if (once(n))
        for (int g = 0; g < 3; ++g)
          o_ << "      const double " << n << "_" << g << " = a" << p << " == " << g << " ? 1.0 : 0.0;\n";
      return n;
    }
    if (once(n)) {
      std::vector<std::string> in = {loc(p)};
      for (int F2 : g_.nb[p])
        if (F2 != F) in.push_back(fac2var(F2, p));
      for (int g = 0; g < 3; ++g) {
        o_ << "      const double " << n << "_" << g << " = ";
        for (size_t k = 0; k < in.size(); ++k) o_ << (k ? " * " : "") << in[k] << "_" << g;
        o_ << ";\n";
      }
    }
    return n;
  }

  // child summary a<id>_{gm}{gf} = sum_gc T_c[gc|gm,gf] * v{c}f{F}_gc.  Deliberately NOT memoised:
  // a summary is needed by several messages of its family, and keeping 9 doubles per child alive
  // across the whole program costs more (registers -> scratch -> HBM traffic) than recomputing
  // 27 multiply-adds in a kernel whose vector ALU is mostly idle.
  std::string child_sum(int F, int c) {
    const std::string x = var2fac(c, F);
    const std::string n = "a" + num(uid_++);
    for (int gm = 0; gm < 3; ++gm)
      for (int gf = 0; gf < 3; ++gf)
        o_ << "      const double " << n << "_" << gm << gf << " = __builtin_fma(" << T(c, 2, gm, gf) << ", " << x
           << "_2, __builtin_fma(" << T(c, 1, gm, gf) << ", " << x << "_1, " << T(c, 0, gm, gf) << " * " << x << "_0));\n";
    fence(2);
    return n;
  }

  // family -> member message f{F}v{t}_g
  std::string fac2var(int F, int t) {
    const std::string n = "f" + num(F) + "v" + num(t);
    if (!once(n)) return n;
    const Family &fam = g_.fam[F];
    std::vector<std::string> sums;
    for (int c : fam.kids)
      if (c != t) sums.push_back(child_sum(F, c));
    const std::string xm = t == fam.mo ? "" : var2fac(fam.mo, F);
    const std::string xf = t == fam.fa ? "" : var2fac(fam.fa, F);
    // C_{gm}{gf}: product of the other children's summaries, times the parents' messages present
    const std::string C = n + "w";
    for (int gm = 0; gm < 3; ++gm)
      for (int gf = 0; gf < 3; ++gf) {
        std::vector<std::string> terms;
        if (!xm.empty()) terms.push_back(xm + "_" + num(gm));
        if (!xf.empty()) terms.push_back(xf + "_" + num(gf));
        for (const std::string &s : sums) terms.push_back(s + "_" + num(gm) + num(gf));
        o_ << "      const double " << C << "_" << gm << gf << " = ";
        if (terms.empty()) o_ << "1.0";
        for (size_t k = 0; k < terms.size(); ++k) o_ << (k ? " * " : "") << terms[k];
        o_ << ";\n";
      }
    for (int g = 0; g < 3; ++g) {
      o_ << "      const double " << n << "_" << g << " = ";
      if (t == fam.mo) {
        o_ << "(" << C << "_" << g << "0 + " << C << "_" << g << "1) + " << C << "_" << g << "2";
      } else if (t == fam.fa) {
        o_ << "(" << C << "_0" << g << " + " << C << "_1" << g << ") + " << C << "_2" << g;
      } else {
        std::string e;
        for (int gm = 0; gm < 3; ++gm)
          for (int gf = 0; gf < 3; ++gf) {
            const std::string term = T(t, g, gm, gf) + ", " + C + "_" + num(gm) + num(gf);
            e = e.empty() ? "(" + T(t, g, gm, gf) + " * " + C + "_" + num(gm) + num(gf) + ")"
                          : "__builtin_fma(" + term + ", " + e + ")";
          }
        o_ << e;
      }
      o_ << ";\n";
    }
    fence(1);
    return n;
  }

  // unnormalised marginal m{p}_g = local * prod of the messages of the adjacent families
  void marginal(int p) {
    if (!once("m" + num(p))) return;
    std::vector<std::string> in = {loc(p)};
    for (int F : g_.nb[p]) in.push_back(fac2var(F, p));
    for (int g = 0; g < 3; ++g) {
      o_ << "      const double m" << p << "_" << g << " = ";
      for (size_t k = 0; k < in.size(); ++k) o_ << (k ? " * " : "") << in[k] << "_" << g;
      o_ << ";\n";
    }
  }

  // row p of the output: `from`_g / sum, with the reference's failure rule (family.cpp:943-954)
  void normalise(int p, const std::string &from) {
    // one division per row and three products (this engine is not bit-ordered anyway); a row sum in the subnormal range,
    // whose reciprocal overflows, keeps the three divisions behind a real branch (as the enumeration kernel does)
    auto o3 = [&](int g) { return out_ + "[" + num(3 * p + g) + "]"; };
    o_ << "      { const double s = (" << from << "_0 + " << from << "_1) + " << from << "_2; if (s <= 0) bn_fail = true;\n"
       << "        if (s < 1e-290) { asm volatile(\"\" ::: \"memory\"); " << o3(0) << " = " << from << "_0 / s; " << o3(1) << " = " << from << "_1 / s; "
       << o3(2) << " = " << from << "_2 / s; }\n"
       << "        else { const double r = 1.0 / s; " << o3(0) << " = " << from << "_0 * r; " << o3(1) << " = " << from << "_1 * r; " << o3(2) << " = "
       << from << "_2 * r; } }\n";
  }
};

}  // namespace

bool elim_supported(const Model &m, std::string *why) {
  Graph g;
  return build_graph(m, g, why);
}

int elim_conditioned_members(const Model &m) {
  Graph g;
  return build_graph(m, g, nullptr) ? (int)g.cut.size() : -1;
}

// One-wave workgroups and no register cap, as for the enumeration kernel (enumgen_block_threads): 8 M five-member
// sites 0.662 -> 0.617 ms, quads 0.507 -> 0.481; six members 0.459 -> 0.403 ms per 4 M sites, seven 0.570 -> 0.512,
// eight 0.718 -> 0.607 (the fence-free variant at ONE wave per SIMD with its overflow in AGPRs), ten 0.664 -> 0.640 and
// fifteen 1.118 -> 1.047 (the fence-per-message variant, which fits two waves): profiles/r02c/exp_elim_waves*.txt.
// Which variant runs best is a property of the pedigree more than of its size: on 90 randomly grown pedigrees of 8-14
// members (PL-shaped rows; 2 M sites and, streaming from HBM, 8 M: profiles/r02c/tune_survey_*_sites.txt) the fence-free
// variant wins two times in three, and starting from it loses 3.1 % on average to the better of the two where "the
// fenced one from nine members on" (fitted to the two benchmark pedigrees, which both prefer the fenced one) loses
// 5.6 %.  The one size that goes the other way in both surveys is eleven members (15 of 17 pedigrees, by 10 % on
// average): with that exception 1.3 %.  So the picker starts from the fence-free variant, at eleven members from the
// fenced one, and a pedigree that has been measured — famseq_set_option "tune", or the table of measured picks that
// build() ships for the pedigrees it pre-builds (the benchmark pedigrees among them) — starts from its note.
// The fused call-path form is another kernel — paced by the sixty logarithms per site between its barriers, not by
// memory — and keeps 256-lane workgroups at two waves per SIMD: 0.317 ms per 1 M ten-member sites against 0.487 in
// one-wave workgroups (tools/call_ab.sh).
int elim_block_threads(const Model &m, bool call_mode) {
  if (const char *e = std::getenv("FAMSEQ_ELIM_BT")) return std::atoi(e);  // tuning aid
  if (call_mode) return m.n_members <= 10 ? 256 : 128;
  return 64;
}

// Registers-first from five members on (see kElimVariants); within a family the fence-free variant, with which
// jit_pick_variant starts unless it spills.  (Round 2's "the fenced variant at eleven members" was fitted to the r = 0
// family's survey and is not carried over; a pedigree that has been MEASURED starts from its note either way.)
// From forty members on the form without LDS staging (variants 8..): at 48 members the staged form's 74 KB of LDS rows per wave
// leave two waves per CU and it runs at 0.28 of HBM peak (the unstaged one, four waves per CU: 0.35); from about a hundred the
// rows no longer fit the CU's 160 KB at all (profiles/r03a/exp_elim_unstaged.txt: 24 members 0.52 staged / 0.37 unstaged, 32:
// 0.52 / 0.38, 48: 0.28 / 0.35, 64: - / 0.24, 96: - / 0.15 — past sixty members the registers spill and scratch traffic sets the pace).
int elim_first_variant(const Model &m, bool call_mode) {
  if (call_mode) return 0;
  return m.n_members >= 40 ? 8 : (m.n_members >= 5 ? 4 : 0);
}

// Text every generated kernel carries for the fused call path (SURVEY.md 8(f) rows N2 + N4): packed
// integer PLs staged straight into the LDS rows (the reference's lk = pow(10, -|PL| / 10), file.cpp:588-590,
// through the host-filled table; missing sample / unsequenced member = {1,1,1}, :565, :794-809) and the
// drivers' per-sample outputs formed while the rows are stored: GPP / FPP = fabs(-10 log10 p), +inf ->
// 99999 (file.cpp:696-745) and FGT = arg-max with strict '<' from -1 (family.cpp:636-665), gathered in
// VCF column order.  A failed site's rows are NaN, which gives NaN Phred values and FGT -1 by itself.
#define FS_PHRED_DEF(...) #__VA_ARGS__
const char FS_PHRED_TEXT_[] =
#include "phred_src.h"
    ;
#undef FS_PHRED_DEF
#define FS_TAB_ROW(a, b) #a ", " #b ",\n"
const char FS_TAB_TEXT_[] =
#include "phred_tab.h"
    ;
#undef FS_TAB_ROW
const std::string kCallHelpers = std::string(R"(
#ifndef FS_RCP
#define FS_RCP(x) __builtin_amdgcn_rcp(x)
#define FS_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define FS_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define FS_UMULHI(a, b) __umulhi(a, b)
#define FS_IS_POS_FINITE(x) __builtin_amdgcn_class(x, 0x180)  /* +normal | +denormal */
#define FS_KEEP_BRANCH() asm volatile("" ::: "memory")
#define FS_HI32(x) __double2hiint(x)
#endif
typedef double fs_v2d __attribute__((ext_vector_type(2)));
// Pointers that reach the kernel through the argument struct are generic to the compiler, and every access
// through them would be a flat_* instruction (counted on both vmcnt and lgkmcnt, waited for with both at 0).
// They are global memory: say so.
#ifndef FS_GLOBAL
#define FS_GLOBAL __attribute__((address_space(1)))
#endif
)") + FS_PHRED_TEXT_ + "\n// fs_phred's table (phred_tab.h): staged into LDS (s_lt) at kernel start\n__device__ const double fs_logtab[258] = {\n" +
                                 FS_TAB_TEXT_ + "};\n" + std::string(R"(
// The call path's arguments sit in one small struct in device memory behind a pointer that is null on the
// plain path: its fields are fetched (scalar loads) only inside the stages that use them.  As ten more
// kernel arguments they stayed live in SGPRs for the whole kernel and pushed the arithmetic into scratch.
struct fs_call_args {
  const FS_GLOBAL unsigned short *pl;  // [n_sites][n_seq][3] packed PLs, or null: fp64 likelihood rows come in as usual
  const FS_GLOBAL double *lut;         // pow(10, -k / 10), k < 4096
  const FS_GLOBAL int *col, *slot;     // member -> VCF column or -1; member -> its slot of the output row (its column, or one behind the columns)
  FS_GLOBAL double *gpp, *fpp;         // [n_sites][n_seq][3], either may be null
  FS_GLOBAL signed char *fgt;          // [n_sites][n_seq] or null
  int n_seq;
  unsigned magic_w, magic_n;  // 2^32 / (3 n_seq) + 1, 2^32 / n_seq + 1: e / d = umulhi(e, magic) for e < 2^16; magic_n = 0 when n_seq = 1
};
// packed PLs of the chunk -> likelihood rows in LDS: one (site, member) item per lane and step.  Whole chunks
// take the unrolled, predicate-free walk (all of a lane's loads in flight together); the table look-ups
// depend on the PLs, so that is two memory latencies per chunk instead of two per member.
#define PL_ITEM(it_) { const int s_ = (it_) / NMEM, i_ = (it_) - s_ * NMEM, c_ = s_col[i_]; \
    double v0_ = 1.0, v1_ = 1.0, v2_ = 1.0; \
    if (c_ >= 0) { const FS_GLOBAL unsigned short *q_ = p_ + (s_ * n_seq + c_) * 3; const unsigned a_ = q_[0], b_ = q_[1], d_ = q_[2]; \
      if (!(a_ == 0xFFFFu && b_ == 0xFFFFu && d_ == 0xFFFFu)) { \
        v0_ = a_ < 4096u ? lut_[a_] : 0.0; v1_ = b_ < 4096u ? lut_[b_] : 0.0; v2_ = d_ < 4096u ? lut_[d_] : 0.0; } } \
    double *w_ = s_io + s_ * ROW + 3 * i_; w_[0] = v0_; w_[1] = v1_; w_[2] = v2_; }
#define STAGE_IN_PL() { const int n_seq = call_g->n_seq; const FS_GLOBAL double *lut_ = call_g->lut; \
  const FS_GLOBAL unsigned short *p_ = call_g->pl + site0 * n_seq * 3; \
  if (whole) { _Pragma("unroll") for (int j_ = 0; j_ < NMEM; ++j_) PL_ITEM(tid + j_ * BT) } \
  else { for (int it_ = tid; it_ < ns * NMEM; it_ += BT) PL_ITEM(it_) } }
// What is printed, formed from REGISTERS and laid down in OUTPUT order.  V[3 p + g] holds member p's probabilities (single or
// BN posterior, NaN where the site failed); member p's printed values fabs(-10 log10 p) go to slot slot_r_[p] of the lane's row —
// its VCF column if it has one, a slot behind the columns otherwise — and its arg-max genotype (strict '<' from -1: ties to the
// lower genotype, NaN rows give -1) to the same slot of the byte table.  The rows then ARE the output, site by site: the
// stage-out is a copy (rounds 1-3 kept member order in the row and gathered on the way out: an index read, a value read and
// twenty integer instructions per element, 36 % of the kernel's wave cycles at ten members).  Reading from registers also makes
// the permutation safe (nothing of the row is read while it is rewritten) and spares the row a round trip.
// 3 N independent logarithms per lane — independent for the scheduler only inside one basic block: with fs_phred's own branch
// per logarithm (the special values) each one was a block of its own, table read -> wait -> nine dependent FMAs, thirty times in
// a row.  Two members at a time, the six logarithms branch-free and ONE branch behind them for the rare row with a zero or a NaN.
/* (the lowest genotype within 1e-12 relative of the largest posterior: exact ties of the reference — 0.5 / 0.5 at mutation rate 0 —
   are ties to rounding here; model.cpp famseq_call_genotypes) */
#define ARGMAX3(a0_, a1_, a2_, slot_) { double bs_ = -1; \
    if (bs_ < a0_) bs_ = a0_; if (bs_ < a1_) bs_ = a1_; if (bs_ < a2_) bs_ = a2_; \
    const double th_ = bs_ * (1.0 - 1e-12); \
    s_fgt[tid * NMEM + (slot_)] = bs_ < 0 ? (signed char)-1 : (a0_ >= th_ ? (signed char)0 : (a1_ >= th_ ? (signed char)1 : (signed char)2)); }
#define PUT_CALL(V) { _Pragma("unroll") for (int p_ = 0; p_ + 1 < (FS_PHRED_GROUP == 2 ? NMEM : 0); p_ += 2) { \
    const int o0_ = slot_r_[p_], o1_ = slot_r_[p_ + 1]; \
    const double d0_ = V[3 * p_], d1_ = V[3 * p_ + 1], d2_ = V[3 * p_ + 2], d3_ = V[3 * p_ + 3], d4_ = V[3 * p_ + 4], d5_ = V[3 * p_ + 5]; \
    ARGMAX3(d0_, d1_, d2_, o0_) ARGMAX3(d3_, d4_, d5_, o1_) \
    double q0_ = fs_phred_fast(d0_, s_lt), q1_ = fs_phred_fast(d1_, s_lt), q2_ = fs_phred_fast(d2_, s_lt); \
    double q3_ = fs_phred_fast(d3_, s_lt), q4_ = fs_phred_fast(d4_, s_lt), q5_ = fs_phred_fast(d5_, s_lt); \
    if (!(FS_IS_POS_FINITE(d0_) & FS_IS_POS_FINITE(d1_) & FS_IS_POS_FINITE(d2_) & FS_IS_POS_FINITE(d3_) & FS_IS_POS_FINITE(d4_) & FS_IS_POS_FINITE(d5_))) { \
      FS_KEEP_BRANCH(); q0_ = fs_phred_fix(d0_, q0_); q1_ = fs_phred_fix(d1_, q1_); q2_ = fs_phred_fix(d2_, q2_); \
      q3_ = fs_phred_fix(d3_, q3_); q4_ = fs_phred_fix(d4_, q4_); q5_ = fs_phred_fix(d5_, q5_); } \
    double *w0_ = row + 3 * o0_, *w1_ = row + 3 * o1_; \
    w0_[0] = q0_; w0_[1] = q1_; w0_[2] = q2_; w1_[0] = q3_; w1_[1] = q4_; w1_[2] = q5_; } \
  _Pragma("unroll") for (int p_ = (FS_PHRED_GROUP == 2 ? (NMEM & ~1) : 0); p_ < NMEM; ++p_) { const int o0_ = slot_r_[p_]; \
    const double d0_ = V[3 * p_], d1_ = V[3 * p_ + 1], d2_ = V[3 * p_ + 2]; \
    ARGMAX3(d0_, d1_, d2_, o0_) \
    double q0_ = fs_phred_fast(d0_, s_lt), q1_ = fs_phred_fast(d1_, s_lt), q2_ = fs_phred_fast(d2_, s_lt); \
    if (!(FS_IS_POS_FINITE(d0_) & FS_IS_POS_FINITE(d1_) & FS_IS_POS_FINITE(d2_))) { \
      FS_KEEP_BRANCH(); q0_ = fs_phred_fix(d0_, q0_); q1_ = fs_phred_fix(d1_, q1_); q2_ = fs_phred_fix(d2_, q2_); } \
    double *w0_ = row + 3 * o0_; w0_[0] = q0_; w0_[1] = q1_; w0_[2] = q2_; } }
// ... and from a row that holds them in member order (the registers-first shells, whose body writes its marginals there): every
// value is read before the first is written
#define ROW_TO_CALL() { double u_[W3]; _Pragma("unroll") for (int k_ = 0; k_ < W3; ++k_) u_[k_] = row[k_]; PUT_CALL(u_) }
// rows -> [site][VCF column][genotype]: element e = s w + r of the chunk sits at s ROW + r (w = 3 n_seq <= ROW); FGT likewise
// from the byte table (s NMEM + k)
#define OUT_ELEM(e_) { const int s_ = (int)FS_UMULHI((unsigned)(e_), mg_), r_ = (e_) - s_ * w_; \
    __builtin_nontemporal_store(s_io[s_ * ROW + r_], g_ + (e_)); }
// two neighbouring elements per lane and store (16 B; the second may be the next site's first)
#define OUT_PAIR(p2_) { const int e_ = 2 * (p2_), s_ = (int)FS_UMULHI((unsigned)e_, mg_), r_ = e_ - s_ * w_, a_ = s_ * ROW + r_; \
    fs_v2d v_; v_.x = s_io[a_]; v_.y = s_io[r_ + 1 == w_ ? a_ + 1 + (ROW - w_) : a_ + 1]; \
    __builtin_nontemporal_store(v_, (FS_GLOBAL fs_v2d *)(g_ + e_)); }
// (NSEQ_CT: the pedigree's number of sequenced members, what n_seq is unless the caller names another set of columns.  With the
// row width a constant the walk over a whole chunk has no bound to test per step — each test was a branch, each branch a basic
// block of its own — and its loads go out together.)
// FS_CT_OUT = 2: the walk in groups of four steps, fenced one from the next (fewer loads in flight, fewer registers).
// The walks' indices depend on the lane only.  Left alone, hipcc computes them once, before the chunk loop — free while registers
// are (small pedigrees), sixty spilled registers at the 256 cap otherwise.  FS_OPAQUE_LANE (from seven members on; the
// enumeration's form from five) makes the lane id opaque at each walk, so that they are formed again per chunk.
#define FS_HIDE_LANE(t_) if (FS_OPAQUE_LANE) asm volatile("" : "+v"(t_))
#define FS_OUT_FENCE(j_) if (FS_CT_OUT == 2 && ((j_) & 3) == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define WOUT_CT (3 * NSEQ_CT)
#define OUT_PAIR_CT(p2_) { const int e_ = 2 * (p2_), s_ = e_ / WOUT_CT, r_ = e_ - s_ * WOUT_CT, a_ = s_ * ROW + r_; \
    fs_v2d v_; v_.x = s_io[a_]; v_.y = s_io[(WOUT_CT & 1) && r_ + 1 == WOUT_CT ? a_ + 1 + (ROW - WOUT_CT) : a_ + 1]; \
    __builtin_nontemporal_store(v_, (FS_GLOBAL fs_v2d *)(g_ + e_)); }
#define STAGE_OUT_CALL(Gp) { const int w_ = 3 * call_g->n_seq; const unsigned mg_ = call_g->magic_w; FS_GLOBAL double *g_ = (Gp) + site0 * w_; \
  if (FS_CT_OUT && whole && (BT & 1) == 0 && ((unsigned long)g_ & 15) == 0 && w_ == WOUT_CT) { \
    int t_ = tid; FS_HIDE_LANE(t_); \
    _Pragma("unroll") for (int j_ = 0; j_ < WOUT_CT / 2; ++j_) { OUT_PAIR_CT(t_ + j_ * BT) FS_OUT_FENCE(j_) } \
    if (WOUT_CT & 1) { if (t_ < BT / 2) OUT_PAIR_CT(t_ + WOUT_CT / 2 * BT) } } \
  else if (whole && (BT & 1) == 0 && ((unsigned long)g_ & 15) == 0) { const int half_ = BT / 2 * w_; \
    _Pragma("unroll") for (int j_ = 0; j_ < (3 * NMEM + 1) / 2; ++j_) if (tid + j_ * BT < half_) OUT_PAIR(tid + j_ * BT) } \
  else if (whole) { _Pragma("unroll") for (int j_ = 0; j_ < 3 * NMEM; ++j_) if (j_ < w_) OUT_ELEM(tid + j_ * BT) } \
  else { for (int e = tid; e < ns * w_; e += BT) OUT_ELEM(e) } }
#define STAGE_FGT(Gp) { const int n_seq = call_g->n_seq; const unsigned mg_ = call_g->magic_n; FS_GLOBAL signed char *g_ = (Gp) + site0 * n_seq; \
  if (FS_CT_OUT && whole && n_seq == NSEQ_CT) { int t_ = tid; FS_HIDE_LANE(t_); \
    _Pragma("unroll") for (int j_ = 0; j_ < NSEQ_CT; ++j_) { const int it_ = t_ + j_ * BT, s_ = it_ / NSEQ_CT, k_ = it_ - s_ * NSEQ_CT; \
    g_[it_] = s_fgt[s_ * NMEM + k_]; FS_OUT_FENCE(j_) } } \
  else for (int it_ = tid; it_ < ns * n_seq; it_ += BT) { const int s_ = mg_ ? (int)FS_UMULHI((unsigned)it_, mg_) : it_, k_ = it_ - s_ * n_seq; \
    g_[it_] = s_fgt[s_ * NMEM + k_]; } }
)");
// The sum-product kernel's call-path form spent 44 % of its wave cycles in STAGE_IN_PL (FAMSEQ_PHASE_CLOCK, ten members): 30 two-byte
// loads per lane, then 30 table look-ups that each touch up to 64 cache lines, both waited for by every wave of the workgroup
// at once.  "Flat" staging replaces it for whole chunks:
//   * the chunk's packed PLs are BT * n_seq * 6 contiguous bytes from a 16-byte boundary: fetched as 16-byte pieces, four per
//     lane, perfectly coalesced — and fetched for the NEXT chunk right after this chunk's message passing, so that they land
//     during the two output phases (16 registers through phases that have them to spare);
//   * the pieces go through the top of the row area (free at that moment), from where every lane picks its items' three
//     16-bit values;
//   * the first FS_LUT_LDS entries of the pow(10, -k / 10) table live in LDS (8 KB: what two 256-lane workgroups per CU leave
//     of the 160 KB at ten members): an LDS gather instead of an L1 one; larger PLs (rare) still go to the global table.
const char kCallFlat[] = R"(
typedef unsigned fs_v4u __attribute__((ext_vector_type(4)));
#define RAW_V4 ((BT * NMEM * 6 + 15) / 16)  /* n_seq <= NMEM */
#define RAW_K ((RAW_V4 + BT - 1) / BT)
/* the table value of a PL beyond the LDS part (rare: one real branch per item, nothing of it on the usual path) */
#define FS_LUT_BIG(x, v) ((x) < (unsigned)FS_LUT_LDS ? (v) : ((x) < 4096u ? lut_[x] : 0.0))
#define PL_FETCH(S0) { const FS_GLOBAL fs_v4u *r_ = (const FS_GLOBAL fs_v4u *)(call_g->pl + (S0) * call_g->n_seq * 3); \
  const int nv_ = (BT * call_g->n_seq * 6) / 16; int t_ = tid; FS_HIDE_LANE(t_); \
  _Pragma("unroll") for (int j_ = 0; j_ < RAW_K; ++j_) if (t_ + j_ * BT < nv_) praw[j_] = r_[t_ + j_ * BT]; }
#define STAGE_IN_PL_FLAT() { const int n_seq = call_g->n_seq; const FS_GLOBAL double *lut_ = call_g->lut; const int nv_ = (BT * n_seq * 6) / 16; \
  fs_v4u *raw4_ = (fs_v4u *)((char *)s_io + (BT * ROW * 8 - RAW_V4 * 16)); int t_ = tid; FS_HIDE_LANE(t_); \
  _Pragma("unroll") for (int j_ = 0; j_ < RAW_K; ++j_) if (t_ + j_ * BT < nv_) raw4_[t_ + j_ * BT] = praw[j_]; \
  LDS_BARRIER(); \
  const unsigned short *raw_ = (const unsigned short *)raw4_; \
  unsigned a_[NMEM], b_[NMEM], d_[NMEM]; \
  _Pragma("unroll") for (int j_ = 0; j_ < NMEM; ++j_) { const int it_ = t_ + j_ * BT, s_ = it_ / NMEM, i_ = it_ - s_ * NMEM, c_ = s_col[i_]; \
    a_[j_] = b_[j_] = d_[j_] = 0xFFFFu;  /* a member without a column is a missing sample: {1, 1, 1} */ \
    if (c_ >= 0) { const unsigned short *q_ = raw_ + (s_ * n_seq + c_) * 3; a_[j_] = q_[0]; b_[j_] = q_[1]; d_[j_] = q_[2]; } } \
  LDS_BARRIER();  /* every lane holds its items: the rows (the raw pieces' place among them) may be written */ \
  _Pragma("unroll") for (int j_ = 0; j_ < NMEM; ++j_) { const int it_ = t_ + j_ * BT, s_ = it_ / NMEM, i_ = it_ - s_ * NMEM; \
    const unsigned x0_ = a_[j_], x1_ = b_[j_], x2_ = d_[j_], top_ = FS_LUT_LDS - 1; \
    const bool miss_ = (x0_ & x1_ & x2_) == 0xFFFFu;  /* 16-bit values: all three 0xFFFF */ \
    double v0_ = s_lut[x0_ < top_ ? x0_ : top_], v1_ = s_lut[x1_ < top_ ? x1_ : top_], v2_ = s_lut[x2_ < top_ ? x2_ : top_]; \
    if (miss_) { v0_ = 1.0; v1_ = 1.0; v2_ = 1.0; } \
    else if ((x0_ | x1_ | x2_) >= (unsigned)FS_LUT_LDS) { FS_KEEP_BRANCH(); v0_ = FS_LUT_BIG(x0_, v0_); v1_ = FS_LUT_BIG(x1_, v1_); v2_ = FS_LUT_BIG(x2_, v2_); } \
    double *w_ = s_io + s_ * ROW + 3 * i_; w_[0] = v0_; w_[1] = v1_; w_[2] = v2_; } }
)";

// FAMSEQ_PHASE_CLOCK (measuring aid): the argument struct gets a counter array and the source a macro that adds the cycles since
// the previous mark to counter i, once per wave.  Off, the generated text — and so every cached code object — is unchanged.
std::string with_phase_clock(std::string h) {
  const std::string at = "magic_n = 0 when n_seq = 1\n};\n";
  const size_t p = h.find(at);
  if (p == std::string::npos) throw std::runtime_error("with_phase_clock: the argument struct has moved");
  h.insert(p + at.size() - 3,
           "  FS_GLOBAL unsigned long long *phase_clk;  // cycles per phase, one add per wave and phase\n");
  // (the sums stay in the wave's registers until the kernel ends: an atomic per mark made the kernel four times slower)
  h += "#define PH(i) { const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_##i += t_ - ph_last_; ph_last_ = t_; }\n"
       "#define PH_FLUSH(i) atomicAdd((unsigned long long *)call_g->phase_clk + (i), ph_acc_##i)\n";
  return h;
}
// ... and the arguments that go with it, after the plain ones: all null / 0 on the plain path
const char kCallArgs[] = ", const struct fs_call_args *__restrict__ call_g";

// Statements of the single posterior (family.cpp:1426-1445) and of the shortcut vote (:767-789), the
// same as in bn_kernel.hip; they read l<p>_<g> and tcf[], set single_fail / full, and (store) write
// the normalised rows to row[].  Shared by every generated shell.
// The single posterior's three quotients p0 / s, p1 / s, p2 / s (family.cpp:1437-1441) — bit for bit what `/` gives, at half its cost.
// hipcc's fp64 division is v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (four FMAs), q = n r, e = fma(-d, q, n),
// v_div_fmas (an FMA: fma(e, r, q)) and v_div_fixup: eleven instructions, of which v_div_scale (moving extreme operands into range)
// is the identity and v_div_fixup (zeros, infinities, denormal results) a copy whenever every operand and intermediate is a normal
// number anyway.  FS_DIV_OK says when that is certain — every numerator >= 2^-300, the common denominator <= 2^300 (it is >= a
// numerator) — and FS_DIV3_FAST then runs the SAME sequence with the reciprocal's five instructions done once for the three
// quotients: 14 + 5 for the test instead of 33 (and one quarter-rate v_rcp_f64 instead of three).  Anything else — a likelihood of
// exactly 0, a huge LK-file scale — takes the plain divisions.  Divisions were a quarter of a small pedigree's instructions:
// quad 0.556 -> 0.526, five members 0.714 -> 0.679 ms per 8 M sites before the test was added.  (A host build of this text
// defines both macros itself: FS_DIV_OK 0.)
const char kDiv3Text[] = R"(
#ifndef FS_DIV_OK
#define FS_DIV_OK(p0, p1, p2, s) ((__builtin_fmin(__builtin_fmin((p0), (p1)), (p2)) >= 0x1p-300) & ((s) <= 0x1p300))
#define FS_DIV3_FAST(p0, p1, p2, s, o0, o1, o2) { double r_ = __builtin_amdgcn_rcp(s), e_ = __builtin_fma(-(s), r_, 1.0); \
    r_ = __builtin_fma(r_, e_, r_); e_ = __builtin_fma(-(s), r_, 1.0); r_ = __builtin_fma(r_, e_, r_); \
    const double t0_ = (p0) * r_, t1_ = (p1) * r_, t2_ = (p2) * r_; \
    o0 = __builtin_fma(__builtin_fma(-(s), t0_, (p0)), r_, t0_); o1 = __builtin_fma(__builtin_fma(-(s), t1_, (p1)), r_, t1_); \
    o2 = __builtin_fma(__builtin_fma(-(s), t2_, (p2)), r_, t2_); }
#endif
)";

std::string single_posterior_statements(const Model &m, bool flags_pass, bool store, bool fence_single, const char *dst) {
  std::ostringstream s;
  const int N = m.n_members;
  // members in groups of four (one at a time in the fenced variants: interleaved division sequences would spill): the products
  // and sums of the group, ONE test and branch for its quotients, then the shortcut vote's part
  // (wide pedigrees, whose registers hold 3 N likelihoods: one at a time as well — four members' products pushed the 32-member
  // kernel's fence-free variant into scratch and the contest onto the fenced one, 1.06-1.13 -> 1.21 ms per 2 M sites)
  int G = fence_single || N > 12 ? 1 : 4;
  if (const char *e = std::getenv("FAMSEQ_DIV_GROUP")) G = std::max(1, std::atoi(e));  // tuning aid
  // (beyond twelve members — one member at a time, a branch each — it is a wash: fifteen members -2 %, 24: -1 %, 32: +2.5 %, 48: -4 %
  // per 2 M sites; plain divisions there)
  bool div_fast = N <= 12;
  if (const char *e = std::getenv("FAMSEQ_DIV_FAST")) div_fast = std::atoi(e) != 0;  // tuning aid: 0 = plain divisions everywhere
  for (int lo = 0; lo < N; lo += G) {
    const int hi = std::min(N, lo + G);
    s << "    {\n";
    for (int p = lo; p < hi; ++p) {
      const int fk = m.gender[p] == 1 ? 0 : 1;
      const std::string k = std::to_string(p);
      s << "      const double a" << k << "_0 = l" << p << "_0, a" << k << "_1 = l" << p << "_1, a" << k << "_2 = l" << p << "_2;\n"
        << "      const double p" << k << "_0 = a" << k << "_0 * tcf[" << fk * 27 << "], p" << k << "_1 = a" << k << "_1 * tcf[" << fk * 27 + 9
        << "], p" << k << "_2 = a" << k << "_2 * tcf[" << fk * 27 + 18 << "];\n      const double s" << k << " = (p" << k << "_0 + p" << k
        << "_1) + p" << k << "_2;";
      if (flags_pass) s << " if (s" << k << " <= 0) single_fail = true;";
      s << "\n";
    }
    if (store) {
      s << "      if (";
      for (int p = lo; p < hi; ++p)
        s << (p > lo ? " & " : "") << (div_fast ? "FS_DIV_OK(p" : "0 && FS_DIV_OK(p") << p << "_0, p" << p << "_1, p" << p << "_2, s" << p << ")";
      s << ") {\n";
      for (int p = lo; p < hi; ++p)
        s << "        FS_DIV3_FAST(p" << p << "_0, p" << p << "_1, p" << p << "_2, s" << p << ", " << dst << "[" << 3 * p << "], " << dst << "["
          << 3 * p + 1 << "], " << dst << "[" << 3 * p + 2 << "]);\n";
      s << "      } else {\n";
      for (int p = lo; p < hi; ++p)
        s << "        " << dst << "[" << 3 * p << "] = p" << p << "_0 / s" << p << "; " << dst << "[" << 3 * p + 1 << "] = p" << p << "_1 / s" << p
          << "; " << dst << "[" << 3 * p + 2 << "] = p" << p << "_2 / s" << p << ";\n";
      s << "      }\n";
    }
    for (int p = lo; p < hi; ++p)
      if (flags_pass && m.sequenced[p])
        s << "      { double big = 0; if (big < a" << p << "_0) big = a" << p << "_0; if (big < a" << p << "_1) big = a" << p << "_1; if (big < a"
          << p << "_2) big = a" << p << "_2;\n"
          << "        const double sum = (a" << p << "_0 + a" << p << "_1) + a" << p << "_2; big = big / sum; if (big < lc) full = true; }\n";
    s << "    }\n";
    if (fence_single) s << "    asm volatile(\"\" ::: \"memory\");\n";
  }
  return s.str();
}

// The part every generated engine shares: I/O staging through padded LDS rows, the single
// posterior, the shortcut vote and the status byte.  `body` runs for sites that need the full
// computation; it reads l<p>_<g> and tcf[...], and must set bn_fail on a row sum <= 0.
//   regs_l = true : the likelihood row is held in registers (l<p>_<g> are variables); the body
//                   writes the normalised marginals to row[0..W3) (the row is free by then).
//   regs_l = false: l<p>_<g> read the LDS row each time (short live ranges, no spills in the
//                   message-passing code); the body writes the marginals to q[0..W3) and runs
//                   BEFORE the single posterior takes over the row.
std::string kernel_shell(const Model &m, const std::string &entry, const std::string &comment,
                         const std::string &body, int bt, int min_waves, bool regs_l, bool fence_single,
                         bool chrx_loop, int row_doubles, bool call_mode, bool lane_body, bool call_ct_out) {
  // ROW: the lane's LDS row, W3 doubles padded to an odd count (conflict-free ds_read_b64); a
  // generator may ask for more (spare slots it uses itself), odd again
  const int N = m.n_members, W3 = 3 * N, ROW = (row_doubles > 0 ? row_doubles : W3) | 1;
  // Prefetching the next chunk costs W3 doubles of registers next to the W3 marginals; it pays while it costs
  // neither a spill nor a wave.  The register-resident shell (the enumeration kernel) already holds the row: with the
  // prefetch a five-member kernel needs 184 VGPRs = two waves per SIMD, without it 3 fit (its 34 KB of LDS rows
  // allow four workgroups per CU) and 8 M five-member sites take 0.760 instead of 0.796 ms; six members 0.598
  // against 0.612, seven 0.963 against 0.986 (the prefetch spills there); trios and quads keep it (0.376 against
  // 0.387, 0.585 against 0.583: tools/kernel_bench, profiles/r02c/exp_small_peds_3x.txt, exp_sib678.txt).
  int prefetch_max_n = regs_l ? 4 : 10;
  if (const char *e = std::getenv("FAMSEQ_PREFETCH_MAXN")) prefetch_max_n = std::atoi(e);  // tuning aid
  // call_mode: the fused call path's form of the kernel (famseq_bn_call_batch): input packed PLs or fp64
  // rows, outputs GPP / FPP / FGT / status only, arguments behind call_g; no prefetch (the kernel is
  // paced by the logarithms of its outputs).  The plain form carries none of this: its code is unchanged.
  const bool prefetch = N <= prefetch_max_n && !call_mode;
  // Where the next chunk's loads are issued: after the arithmetic (registers are free there, the
  // loads overlap the output phases), or — early — right after this chunk's rows went to LDS (a whole
  // chunk of time to land, but K2 * 4 registers live through the arithmetic).  Early measured no
  // faster on MI355X for 5 members and costs the fence-free variant its registers: off by default.
  int early_max_n = 0;
  if (const char *e = std::getenv("FAMSEQ_PREFETCH_EARLY_MAXN")) early_max_n = std::atoi(e);  // tuning aid
  const bool early = prefetch && N <= early_max_n;
  // How chunks are dealt to workgroups: contiguous ranges (default), or — FAMSEQ_CHUNK_STRIDE=1, an
  // experiment — round robin, so that at any moment the grid reads one contiguous window of each array
  bool strided = false;
  if (const char *e = std::getenv("FAMSEQ_CHUNK_STRIDE")) strided = std::atoi(e) != 0;  // tuning aid
  std::ostringstream s;
  s << "// generated by famseq_amd/csrc for a " << N << "-member pedigree: " << comment << "\n"
    // (the in-process compiler, hiprtc, brings the device built-ins itself and has no include path for the header)
    << "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n#pragma clang fp contract(off)\n"
    << "#define W3 " << W3 << "\n#define ROW " << ROW << "\n#define BT " << bt << "\n"
    // Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e.
    // every barrier would wait for this wave's global stores to reach memory; nothing here hands
    // global data between lanes, so only the LDS counter has to be zero.
    << "#define LDS_BARRIER() asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\")\n"
    << kDiv3Text
    // Staging between global memory and the padded LDS rows.  Element e of the chunk (e = site-in-
    // chunk * W3 + column) lives at LDS index a = (e / W3) * ROW + e % W3.
    // Whole chunks (all but the last of a launch) take a branch-free, fully unrolled walk: quotient
    // and remainder are stepped incrementally, and with no per-element predicate hipcc issues all of
    // a lane's global accesses back to back (behind `if (e < nel)` it emitted load, s_waitcnt
    // vmcnt(0), ds_write per element: W3 serialised HBM round trips per chunk).  The empty asm makes
    // the lane id opaque at each use: otherwise hipcc hoists the staging addresses out of the chunk
    // loop and keeps ~90 registers of them alive, spilled, kernel-wide.
    //   WALK16: lane t handles the PAIRS 2t, 2t + 2 BT, ... (16 B per lane on the global side: at the
    //           2 waves per SIMD these kernels run at, the bare skeleton moves 0.63 of the HBM peak
    //           with 16-B accesses and 0.45 with 8-B ones — tools/io_ceiling.hip).  A chunk starts at
    //           a multiple of BT * W3 * 8 bytes, so pairs are 16-B aligned whenever the array base is
    //           (checked once per launch: v16).  a1 is the pair's second element, possibly in the
    //           next row.  With W3 odd the last step covers the lower half of the lanes only.
    //   WALK8 : lane t handles elements t, t + BT, ... (arrays that are only 8-B aligned)
    //   TAIL  : the partial last chunk, a plain predicated loop
    << "typedef double v2d __attribute__((ext_vector_type(2)));\n"
    << "#define K2 ((W3 + 1) / 2)\n"
    // (every step's index is formed from the lane's own quotient/remainder and per-step constants,
    // not from the previous step's: a stepped index chains the LDS accesses one behind the other)
    // The LDS index of chunk element e = q W3 + r is q ROW + r = e + q (ROW - W3).  With W3 odd and no spare
    // slots ROW = W3 and the LDS image IS the global one (a = e: no quotient, no remainder — the general
    // form's integer division, multiplications and selects, formed anew in every staging pass because the lane id is
    // opaque, were a fifth of a five-member kernel's vector instructions, several of them quarter-rate);
    // otherwise the quotient of the lane's first element, per-step constants and one carry per step.
    << (ROW == W3
            ? "#define WALK8(stmt) { int t_ = tid; asm volatile(\"\" : \"+v\"(t_)); \\\n"
              "  _Pragma(\"unroll\") for (int k = 0; k < W3; ++k) { const int e = t_ + k * BT, a = e; { stmt; } } }\n"
              "#define WALK16(stmt) { int t_ = tid; asm volatile(\"\" : \"+v\"(t_)); \\\n"
              "  _Pragma(\"unroll\") for (int k = 0; k < K2; ++k) { if (k < W3 / 2 || t_ < BT / 2) { \\\n"
              "    const int e = 2 * (t_ + k * BT), a = e, a1 = e + 1; stmt; } } }\n"
            : "#define PAD (ROW - W3)\n"
              "#define WALK8(stmt) { int t_ = tid; asm volatile(\"\" : \"+v\"(t_)); \\\n"
              "  const int q0_ = t_ / W3, r0_ = t_ - q0_ * W3, b0_ = t_ + q0_ * PAD; \\\n"
              "  _Pragma(\"unroll\") for (int k = 0; k < W3; ++k) { \\\n"
              "    const int c_ = r0_ + (k * BT) % W3 >= W3; \\\n"
              "    const int e = t_ + k * BT, a = b0_ + (k * BT + (k * BT) / W3 * PAD) + (c_ ? PAD : 0); { stmt; } } }\n"
              "#define WALK16(stmt) { int t_ = tid; asm volatile(\"\" : \"+v\"(t_)); \\\n"
              "  const int q0_ = (2 * t_) / W3, r0_ = 2 * t_ - q0_ * W3, b0_ = 2 * t_ + q0_ * PAD; \\\n"
              "  _Pragma(\"unroll\") for (int k = 0; k < K2; ++k) { if (k < W3 / 2 || t_ < BT / 2) { \\\n"
              "    const int rr_ = r0_ + (2 * k * BT) % W3, c_ = rr_ >= W3, r = rr_ - (c_ ? W3 : 0); \\\n"
              "    const int e = 2 * (t_ + k * BT), a = b0_ + (2 * k * BT + (2 * k * BT) / W3 * PAD) + (c_ ? PAD : 0); \\\n"
              // (a pair starts at an even element: with W3 even its second element is in the same row)
              "    const int a1 = (W3 % 2 == 0 || r + 1 < W3) ? a + 1 : a + 1 + PAD; (void)r; stmt; } } }\n")
    << "#define TAIL(stmt) { for (int e = tid; e < nel; e += BT) { const int a = (e / W3) * ROW + e % W3; stmt; } }\n"
    // s_io <- G[site0 * W3 ...];  G[site0 * W3 ...] <- s_io;  pre <- next (whole) chunk;  s_io <- pre
    // (the outputs are written once and not read again here: non-temporal stores, +13 % on trios,
    // +1…3 % on the wider pedigrees; non-temporal LOADS were a loss for the lane kernel)
    << "#define STAGE_IN(G) { const double *g_ = (G) + site0 * W3; \\\n"
    << "  if (!whole) { TAIL(s_io[a] = g_[e]) } \\\n"
    << "  else if (v16) { WALK16(const v2d v_ = *(const v2d *)(g_ + e); s_io[a] = v_.x; s_io[a1] = v_.y) } \\\n"
    << "  else { WALK8(s_io[a] = g_[e]) } }\n"
    << "#define STAGE_OUT(G) { double *g_ = (G) + site0 * W3; \\\n"
    << "  if (!whole) { TAIL(g_[e] = s_io[a]) } \\\n"
    << "  else if (v16) { WALK16(v2d v_; v_.x = s_io[a]; v_.y = s_io[a1]; __builtin_nontemporal_store(v_, (v2d *)(g_ + e))) } \\\n"
    << "  else { WALK8(__builtin_nontemporal_store(s_io[a], g_ + e)) } }\n"
    << "#define PREFETCH(G) { const double *g_ = (G) + (site0 + " << (strided ? "(long)gridDim.x * BT" : "BT") << ") * W3; \\\n"
    << "  if (v16) { WALK16(pre[k] = *(const v2d *)(g_ + e); (void)a1) } \\\n"
    << "  else { WALK8(((double *)pre)[k] = g_[e]) } }\n"
    << "#define STAGE_PRE() { \\\n"
    << "  if (v16) { WALK16(s_io[a] = pre[k].x; s_io[a1] = pre[k].y) } \\\n"
    << "  else { WALK8(s_io[a] = ((double *)pre)[k]) } }\n";
  // FAMSEQ_PHASE_CLOCK (measuring aid, call path only): the waves add the cycles of each phase of the chunk loop to counters
  // behind call_g->phase_clk — 0 stage in, 1 single posterior + Phred, 2 GPP out, 3 message passing, 4 Phred of the marginals,
  // 5 FPP / FGT / status out (famseq_bn_call_batch prints the shares).  Without the variable the source is unchanged.
  const bool phase_clock = std::getenv("FAMSEQ_PHASE_CLOCK") != nullptr;  // (plain kernels: counters in a module global, fs_phase_clk)
  // How much of the output work the scheduler sees at once (kCallHelpers): logarithms of two members or of one per basic block,
  // the stage-out walk with the row width as a constant or as read from the arguments.  More at once = more registers.
  int phred_group = 2;
  const bool ct_out = call_ct_out;
  if (const char *e = std::getenv("FAMSEQ_CALL_PHRED_GROUP")) phred_group = std::atoi(e) == 1 ? 1 : 2;  // tuning aid
  auto PH = [&](int i) { return phase_clock ? "    PH(" + std::to_string(i) + ");\n" : std::string(); };
  if (phase_clock && !call_mode)
    s << "__device__ unsigned long long fs_phase_clk[8];\n"
         "#define PH(i) { const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_##i += t_ - ph_last_; ph_last_ = t_; }\n"
         "#define PH_FLUSH(i) atomicAdd(&fs_phase_clk[i], ph_acc_##i)\n";
  if (call_mode)
    s << "#define NMEM " << N << "\n#define NSEQ_CT " << std::max(1, (int)std::count(m.sequenced.begin(), m.sequenced.end(), 1)) << "\n"
      // (measured, ns per 1 M sites, hoisted / formed again: sum-product form 5 members 125 / 133, trio 70 / 77, quad 96 / 103, ten 252 with
      // the leaner walk / 249; enumeration form trio 66 / 88, quad 105 / 108, five members 162 / 143)
      << "#define FS_OPAQUE_LANE " << (N >= (entry == "famseq_elim" ? 7 : 5) ? 1 : 0) << "\n"
      << "#define FS_PHRED_GROUP " << phred_group << "\n#define FS_CT_OUT " << (ct_out ? 1 : 2) << "\n"
      << (phase_clock ? with_phase_clock(kCallHelpers) : kCallHelpers)
      << "#define STAGE_IN_ANY() { if (packed_in) { STAGE_IN_PL(); } else { STAGE_IN(lk_g); } }\n";
  // flat staging of the packed PLs (kCallFlat): the sum-product kernel's call form; FAMSEQ_CALL_FLAT=0 (tuning aid) keeps the item walk
  bool flat_pl = call_mode && entry == "famseq_elim" && bt % 8 == 0;  // (BT * n_seq * 6 bytes are whole 16-byte pieces)
  if (const char *e = std::getenv("FAMSEQ_CALL_FLAT")) flat_pl = flat_pl && std::atoi(e) != 0;
  int lut_lds = 1024;
  if (const char *e = std::getenv("FAMSEQ_CALL_LUT_LDS")) {  // tuning aid; a power of two (the staging tests "any of the three beyond it" on their OR)
    lut_lds = 1;
    while (lut_lds * 2 <= std::min(4096, std::atoi(e))) lut_lds *= 2;
  }
  if (flat_pl) s << "#define FS_LUT_LDS " << lut_lds << "\n" << kCallFlat;

  if (!regs_l)
    for (int p = 0; p < N; ++p)
      for (int gt = 0; gt < 3; ++gt) s << "#define l" << p << "_" << gt << " lrow[" << 3 * p + gt << "]\n";
  s << "extern \"C\" __global__ __launch_bounds__(BT, " << min_waves << ") void " << entry
    << "(const double *__restrict__ lk_g,\n"
    << "    const unsigned char *__restrict__ flags_g, double *__restrict__ post_g, double *__restrict__ single_g,\n"
    << "    unsigned char *__restrict__ status_g, long n_sites, const double *__restrict__ tc_g, double lc" << (call_mode ? kCallArgs : "") << ") {\n"
    << (flat_pl ? "  __shared__ __attribute__((aligned(16))) double s_io[BT * ROW];  // one padded row per lane: conflict-free ds_read_b64\n"
                : "  __shared__ double s_io[BT * ROW];  // one padded row per lane: conflict-free ds_read_b64\n")
    << "  __shared__ double s_tc[432];\n"
    << "  const int tid = threadIdx.x;\n"
    << "  for (int i = tid; i < 432; i += BT) s_tc[i] = tc_g[i];\n"
    << "  const long chunks = (n_sites + BT - 1) / BT;\n"
    << (strided ? "  const long c_lo = blockIdx.x, c_hi = chunks;\n"
                // contiguous ranges of q or q + 1 chunks: every workgroup of the grid has work (ceil(chunks / grid) each left
                // a sixth of them idle at BASELINE's 1 M five-member sites: 15,625 chunks over 3,072 resident workgroups,
                // 0.0705 -> 0.0662 ms)
                : "  const long q_wg = chunks / gridDim.x, r_wg = chunks - q_wg * gridDim.x;\n"
                  "  const long c_lo = (long)blockIdx.x * q_wg + (blockIdx.x < r_wg ? blockIdx.x : r_wg), c_hi = c_lo + q_wg + (blockIdx.x < r_wg ? 1 : 0);\n")
    << "  const double kNaN = __builtin_nan(\"\");\n"
    << "  double *row = s_io + tid * ROW;\n"
    // (regs_l = false) volatile forces a fresh LDS read per use.  The address space is spelled out:
    // hipcc does not infer it for volatile accesses and would emit flat_load instead of ds_read.
    << "  typedef const volatile __attribute__((address_space(3))) double lds_cvd;\n"
    << "  lds_cvd *lrow = (lds_cvd *)row;\n"
    << "  v2d pre[K2];  // (prefetch) this lane's share of the NEXT chunk, loaded ahead\n"
    << "  const bool v16 = (((unsigned long)lk_g | (unsigned long)post_g | (unsigned long)single_g) & 15) == 0;\n"
    << "  bool have_pre = false;\n"
    << (call_mode ? "  const bool packed_in = call_g->pl != nullptr;  // fed with packed PLs (else fp64 rows)\n"
                    "  __shared__ int s_col[NMEM];  // member -> VCF column or -1\n"
                    // member -> slot of the output row, the same for every lane and chunk.  Small pedigrees keep them in scalar registers
                    // (per use an LDS read is a latency the trio's short phases feel: 0.063 against 0.082 ms per 1 M sites); from seven
                    // members on they live in LDS (ten more live SGPRs at ten members: every variant spills, 0.253 against 0.213 ms)
                    "#if NMEM < 7\n  int slot_r_[NMEM];\n#pragma unroll\n  for (int i = 0; i < NMEM; ++i) slot_r_[i] = call_g->slot[i];\n"
                    "#else\n  __shared__ int slot_r_[NMEM];\n  for (int i = tid; i < NMEM; i += BT) slot_r_[i] = call_g->slot[i];\n#endif\n"
                    "  __shared__ signed char s_fgt[BT * NMEM];  // arg-max genotype of every member of every site of the chunk\n"
                    "  __shared__ __attribute__((aligned(16))) double s_lt[258];  // fs_phred's table: one 16-byte LDS read per logarithm\n"
                    "  for (int i = tid; i < 258; i += BT) s_lt[i] = fs_logtab[i];\n"
                    "  for (int i = tid; i < NMEM; i += BT) s_col[i] = call_g->col[i];\n"
                  : "")
    << (flat_pl ? "  __shared__ double s_lut[FS_LUT_LDS];  // pow(10, -k / 10), k < FS_LUT_LDS\n"
                  "  if (packed_in) for (int i = tid; i < FS_LUT_LDS; i += BT) s_lut[i] = call_g->lut[i];\n"
                  "  const bool flat_ok = packed_in && ((unsigned long)call_g->pl & 15) == 0;\n"
                  "  fs_v4u praw[RAW_K];  // the NEXT chunk's packed PLs, fetched ahead\n"
                  "  bool have_raw = false;\n"

                  : "")
    << (phase_clock ? "  unsigned long long ph_last_ = 0, ph_acc_0 = 0, ph_acc_1 = 0, ph_acc_2 = 0, ph_acc_3 = 0, ph_acc_4 = 0, ph_acc_5 = 0, ph_acc_6 = 0, ph_acc_7 = 0;\n" : "")
    << (strided ? "  for (long ch = c_lo; ch < c_hi; ch += gridDim.x) {\n" : "  for (long ch = c_lo; ch < c_hi; ++ch) {\n")
    << "    const long site0 = ch * BT;\n"
    << "    const int ns = n_sites - site0 < BT ? (int)(n_sites - site0) : BT;\n"
    << "    const int nel = ns * W3;\n"
    << "    const bool whole = ns == BT;\n"
    << (phase_clock ? "    ph_last_ = __builtin_readcyclecounter();\n" : "")
    << "    LDS_BARRIER();\n";
  if (prefetch) {
    // the next chunk's rows were requested during the previous chunk's output phases
    s << "    if (have_pre) { STAGE_PRE(); } else { STAGE_IN(lk_g); }\n";
    if (early)
      s << "    have_pre = " << (strided ? "ch + gridDim.x < c_hi && (ch + gridDim.x + 1) * BT <= n_sites" : "ch + 1 < c_hi && site0 + 2 * BT <= n_sites") << ";  // only whole chunks are prefetched\n"
        << "    if (have_pre) { PREFETCH(lk_g); }\n";
  } else {
    s << (flat_pl ? "    if (flat_ok && whole) { if (!have_raw) { PL_FETCH(site0); } STAGE_IN_PL_FLAT(); } else { STAGE_IN_ANY(); }\n"
                  : (call_mode ? "    STAGE_IN_ANY();\n" : "    STAGE_IN(lk_g);\n"));
  }
  s << "    LDS_BARRIER();\n" << PH(0)
    << "    const int fl = (tid < ns && flags_g) ? (flags_g[site0 + tid] & 3) : 0;\n"
    << "    const double *tcf = s_tc + fl * 108;\n"
    << "    bool single_fail = false, full = false, bn_fail = false;\n";
  auto single_pass = [&](bool flags_pass, bool store, const char *dst = "row") {
    s << single_posterior_statements(m, flags_pass, store, fence_single, dst);
  };
  if (regs_l) {
    for (int p = 0; p < N; ++p)
      for (int gt = 0; gt < 3; ++gt) s << "    const double l" << p << "_" << gt << " = row[" << 3 * p + gt << "];\n";
    s << "    LDS_BARRIER();  // every lane holds its row in registers: the rows become the output stage\n";
    single_pass(true, true);
    s << "    if (single_fail) for (int k = 0; k < W3; ++k) row[k] = kNaN;\n"
      << (call_mode ? "    ROW_TO_CALL();  // the single posterior as printed (GPP) and its arg-max (FGT of shortcut sites)\n" : "")
      << "    LDS_BARRIER();\n" << PH(1)
      << (call_mode ? "    if (call_g->gpp) { STAGE_OUT_CALL(call_g->gpp); }\n" : "    if (single_g) { STAGE_OUT(single_g); }\n")
      << "    LDS_BARRIER();  // single rows are stored; sites that need the full computation overwrite theirs\n" << PH(2);
    if (chrx_loop)
      // (as for the sum-product body below) the children's transmission entries depend on the site's chrX bit only: the
      // body reads them through the wave-uniform pointer tcx (scalar loads) and runs once per chrX value present in the wave
      s << "    {\n      const int chrx_ = fl >> 1;\n"
        << "#pragma unroll 1\n"
        << "      for (int x_ = 0; x_ < 2; ++x_) {\n"
        << "        const bool mine_ = full && !single_fail && chrx_ == x_;\n"
        << "        if (__builtin_amdgcn_ballot_w64(mine_) == 0) continue;\n"
        << "        const double *tcx = tc_g + x_ * 216;\n"
        << "        if (mine_) {\n";
    else
      s << "    if (full && !single_fail) {\n";
    s << "      const double *lg = lk_g + (site0 + (tid < ns ? tid : 0)) * W3;  // this lane's row in global memory (fp64 input only)\n"
      << "      (void)lg;\n"
      << body
      << "      if (bn_fail) for (int k = 0; k < W3; ++k) row[k] = kNaN;\n"
      << (call_mode ? "      ROW_TO_CALL();  // the BN posterior as printed (FPP) and the genotype call\n" : "")
      << (chrx_loop ? "        }\n      }\n    }\n" : "    }\n");
    if (prefetch && !early)
      s << "    have_pre = " << (strided ? "ch + gridDim.x < c_hi && (ch + gridDim.x + 1) * BT <= n_sites" : "ch + 1 < c_hi && site0 + 2 * BT <= n_sites") << ";  // only whole chunks are prefetched\n"
        << "    if (have_pre) { PREFETCH(lk_g); }\n";
    s << (flat_pl ? "    have_raw = flat_ok && ch + 1 < c_hi && site0 + 2 * BT <= n_sites;  // the next chunk, if it is a whole one: its packed PLs land during the output phases\n"
                  "    if (have_raw) { PL_FETCH(site0 + BT); }\n" : "")
      << "    LDS_BARRIER();\n" << PH(3)
      << (call_mode ? "    if (call_g->fpp) { STAGE_OUT_CALL(call_g->fpp); }\n" + PH(7) + "    if (call_g->fgt) { STAGE_FGT(call_g->fgt); }\n"
                    : std::string("    STAGE_OUT(post_g);\n"))
      << "    if (status_g && tid < ns) status_g[site0 + tid] = single_fail ? 1 : (!full ? 0x80 : (bn_fail ? 2 : 0));\n"
      << PH(5) << (phase_clock ? std::string("  }\n  if ((tid & 63) == 0") + (call_mode ? " && call_g->phase_clk" : "") + ") { PH_FLUSH(0); PH_FLUSH(1); PH_FLUSH(2); PH_FLUSH(3); PH_FLUSH(4); PH_FLUSH(5); PH_FLUSH(6); PH_FLUSH(7); }\n}\n" : std::string("  }\n}\n"));
  } else {
    // Outputs are staged through the same LDS rows as the input (coalesced 8 B/lane stores).
    // Writing each lane's row straight from registers was measured 20 % slower on MI355X
    // (64 partial-line requests per store instruction), so the extra barriers stay.
    single_pass(true, false);
    s << PH(6) << "    double q[W3];\n";
    if (chrx_loop) {
      // The transmission entries depend on the site's chrX bit only.  The body reads them through a
      // wave-uniform pointer (scalar loads: no LDS traffic, no VGPRs) and runs once per chrX value
      // present in the wave, with the lanes of that value active — one pass in practice.
      s << "    {\n      const int chrx_ = fl >> 1;\n"
        << "#pragma unroll 1\n"
        << "      for (int x_ = 0; x_ < 2; ++x_) {\n"
        << "        const bool mine_ = full && !single_fail && chrx_ == x_;\n"
        << "        if (__builtin_amdgcn_ballot_w64(mine_) == 0) continue;\n"
        << "        const double *tcx = tc_g + x_ * 216;\n"
        << "        if (mine_) {\n"
        << body << "        }\n      }\n    }\n";
    } else {
      s << "    if (full && !single_fail) {\n"
        << (lane_body ? "      double *srow = row + W3;  // the enumeration's scratch slots, behind the likelihoods\n"
                        "      const double *lg = lk_g + (site0 + (tid < ns ? tid : 0)) * W3;  // this lane's row in global memory (fp64 input only)\n"
                        "      (void)srow; (void)lg;\n" : "")
        << body << "    }\n";
    }
    if (prefetch && !early)
      // software prefetch: issue the next chunk's loads now; they stay in flight while this
      // chunk's two output phases run (the barriers below do not wait for vmcnt)
      s << "    have_pre = " << (strided ? "ch + gridDim.x < c_hi && (ch + gridDim.x + 1) * BT <= n_sites" : "ch + 1 < c_hi && site0 + 2 * BT <= n_sites") << ";  // only whole chunks are prefetched\n"
      << "    if (have_pre) { PREFETCH(lk_g); }\n";
    s << PH(3) << (flat_pl ? "    have_raw = flat_ok && ch + 1 < c_hi && site0 + 2 * BT <= n_sites;  // the next chunk, if it is a whole one: its packed PLs land during the output phases\n"
                  "    if (have_raw) { PL_FETCH(site0 + BT); }\n" : "");
    if (call_mode) {
      // the call path: the single posterior stays in registers, and what is printed of it (GPP; the arg-max is the FGT of
      // shortcut sites) goes to the row in output order — every likelihood has been read by then
      s << "    double u_[W3];\n";
      single_pass(false, true, "u_");
      s << "    if (single_fail) for (int k = 0; k < W3; ++k) u_[k] = kNaN;\n    PUT_CALL(u_);\n";
    } else {
      single_pass(false, true);  // now the single posterior may take the row over
      s << "    if (single_fail) for (int k = 0; k < W3; ++k) row[k] = kNaN;\n";
    }
    s << "    LDS_BARRIER();\n" << PH(1)
      << (call_mode ? "    if (call_g->gpp) { STAGE_OUT_CALL(call_g->gpp); }\n" : "    if (single_g) { STAGE_OUT(single_g); }\n")
      << "    LDS_BARRIER();  // single rows are stored; sites that ran the full computation overwrite theirs\n" << PH(2)
      << "    if (full && !single_fail) {\n"
      << (call_mode ? "      if (bn_fail) for (int k = 0; k < W3; ++k) q[k] = kNaN;\n"
                      "      PUT_CALL(q);  // the BN posterior as printed (FPP) and the genotype call, straight from the registers\n"
                    : "#pragma unroll\n      for (int k = 0; k < W3; ++k) row[k] = bn_fail ? kNaN : q[k];\n")
      << "    }\n"
      << "    LDS_BARRIER();\n" << PH(4)
      << (call_mode ? "    if (call_g->fpp) { STAGE_OUT_CALL(call_g->fpp); }\n" + PH(7) + "    if (call_g->fgt) { STAGE_FGT(call_g->fgt); }\n"
                    : std::string("    STAGE_OUT(post_g);\n"))
      << "    if (status_g && tid < ns) status_g[site0 + tid] = single_fail ? 1 : (!full ? 0x80 : (bn_fail ? 2 : 0));\n"
      << PH(5) << (phase_clock ? std::string("  }\n  if ((tid & 63) == 0") + (call_mode ? " && call_g->phase_clk" : "") + ") { PH_FLUSH(0); PH_FLUSH(1); PH_FLUSH(2); PH_FLUSH(3); PH_FLUSH(4); PH_FLUSH(5); PH_FLUSH(6); PH_FLUSH(7); }\n}\n" : std::string("  }\n}\n"));
  }
  return s.str();
}

namespace {

// The shell without LDS staging (variants 8..11: the widest pedigrees): a lane reads its site's row straight from global
// memory into registers and stores its single posterior and marginal rows straight back — 8 bytes per lane and
// instruction, a cache line per lane.  What the staged shell buys with its LDS rows (coalesced 16-byte accesses) costs it
// the CU's LDS: 3N doubles per lane leave two waves per CU at 48 members and nothing beyond about a hundred; this form
// needs 3.4 KB of LDS (the factor tables) whatever N is, runs four waves per CU, and has no barrier after the first.
std::string direct_shell(const Model &m, const std::string &comment, const std::string &body, int bt, bool fence_single, bool chrx_loop,
                         bool lean, int lds_from) {
  // lds_from: members lds_from .. N-1 keep their likelihoods in a per-lane LDS row (3 doubles each, odd stride) and are read
  // from there at each use; the others live in registers.  The last members' local factors have the longest live ranges (the
  // upward pass of the first marginal touches every member, and member p's factor is needed again at its own marginal).
  const int N = m.n_members, W3 = 3 * N, n_lds = lds_from < N ? N - lds_from : 0, LP = (3 * n_lds) | 1;
  std::ostringstream s;
  s << "// generated by famseq_amd/csrc for a " << N << "-member pedigree: " << comment << "\n"
    << "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n#pragma clang fp contract(off)\n"
    << "#define W3 " << W3 << "\n#define BT " << bt << "\n"
    << "#define LDS_BARRIER() asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\")\n"
    << kDiv3Text;
  if (lean)  // a likelihood is read from the lane's row in global memory at each use (volatile: never kept in a register)
    for (int p = 0; p < N; ++p)
      for (int gt = 0; gt < 3; ++gt) s << "#define l" << p << "_" << gt << " lgv[" << 3 * p + gt << "]\n";
  else
    for (int p = lds_from; p < N; ++p)
      for (int gt = 0; gt < 3; ++gt) s << "#define l" << p << "_" << gt << " lrow[" << 3 * (p - lds_from) + gt << "]\n";
  int min_waves = 1;
  if (const char *e = std::getenv("FAMSEQ_ELIM_MINWAVES")) min_waves = std::max(1, std::atoi(e));  // tuning aid
  s << "extern \"C\" __global__ __launch_bounds__(BT, " << min_waves << ") void famseq_elim(const double *__restrict__ lk_g,\n"
    << "    const unsigned char *__restrict__ flags_g, double *__restrict__ post_g, double *__restrict__ single_g,\n"
    << "    unsigned char *__restrict__ status_g, long n_sites, const double *__restrict__ tc_g, double lc) {\n"
    << "  __shared__ double s_tc[432];\n"
    << "  const int tid = threadIdx.x;\n"
    << "  for (int i = tid; i < 432; i += BT) s_tc[i] = tc_g[i];\n";
  if (n_lds > 0)
    s << "  __shared__ double s_l[BT * " << LP << "];  // the last " << n_lds << " members' likelihoods, one padded row per lane\n"
      << "  typedef const volatile __attribute__((address_space(3))) double lds_cvd;\n"
      << "  double *lw = s_l + tid * " << LP << ";\n  lds_cvd *lrow = (lds_cvd *)lw;\n";
  s << "  LDS_BARRIER();\n"
    << "  const long chunks = (n_sites + BT - 1) / BT;\n"
    << "  const long q_wg = chunks / gridDim.x, r_wg = chunks - q_wg * gridDim.x;\n"
    << "  const long c_lo = (long)blockIdx.x * q_wg + (blockIdx.x < r_wg ? blockIdx.x : r_wg), c_hi = c_lo + q_wg + (blockIdx.x < r_wg ? 1 : 0);\n"
    << "  const double kNaN = __builtin_nan(\"\");\n"
    << "  for (long ch = c_lo; ch < c_hi; ++ch) {\n"
    // a lane beyond the batch's end works on the last site: the same values to the same addresses as that site's own lane
    << "    const long site = ch * BT + tid < n_sites ? ch * BT + tid : n_sites - 1;\n"
    << "    const double *lg = lk_g + site * W3;\n"
    << "    double *pg = post_g + site * W3;\n"
    << "    double *row = single_g ? single_g + site * W3 : pg;  // where the single posterior goes\n"
    << "    const int fl = flags_g ? (flags_g[site] & 3) : 0;\n"
    << "    const double *tcf = s_tc + fl * 108;\n"
    << "    bool single_fail = false, full = false, bn_fail = false;\n";
  if (lean)
    s << "    typedef const volatile __attribute__((address_space(1))) double glb_cvd;\n    glb_cvd *lgv = (glb_cvd *)lg;\n";
  else {
    for (int p = 0; p < std::min(N, lds_from); ++p)
      for (int gt = 0; gt < 3; ++gt) s << "    const double l" << p << "_" << gt << " = lg[" << 3 * p + gt << "];\n";
    for (int k = 3 * std::min(N, lds_from); k < W3; ++k) s << "    lw[" << k - 3 * lds_from << "] = lg[" << k << "];\n";
  }
  s << single_posterior_statements(m, true, true, fence_single)
    << "    if (single_fail) {\n#pragma unroll 1\n      for (int k = 0; k < W3; ++k) row[k] = kNaN;\n    }\n"
    // a site that does not take the full computation: its posterior IS the single posterior (family.cpp:793-878) or NaN
    << "    if (single_g && !(full && !single_fail)) {\n#pragma unroll 1\n      for (int k = 0; k < W3; ++k) pg[k] = row[k];\n    }\n";
  if (chrx_loop)
    s << "    {\n      const int chrx_ = fl >> 1;\n"
      << "#pragma unroll 1\n"
      << "      for (int x_ = 0; x_ < 2; ++x_) {\n"
      << "        const bool mine_ = full && !single_fail && chrx_ == x_;\n"
      << "        if (__builtin_amdgcn_ballot_w64(mine_) == 0) continue;\n"
      << "        const double *tcx = tc_g + x_ * 216;\n"
      << "        if (mine_) {\n";
  else
    s << "    if (full && !single_fail) {\n";
  s << body
    << "      if (bn_fail) {\n#pragma unroll 1\n        for (int k = 0; k < W3; ++k) pg[k] = kNaN;\n      }\n"
    << (chrx_loop ? "        }\n      }\n    }\n" : "    }\n")
    << "    if (status_g) status_g[site] = single_fail ? 1 : (!full ? 0x80 : (bn_fail ? 2 : 0));\n"
    << "  }\n}\n";
  return s.str();
}

}  // namespace

std::string elim_source(const Model &m, int variant, bool call_mode) {
  Graph g;
  std::string why;
  if (!build_graph(m, g, &why)) throw std::runtime_error("elimination engine: " + why);
  if (variant >= 8 && !call_mode) {  // no LDS staging: see direct_shell
    const int f = variant & 3;
    bool lean = false;
    if (const char *e = std::getenv("FAMSEQ_ELIM_LEAN")) lean = std::atoi(e) != 0;  // tuning aid
    // members whose likelihoods live in the lane's LDS row rather than in registers: at four waves per CU a lane has 73 doubles
    // of LDS, 24 members.  Pays from the mid-fifties on, where the scratch it spares outweighs the LDS latency it adds (2 M
    // sites: 48 members 2.37-2.50 -> 2.55-2.59 ms, 64: 4.72-4.91 -> 4.25-4.44, 96: 10.3-10.5 -> 9.07; scratch 1276 -> 956 B at 64)
    int n_lds = m.n_members >= 56 ? 24 : 0;
    if (const char *e = std::getenv("FAMSEQ_ELIM_LDSL")) n_lds = std::max(0, std::min(std::atoi(e), m.n_members));  // tuning aid
    const int lds_from = lean ? m.n_members : m.n_members - n_lds;
    return direct_shell(m,
                        "exact sum-product over " + std::to_string(g.fam.size()) + " nuclear families" +
                            (g.cut.empty() ? "" : ", conditioned on " + std::to_string(g.cut.size()) + " member(s)") + ", variant " +
                            std::to_string(variant) + " (rows straight from and to global memory)",
                        Emitter(m, g, f < 2 ? f : 2, /*scalar_t=*/f >= 1, "pg", lean, lds_from).body(), elim_block_threads(m, false), f >= 3,
                        /*chrx_loop=*/f >= 1, lean, lds_from);
  }
  const int bt = elim_block_threads(m, call_mode);
  int min_waves = call_mode && m.n_members <= 10 ? 2 : 1;
  if (const char *e = std::getenv("FAMSEQ_ELIM_MINWAVES")) min_waves = std::atoi(e);  // tuning aid
  // From variant 1 on the transmission tables are read through scalar loads (measured: +9 % at 10
  // members where registers are tight, -7 % on the fence-free 5-member kernel, which keeps the LDS table).
  // variant 0: no compiler fences (most overlap between the message blocks; fits small pedigrees),
  //         1: a fence after every family->member message, 2: also after local factors and child
  //         summaries, 3: also between the members of the single posterior
  // Where the likelihoods live during the message passing: re-read from the lane's LDS row at each use (short live ranges:
  // what the narrow pedigrees' kernels want, they run at two or more waves per SIMD), or — registers-first — read once into
  // registers, the row then being the output stage (no q[] array, a tenth of the LDS reads, all of them issued together).
  bool regs_l = call_mode ? std::getenv("FAMSEQ_ELIM_CALL_REGS") != nullptr : variant >= 4;  // (the call path: r = 0 unless the tuning aid says otherwise)
  if (const char *e = std::getenv("FAMSEQ_ELIM_REGS")) regs_l = std::atoi(e) != 0 && !call_mode;  // tuning aid
  const bool ct_out = !(call_mode && (variant & 4));
  variant &= 3;  // the fence level
  const std::string what = "exact sum-product over " + std::to_string(g.fam.size()) + " nuclear families" +
                           (g.cut.empty() ? "" : ", conditioned on " + std::to_string(g.cut.size()) + " member(s)") + ", variant " +
                           std::to_string(variant + (call_mode ? (ct_out ? 0 : 4) : (regs_l ? 4 : 0))) + (regs_l ? " (likelihoods in registers)" : "") +
                           (call_mode ? ", call path" : "");
  return kernel_shell(m, "famseq_elim", what,
                      Emitter(m, g, variant < 2 ? variant : 2, /*scalar_t=*/variant >= 1, regs_l ? "row" : "q").body(), bt, min_waves,
                      regs_l, variant >= 3, /*chrx_loop=*/variant >= 1, 0, call_mode, /*lane_body=*/false, ct_out);
}

}  // namespace famseq
