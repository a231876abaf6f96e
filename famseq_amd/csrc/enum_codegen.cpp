// enum_codegen.cpp — generates the lane-per-site 3^N enumeration kernel for one pedigree.
//
// Same computation as bn_enum_kernel (bn_kernel.hip) and as the reference's odometer
// (/root/reference/src/family.cpp:894-941, :1014-1106): every one of the 3^N joint genotype
// weights 1e7 * prod_m f_m is formed and added to the marginals.  What changes is the mapping:
// here ONE LANE owns a whole site, so there is no cross-lane reduction, no workgroup barrier in
// the hot loop and no per-step table traffic — everything a site needs lives in registers.
//   * "outer" members (ancestors) are walked by ordinary nested loops; the loop digits are the
//     same in every lane, so their table offsets are scalar;
//   * the last <= 6 members in descent order ("unrolled" set U, closed under children) form a
//     fully unrolled block of 3^|U| configurations whose factor tables (indexed by the unrolled
//     parents' digits) are rebuilt in registers once per outer step;
//   * inside the block prefix products are shared level by level; the deepest 2-3 levels (the
//     "super-leaf") are multiplied once per outer step into a table W, and every configuration
//     costs exactly one FMA, prefix * W[c] into the accumulator of its super-leaf digits c; those
//     3^sl accumulators run over the whole site and are summed into the super-leaf members'
//     marginals at the end; the other unrolled members receive block sums (Q) per prefix, the
//     looped ones block totals kept in the lane's LDS row.
// The generic team-per-site kernel remains the fallback (small batches, no compiler at run time).
#include "enum_codegen.h"

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "elim_codegen.h"

namespace famseq {

namespace {

std::string num(int x) { return std::to_string(x); }

struct Shape {
  int N = 0;
  std::vector<int> outer, unrolled;  // both in parents-before-children order
  std::vector<int> upos;             // member -> level in `unrolled` or -1
};

int kind_of(const Model &m, int p) {
  const bool male = m.gender[p] == 1;
  return m.mother[p] < 0 ? (male ? 0 : 1) : (male ? 2 : 3);
}

Shape choose_shape(const Model &m, int cap) {
  const int N = m.n_members;
  std::vector<std::vector<int>> kids(N);
  for (int i = 0; i < N; ++i)
    if (m.mother[i] >= 0) {
      kids[m.mother[i]].push_back(i);
      kids[m.father[i]].push_back(i);
    }
  // depth = longest chain of ancestors; sorting by it gives a parents-before-children order
  std::vector<int> depth(N, 0);
  for (int pass = 0; pass < N; ++pass)
    for (int i = 0; i < N; ++i)
      if (m.mother[i] >= 0) depth[i] = std::max(depth[i], 1 + std::max(depth[m.mother[i]], depth[m.father[i]]));
  std::vector<int> order(N);
  for (int i = 0; i < N; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
  std::vector<char> inU(N, 0);
  int nu = 0;
  // (four members at most: 60 — two founders and two children of both are 3 + 3 + 27 + 27 — so that a quad is ONE unrolled block of
  // 81 configurations instead of a three-step loop over 27: 0.556 -> 0.512 ms per 8 M sites; five members all unrolled need 120 and
  // run at half the waves, 0.70 -> 0.88)
  int table_budget = N <= 4 ? 60 : 48;
  if (const char *e = std::getenv("FAMSEQ_LANE_TABLE_BUDGET")) table_budget = std::atoi(e);  // tuning aid
  auto table_doubles = [&]() {
    int t = 0;
    for (int i = 0; i < N; ++i)
      if (inU[i]) {
        int e = 3;
        if (m.mother[i] >= 0) e *= (inU[m.mother[i]] ? 3 : 1) * (inU[m.father[i]] ? 3 : 1);
        t += e;
      }
    return t;
  };
  // children first: a member may join U only when all its children already have
  for (int k = N - 1; k >= 0 && nu < cap; --k) {
    const int i = order[k];
    bool ok = true;
    for (int c : kids[i]) ok = ok && inU[c];
    if (!ok) continue;
    inU[i] = 1;
    if (table_doubles() > table_budget) {  // register budget for the block's factor tables
      inU[i] = 0;
      continue;
    }
    ++nu;
  }
  Shape s;
  s.N = N;
  s.upos.assign(N, -1);
  for (int i : order)
    if (!inU[i]) s.outer.push_back(i);
  // Unrolled members in depth-first order (parents before children, each subtree contiguous):
  // the block sums below a level then depend on as few upper digits as possible.
  std::function<void(int)> place = [&](int i) {
    if (!inU[i] || s.upos[i] >= 0) return;
    if (m.mother[i] >= 0)
      for (int par : {m.mother[i], m.father[i]})
        if (inU[par] && s.upos[par] < 0) return;  // placed later, from its other parent
    s.upos[i] = (int)s.unrolled.size();
    s.unrolled.push_back(i);
    for (int c : kids[i]) place(c);
  };
  for (int i : order) place(i);
  return s;
}

class Gen {
 public:
  // fixed: the `fixed` outermost looped members do not loop — their digits come from the lane's
  // position in its group (fx0, fx1, ...: lanes-per-site mode, 3^fixed lanes share a site)
  // late: the small-pedigree form — the lane's row keeps the input likelihoods (the shell reads them from
  // LDS and turns them into the single posterior only after this body), so the body's scratch slots live
  // behind them (srow = row + W3) and the normalised marginals go to registers q[] instead of the row
  // scalar_t: the children's transmission entries come from tcx[] (wave-uniform pointer: scalar loads, no LDS
  // instruction, no VGPR) and the founders' priors are folded into their likelihood slots once per site;
  // prefetch 1: the entries the innermost loop's tables need are loaded one step ahead (loop-carried SGPRs), 2: so are
  // that loop's LDS reads — everything the next step's table statements wait for is in flight during this step's block
  Gen(const Model &m, const Shape &s, int row_len, int fixed = 0, bool late = false, bool scalar_t = false, int prefetch = 0)
      : m_(m), s_(s), nu_((int)s.unrolled.size()), row_len_(row_len), fixed_(fixed), S_(late ? "srow" : "row"),
        O_(late ? "q" : "row"), outer_(s.outer), st_(scalar_t && !s.outer.empty()),
        pre_(scalar_t && fixed < (int)s.outer.size() ? prefetch : 0) {}

  // Lanes-per-site mode, last step (the group's first lane, after the column sums): normalise,
  // failure rule (family.cpp:943-954).
  std::string reduce_body() const {
    std::ostringstream o;
    o << "#pragma unroll 1\n      for (int k = 0; k < W3; k += 3) {\n"
      << "        const double t0 = row[k], t1 = row[k + 1], t2 = row[k + 2];\n"
      << "        const double s = (t0 + t1) + t2; if (s <= 0) bn_fail = true;\n"
      << "        if (s < 1e-290) { asm volatile(\"\" ::: \"memory\"); row[k] = t0 / s; row[k + 1] = t1 / s; row[k + 2] = t2 / s; }\n"
      << "        else { const double r = 1.0 / s; row[k] = t0 * r; row[k + 1] = t1 * r; row[k + 2] = t2 * r; }\n      }\n";
    return o.str();
  }

  // The per-step tables are plain expressions of the loop digits, so the compiler hoists each one
  // to the outermost loop whose digit it mentions.  Put the member whose digit feeds the most
  // table entries outermost (among members of equal depth, parents still enclose children).
  void order_outer_loops() {
    const int N = s_.N;
    std::vector<int> depth(N, 0), cost(N, 0);
    for (int pass = 0; pass < N; ++pass)
      for (int i = 0; i < N; ++i)
        if (m_.mother[i] >= 0) depth[i] = std::max(depth[i], 1 + std::max(depth[m_.mother[i]], depth[m_.father[i]]));
    std::vector<std::vector<char>> feeds(nu_, std::vector<char>(N, 0));  // outer parents of level k
    for (int k = 0; k < nu_; ++k) {
      const int p = s_.unrolled[k];
      if (m_.mother[p] < 0) continue;
      for (int par : {m_.mother[p], m_.father[p]})
        if (s_.upos[par] < 0) feeds[k][par] = 1;
    }
    for (int o : outer_) {
      bool below = false;  // does any level >= k depend on o?
      for (int k = nu_ - 1; k >= 0; --k) {
        const int p = s_.unrolled[k];
        if (feeds[k][o]) {
          int e = 3;
          if (m_.mother[p] >= 0) e *= (s_.upos[m_.mother[p]] >= 0 ? 3 : 1) * (s_.upos[m_.father[p]] >= 0 ? 3 : 1);
          cost[o] += e;
          below = true;
        }
        if (below) cost[o] += pow3((int)dep_[k].size());                                   // Q<k>
        if (below && sl_ >= 2 && k == nu_ - sl_) cost[o] += pow3(sl_ + (int)dep_[k].size()) * 3 / 2;  // W, WQ, X
      }
    }
    std::stable_sort(outer_.begin(), outer_.end(), [&](int a, int b) {
      if (depth[a] != depth[b]) return depth[a] < depth[b];
      return cost[a] > cost[b];
    });
  }

  std::string body() {
    if (const char *e = std::getenv("FAMSEQ_LANE_PIN")) pin_style_ = std::atoi(e);  // tuning aid
    compute_deps();
    choose_superleaf();
    order_outer_loops();
    const int no = (int)outer_.size(), row_len = row_len_;
    l_in_lds_ = 6 * no <= row_len;
    o_ << "      // outer (looped) members:";
    for (int p : outer_) o_ << " " << p;
    o_ << " | unrolled block:";
    for (int p : s_.unrolled) o_ << " " << p;
    o_ << " (" << pow3(nu_) << " configurations per outer step)\n";
    // The lane's LDS row is idle between the single-posterior store and the final write-back:
    // the looped members' marginal accumulators (touched once per iteration of their own loop)
    // and, when they fit, their likelihoods live there instead of in registers, which keeps the
    // unrolled block free of scratch traffic.
    for (int k = 0; k < no; ++k)
      if (folded(outer_[k]))  // the founder's prior goes into its likelihood once per site (the product the loop would form each time)
        for (int g = 0; g < 3; ++g)
          o_ << "      const double pl" << outer_[k] << "_" << g << " = tcf[" << kind_of(m_, outer_[k]) * 27 + 9 * g << "] * l" << outer_[k] << "_" << g << ";\n";
    for (int k = 0; k < no; ++k)
      for (int g = 0; g < 3; ++g) {
        o_ << "      " << S_ << "[" << 3 * k + g << "] = 0;\n";
        if (l_in_lds_) o_ << "      " << S_ << "[" << 3 * no + 3 * k + g << "] = " << l_name(outer_[k], g) << ";\n";
      }
    // The unrolled members' likelihoods are needed only where their tables are rebuilt (outer loop
    // levels).  What is left of the LDS row holds them for the members whose tables sit in the
    // deepest loops (read 3^depth times per site: +3 % on ped10); the others are re-read from the
    // site's own row in global memory there (L2 hits).  Either way 6 registers per member stay free
    // for the block.
    {
      const std::vector<int> wb = table_buckets();
      std::vector<int> order;
      for (int k = 0; k < nu_; ++k)
        if (wb[k] >= 0) order.push_back(k);
        else before_loops_.insert(s_.unrolled[k]);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wb[a] > wb[b]; });
      int next = (l_in_lds_ ? 6 : 3) * no;
      for (int k : order) {
        if (next + 3 > row_len) break;
        lds_slot_[s_.unrolled[k]] = next;
        for (int g = 0; g < 3; ++g) o_ << "      " << S_ << "[" << next + g << "] = l" << s_.unrolled[k] << "_" << g << ";\n";
        next += 3;
      }
    }
    for (int k = 0; k < fixed_; ++k) {  // this lane's digits of the members that do not loop
      int div = 1;
      for (int j = 0; j < k; ++j) div *= 3;
      o_ << "      const int fx" << k << " = (sub / " << div << ") % 3;\n";
    }
    joint_ = sl_ >= 2;
    if (const char *e = std::getenv("FAMSEQ_LANE_JOINT")) joint_ = joint_ && std::atoi(e) != 0;  // tuning aid
    for (int k = 0; k < nu_; ++k) {
      if (joint_ && k >= nu_ - sl_) continue;
      const int p = s_.unrolled[k];
      o_ << "      double b" << p << "_0 = 0, b" << p << "_1 = 0, b" << p << "_2 = 0;\n";
    }
    if (joint_) {
      std::vector<int> dig(nu_, 0);
      for (int c = 0; c < pow3(sl_); ++c) o_ << "      double " << s_name(c, dig) << " = 0;\n";
    }
    o_ << "      const double P_root = 10000000.0;\n";  // family.cpp:911
    tables();
    o_ << bucket_[0];
    outer_level(0, "P_root", "");
    for (int k = 0; k < no; ++k)
      o_ << "      const double b" << outer_[k] << "_0 = " << S_ << "[" << 3 * k << "], b" << outer_[k] << "_1 = " << S_ << "[" << 3 * k + 1
         << "], b" << outer_[k] << "_2 = " << S_ << "[" << 3 * k + 2 << "];\n";
    if (joint_) {  // the super-leaf members' marginals: sums of the joint accumulators over the other digits
      std::vector<int> dig(nu_, 0);
      for (int j = 0; j < sl_; ++j)
        for (int g = 0; g < 3; ++g) {
          std::string e;
          for (int c = 0; c < pow3(sl_); ++c) {
            int cc = c, mine = 0;
            for (int t = sl_ - 1; t >= 0; --t) {
              if (t == j) mine = cc % 3;
              cc /= 3;
            }
            if (mine != g) continue;
            const std::string n = s_name(c, dig);
            e = e.empty() ? n : "(" + e + " + " + n + ")";
          }
          o_ << "      const double b" << s_.unrolled[nu_ - sl_ + j] << "_" << g << " = " << e << ";\n";
        }
    }
    if (fixed_ > 0) {  // this lane's share of the marginals, unnormalised: reduce_body() sums the group's
      for (int p = 0; p < s_.N; ++p)
        for (int g = 0; g < 3; ++g) o_ << "      row[" << 3 * p + g << "] = b" << p << "_" << g << ";\n";
      return o_.str();
    }
    // One division per row and three products (the sums differ from the reference's in their last bits already, by
    // summation order; 35 -> 25 divisions per five-member site, each ten instructions).  A row sum in the subnormal
    // range, whose reciprocal overflows, keeps the three divisions behind a real branch.
    bool div_rows = false;
    if (const char *e = std::getenv("FAMSEQ_LANE_DIVROWS")) div_rows = std::atoi(e) != 0;  // tuning aid: three divisions per row
    for (int p = 0; p < s_.N; ++p) {
      auto out = [&](int g) { return O_ + "[" + std::to_string(3 * p + g) + "]"; };
      const std::string b = "b" + std::to_string(p);
      if (div_rows) {
        o_ << "      { const double s = (" << b << "_0 + " << b << "_1) + " << b << "_2; if (s <= 0) bn_fail = true;\n        " << out(0) << " = "
           << b << "_0 / s; " << out(1) << " = " << b << "_1 / s; " << out(2) << " = " << b << "_2 / s; }\n";
        continue;
      }
      o_ << "      { const double s = (" << b << "_0 + " << b << "_1) + " << b << "_2; if (s <= 0) bn_fail = true;\n"
         << "        if (s < 1e-290) { asm volatile(\"\" ::: \"memory\"); " << out(0) << " = " << b << "_0 / s; " << out(1) << " = " << b
         << "_1 / s; " << out(2) << " = " << b << "_2 / s; }\n"
         << "        else { const double r = 1.0 / s; " << out(0) << " = " << b << "_0 * r; " << out(1) << " = " << b << "_1 * r; " << out(2)
         << " = " << b << "_2 * r; } }\n";
    }
    return o_.str();
  }

 private:
  const Model &m_;
  const Shape &s_;
  const int nu_;
  const int row_len_;  // doubles in the lane's LDS row (>= 3N, odd)
  const int fixed_;    // outermost looped members whose digit is the lane's (lanes-per-site mode)
  const std::string S_, O_;  // the body's scratch array and where the normalised marginals go
  std::vector<int> outer_;  // looped members, outermost first
  const bool st_;           // see the constructor
  const int pre_;
  std::string prologue_, prefetch_;  // (pre_) before the innermost loop / inside it, between its table statements and the block
  std::map<std::string, std::string> tq_;  // (pre_) table entry (index text with the innermost digit as '@') -> its loop-carried variable
  std::ostringstream o_;
  int uid_ = 0;
  bool l_in_lds_ = false;
  bool joint_ = false;  // super-leaf marginals from 3^sl joint accumulators (see superleaf())
  std::map<int, int> lds_slot_;  // unrolled member -> first of its 3 slots in the lane's LDS row
  std::set<int> before_loops_;   // unrolled members whose tables are built once per site

  static int pow3(int e) {
    int r = 1;
    while (e-- > 0) r *= 3;
    return r;
  }

  // table offset of member p's factor for child genotype expression `gc` ("2" or "g7"): literal
  // part + digits of outer parents (runtime, uniform) — unrolled parents are added by the caller
  // `dm`, `dexpr`: the digit of looped member dm is spelled dexpr instead of g<dm> (the prefetch's next digit)
  std::string t_index(int p, const std::string &gc, int um, int uf, int dm = -1, const std::string &dexpr = "") const {
    auto dig = [&](int q) { return q == dm ? dexpr : "g" + num(q); };
    std::string e = num(kind_of(m_, p) * 27) + " + 9 * " + gc;
    if (m_.mother[p] >= 0) {
      e += um >= 0 ? " + " + num(3 * um) : " + 3 * " + dig(m_.mother[p]);
      e += uf >= 0 ? " + " + num(uf) : " + " + dig(m_.father[p]);
    }
    return e;
  }
  // the table a member's factor entries are read from: the lane's LDS copy (flag-selected: Known picks the founders'
  // priors), or — scalar_t, children only — the wave-uniform pointer of the body's chrX pass
  std::string t_tab(int p) const { return st_ && m_.mother[p] >= 0 ? "tcx" : "tcf"; }
  // founder with its prior folded into the likelihood (scalar_t): pl<p>_<g>, formed once per site
  bool folded(int p) const { return st_ && m_.mother[p] < 0; }
  std::string l_name(int p, int g) const { return (folded(p) && outer_pos(p) >= 0 ? "pl" : "l") + num(p) + "_" + num(g); }
  int inner_pos() const { return (int)outer_.size() - 1; }
  // (pre_) the loop-carried variable holding table entry `idx` ('@' = the innermost looped member's digit): loaded for
  // digit 0 ahead of the innermost loop, for the next digit inside it once this digit's table statements are done
  std::string carried_entry(const std::string &tab, const std::string &idx) {
    const std::string key = tab + "[" + idx + "]";
    auto it = tq_.find(key);
    if (it != tq_.end()) return it->second;
    const std::string v = "tq" + num((int)tq_.size()), gi = "g" + num(outer_[inner_pos()]);
    auto spell = [&](const std::string &d) {
      std::string e = key;
      for (size_t k; (k = e.find('@')) != std::string::npos;) e.replace(k, 1, d);
      return e;
    };
    prologue_ += "            double " + v + " = " + spell("0") + ";\n";
    prefetch_ += "              " + v + " = " + spell(gi + "n") + ";\n";
    tq_[key] = v;
    return v;
  }

  void outer_level(size_t k, const std::string &P, const std::string &acc_parent) {
    if (k == outer_.size()) {
      block(P, acc_parent);
      return;
    }
    const int p = outer_[k];
    const int no = (int)outer_.size();
    const std::string g = "g" + num(p), ind(6 + 2 * k, ' ');
    const std::string lk_g = l_in_lds_ ? S_ + "[" + num(3 * no + 3 * (int)k) + " + " + g + "]"
                                       : "(" + g + " == 0 ? " + l_name(p, 0) + " : (" + g + " == 1 ? " + l_name(p, 1) + " : " + l_name(p, 2) + "))";
    const bool inner = pre_ > 0 && (int)k == inner_pos();
    const bool carried_l = inner && pre_ >= 2 && l_in_lds_;
    std::string f_expr = carried_l ? "lq_" : lk_g;
    if (!folded(p))
      f_expr = (inner && m_.mother[p] >= 0 ? carried_entry("tcx", t_index(p, "@", -1, -1, p, "@")) : t_tab(p) + "[" + t_index(p, g, -1, -1) + "]") + " * " + f_expr;
    if (inner) o_ << prologue_;
    if (carried_l)  // this loop's own LDS reads, one step ahead: the member's likelihood, its marginal slot
      o_ << ind << "double lq_ = " << S_ << "[" << 3 * no + 3 * (int)k << "], aq_ = " << S_ << "[" << 3 * (int)k << "];\n";
    if ((int)k < fixed_)
      o_ << ind << "{ const int " << g << " = fx" << k << ";  // one digit per lane of the group\n";
    else
      o_ << "#pragma unroll 1\n"  // keep the walk rolled: an unrolled outer loop triples the block's live state
         << ind << "for (int " << g << " = 0; " << g << " < 3; ++" << g << ") {\n";
    o_ << ind << "  const double f" << p << " = " << f_expr << ";\n"
       << ind << "  const double P" << p << " = " << P << " * f" << p << ";\n"
       << ind << "  double acc" << p << " = 0;\n"
       << bucket_[k + 1];
    if (inner) {
      o_ << ind << "  const int " << g << "n = " << g << " < 2 ? " << g << " + 1 : 2;\n" << prefetch_;
      if (carried_l) o_ << ind << "  const double lq_n = " << S_ << "[" << 3 * no + 3 * (int)k << " + " << g << "n];\n";
      o_ << ind << "  __builtin_amdgcn_sched_barrier(0);\n";
    }
    outer_level(k + 1, "P" + num(p), "acc" + num(p));
    if (carried_l)
      o_ << ind << "  " << S_ << "[" << 3 * (int)k << " + " << g << "] = aq_ + acc" << p << ";\n"
         << ind << "  lq_ = lq_n; aq_ = " << S_ << "[" << 3 * (int)k << " + " << g << "n];\n";
    else
      o_ << ind << "  " << S_ << "[" << 3 * (int)k << " + " << g << "] += acc" << p << ";\n";
    if (!acc_parent.empty()) o_ << ind << "  " << acc_parent << " += acc" << p << ";\n";
    o_ << ind << "}\n";
  }

  // name of table entry of unrolled level k for own digit g given the digits of the levels above
  std::string w_name(int k, int g, const std::vector<int> &dig) const {
    const int p = s_.unrolled[k];
    std::string n = "w" + num(p) + "_" + num(g);
    if (m_.mother[p] >= 0) {
      if (s_.upos[m_.mother[p]] >= 0) n += "m" + num(dig[s_.upos[m_.mother[p]]]);
      if (s_.upos[m_.father[p]] >= 0) n += "f" + num(dig[s_.upos[m_.father[p]]]);
    }
    return n;
  }
  // dep_[k]: unrolled levels < k whose digits the block sums of levels >= k depend on
  std::vector<std::vector<int>> dep_;

  void compute_deps() {
    dep_.assign(nu_ + 1, {});
    for (int k = nu_ - 1; k >= 0; --k) {
      std::vector<char> in(nu_, 0);
      for (int j = k; j < nu_; ++j) {
        const int p = s_.unrolled[j];
        if (m_.mother[p] < 0) continue;
        for (int par : {m_.mother[p], m_.father[p]})
          if (s_.upos[par] >= 0 && s_.upos[par] < k) in[s_.upos[par]] = 1;
      }
      for (int l = 0; l < k; ++l)
        if (in[l]) dep_[k].push_back(l);
    }
  }
  // Super-leaf: the deepest t levels are walked together.  Their factors are multiplied once per
  // outer step into a combined table W (3^t entries per combination of the upper digits they
  // depend on), so each of the 3^t configurations below a prefix costs exactly one FMA
  // (prefix * W into the deepest member's marginal) and the other t-1 members take one FMA per
  // combination of their own and the shallower super-leaf digits (prefix * WQ).  No products are
  // formed inside the block for these levels.
  int sl_ = 1;  // t
  void choose_superleaf() {
    sl_ = 1;
    for (int t = std::min(3, nu_); t >= 2; --t) {
      const int d = (int)dep_[nu_ - t].size();
      int doubles = pow3(t + d);
      for (int j = 0; j + 1 < t; ++j) doubles += pow3(j + 1 + d);
      if (doubles <= 45) {
        sl_ = t;
        break;
      }
    }
  }
  std::string sl_name(const char *prefix, int upto, const std::vector<int> &dig) const {  // W / WQ<j> entry
    const int k0 = nu_ - sl_;
    std::string n = prefix;
    for (int k = k0; k <= upto; ++k) n += "_" + num(dig[k]);
    for (int l : dep_[k0]) n += "_" + num(l) + "d" + num(dig[l]);
    return n;
  }

  // joint accumulator of super-leaf configuration c (most significant digit = shallowest member)
  std::string s_name(int c, std::vector<int> &dig) const {
    const int k0 = nu_ - sl_;
    for (int t = sl_ - 1; t >= 0; --t) {
      dig[k0 + t] = c % 3;
      c /= 3;
    }
    std::string n = "S";
    for (int k = k0; k < nu_; ++k) n += "_" + num(dig[k]);
    return n;
  }

  // Q<k>[digits of dep_[k]] = sum over the configurations of levels k.. of prod w: the total
  // weight below a node, per unit of prefix.  Q<nu> = 1.
  std::string q_name(int k, const std::vector<int> &dig) const {
    if (k >= nu_) return "1.0";
    std::string n = "Q" + num(k);
    for (int l : dep_[k]) n += "_" + num(l) + "d" + num(dig[l]);
    return n;
  }

  // Table statements, bucketed by the innermost outer loop whose digit they mention
  // (index into outer_, -1 = none): each bucket is emitted at the top of that loop's body, so a
  // table is rebuilt only when a digit it depends on changes.
  std::vector<std::string> bucket_;  // [outer position + 1]
  // Lazy tables (FAMSEQ_LANE_LAZY, an experiment that is OFF: measured slower): entries (factor tables of
  // prefix levels, block sums) that depend on the FIRST unrolled member's digit only, and are rebuilt in the
  // innermost loop anyway, are not built at the top of the block for all three digits but inside that
  // digit's part of the unrolled tree: a third of them is live at a time.  At ten members that is 14
  // doubles = 28 VGPRs — the whole of the spill, which the eager form reloads from scratch inside the
  // block (10 scratch loads per outer step): scratch goes from 108 B to 0.  But the LDS reads behind the
  // table statements then sit in the middle of the block: 11.1 ms per 4 M sites against 10.3 (built one
  // digit ahead: 10.6, scratch back at 100 B).  The block total Q0 is summed as the parts go by (same FMA
  // order), so the results are bit-identical either way.
  bool lazy0_ = false, lazy_ahead_ = false;
  std::string lazy0_stmt_[3];
  bool q0_incremental_ = false;

  // where the block's table statements read unrolled member p's likelihood from
  std::string lk_src(int p, int g) const {
    const auto it = lds_slot_.find(p);
    if (it != lds_slot_.end()) return S_ + "[" + num(it->second + g) + "]";
    if (before_loops_.count(p)) return "l" + num(p) + "_" + num(g);  // used once, ahead of all loops: still in registers
    return "lg[" + num(3 * p + g) + "]";
  }

  int outer_pos(int member) const {
    for (size_t k = 0; k < outer_.size(); ++k)
      if (outer_[k] == member) return (int)k;
    return -1;
  }

  // Loop level (index into outer_, -1 = before all loops) at which unrolled level k's factor table
  // has to be rebuilt: that of its innermost looped parent.
  std::vector<int> table_buckets() const {
    std::vector<int> wb(nu_, -1);
    for (int k = 0; k < nu_; ++k) {
      const int p = s_.unrolled[k];
      if (m_.mother[p] >= 0)
        for (int par : {m_.mother[p], m_.father[p]})
          if (s_.upos[par] < 0) wb[k] = std::max(wb[k], outer_pos(par));
    }
    return wb;
  }

  void tables() {
    const std::string ind = "        ";
    bucket_.assign(outer_.size() + 1, "");
    const std::vector<int> wb = table_buckets();  // bucket of level k's factor table
    std::vector<int> qb(nu_ + 1, -1);             // ... and of its block sums
    for (int k = nu_ - 1; k >= 0; --k) qb[k] = std::max(qb[k + 1], wb[k]);
    // which levels' tables / block sums are lazy (see lazy0_): a prefix level k >= 1 whose block sums depend on
    // level 0's digit only and sit in the innermost loop; the chain must be unbroken from level 1 on, because
    // the (eager) sums of a level use those of the next
    const int innermost = (int)outer_.size() - 1;
    const int n_prefix = nu_ - (sl_ >= 2 ? sl_ : 1);  // levels above the (super-)leaf
    std::vector<char> lazy_q(nu_ + 1, 0), lazy_w(nu_ + 1, 0);
    if (const char *e = std::getenv("FAMSEQ_LANE_LAZY")) {  // tuning aid: 0 eager, 1 lazy, 2 lazy one digit ahead
      lazy0_ = std::atoi(e) != 0;
      lazy_ahead_ = std::atoi(e) == 2;
    }
    if (lazy0_ && innermost >= 0 && n_prefix >= 2)
      for (int k = 1; k < n_prefix; ++k) {
        if (!(dep_[k].size() == 1 && dep_[k][0] == 0 && qb[k] == innermost)) break;
        lazy_q[k] = 1;
        const int p = s_.unrolled[k];
        const bool mu = m_.mother[p] >= 0 && s_.upos[m_.mother[p]] >= 0, fu = m_.mother[p] >= 0 && s_.upos[m_.father[p]] >= 0;
        const bool m0 = mu && s_.upos[m_.mother[p]] == 0, f0 = fu && s_.upos[m_.father[p]] == 0;
        lazy_w[k] = wb[k] == innermost && (mu + fu) == 1 && (m0 || f0);
      }
    q0_incremental_ = lazy_q[1];
    for (int k = 0; k < nu_; ++k) {
      std::ostringstream o;
      const int p = s_.unrolled[k];
      const bool has = m_.mother[p] >= 0;
      const bool mu = has && s_.upos[m_.mother[p]] >= 0, fu = has && s_.upos[m_.father[p]] >= 0;
      for (int gm = 0; gm < (mu ? 3 : 1); ++gm)
        for (int gf = 0; gf < (fu ? 3 : 1); ++gf) {
          std::string suffix;
          if (mu) suffix += "m" + num(gm);
          if (fu) suffix += "f" + num(gf);
          for (int g = 0; g < 3; ++g) {
            o << ind << "const double w" << p << "_" << g << suffix << " = ";
            if (pre_ > 0 && has && wb[k] == innermost) {
              // rebuilt in the innermost loop: the entry comes from the variable loaded a step ahead, the likelihood — the
              // same in every step — from a register filled ahead of the loop (prefetch 2) instead of a read per step
              const int pin = outer_[innermost];
              o << carried_entry("tcx", t_index(p, num(g), mu ? gm : -1, fu ? gf : -1, pin, "@")) << " * ";
              const std::string src = lk_src(p, g);
              if (pre_ >= 2 && src.compare(0, S_.size() + 1, S_ + "[") == 0) {
                const std::string v = "lq" + num(p) + "_" + num(g);
                if (prologue_.find(" " + v + " =") == std::string::npos) prologue_ += "            const double " + v + " = " + src + ";\n";
                o << v << ";\n";
              } else {
                o << src << ";\n";
              }
            } else {
              o << t_tab(p) << "[" << t_index(p, num(g), mu ? gm : -1, fu ? gf : -1) << "] * " << lk_src(p, g) << ";\n";
            }
          }
        }
      if (lazy_w[k]) continue;  // emitted per digit of the first unrolled member (below)
      bucket_[wb[k] + 1] += o.str();
    }
    for (int k = 1; k < nu_; ++k) {
      if (!lazy_w[k]) continue;
      const int p = s_.unrolled[k];
      const bool mu = s_.upos[m_.mother[p]] == 0, fu = s_.upos[m_.father[p]] == 0;  // exactly one of them is level 0
      for (int d = 0; d < 3; ++d) {
        std::ostringstream o;
        const std::string suffix = (mu ? "m" : "f") + num(d);
        for (int g = 0; g < 3; ++g)
          o << ind << "const double w" << p << "_" << g << suffix << " = " << t_tab(p) << "[" << t_index(p, num(g), mu ? d : -1, fu ? d : -1)
            << "] * " << lk_src(p, g) << ";\n";
        lazy0_stmt_[d] += o.str();
      }
    }
    // block sums, deepest level first
    for (int k = nu_ - 1; k >= 0; --k) {
      std::ostringstream o;
      const int nd = (int)dep_[k].size();
      std::vector<int> dig(nu_, 0);
      for (int code = 0; code < pow3(nd); ++code) {
        int c = code;
        for (int l : dep_[k]) {
          dig[l] = c % 3;
          c /= 3;
        }
        std::string e;
        for (int g = 0; g < 3; ++g) {
          dig[k] = g;
          const std::string w = w_name(k, g, dig), q = q_name(k + 1, dig);
          if (q == "1.0") e = e.empty() ? w : "(" + e + " + " + w + ")";
          else e = e.empty() ? "(" + w + " * " + q + ")" : "__builtin_fma(" + w + ", " + q + ", " + e + ")";
        }
        if (k == 0 && q0_incremental_) continue;  // summed inside the block as level 0's digits go by
        if (lazy_q[k]) lazy0_stmt_[dig[0]] += ind + "const double " + q_name(k, dig) + " = " + e + ";\n";
        else o << ind << "const double " << q_name(k, dig) << " = " << e << ";\n";
      }
      bucket_[qb[k] + 1] += o.str();
    }
    if (sl_ < 2) return;
    // super-leaf tables: running products over the t levels, per combination of the upper digits
    std::ostringstream o;
    const int k0 = nu_ - sl_, nd = (int)dep_[k0].size();
    std::vector<int> dig(nu_, 0);
    for (int code = 0; code < pow3(nd); ++code) {
      int c = code;
      for (int l : dep_[k0]) {
        dig[l] = c % 3;
        c /= 3;
      }
      std::function<void(int, const std::string &)> walk = [&](int k, const std::string &prod) {
        for (int g = 0; g < 3; ++g) {
          dig[k] = g;
          const std::string w = w_name(k, g, dig);
          std::string here = w;
          if (!prod.empty()) {
            here = sl_name(k == nu_ - 1 ? "W" : "X", k, dig);
            o << ind << "const double " << here << " = " << prod << " * " << w << ";\n";
          }
          if (k < nu_ - 1) {
            if (!joint_)
              o << ind << "const double " << sl_name(("WQ" + num(k - k0)).c_str(), k, dig) << " = " << here << " * "
                << q_name(k + 1, dig) << ";\n";
            walk(k + 1, here);
          }
        }
      };
      walk(k0, "");
    }
    bucket_[qb[k0] + 1] += o.str();
  }

  void superleaf(const std::string &P, std::vector<int> &dig, const std::string &ind) {
    const int k0 = nu_ - sl_, last = s_.unrolled[nu_ - 1];
    if (joint_) {
      // Every configuration: ONE FMA — its joint weight prefix * W is formed and added to the
      // accumulator of its super-leaf digits.  The 3^sl accumulators run over the whole site; the
      // marginals of all sl members are sums of them, taken once at the end (body()).  Against one
      // set of bins per member (39 FMAs per 27 configurations at sl = 3) this is 27 per 27, and
      // the accumulators form 3^sl independent dependency chains.
      // (No pins here: the prefix products are pinned where they are formed, which is what keeps
      // hipcc from forming all of them up front; a pin per three FMAs cost an s_nop each, -5 %.)
      for (int c = 0; c < pow3(sl_); ++c) {
        const std::string a = s_name(c, dig);  // sets dig[k0..]
        o_ << ind << a << " = __builtin_fma(" << P << ", " << sl_name("W", nu_ - 1, dig) << ", " << a << ");\n";
      }
      return;
    }
    // every configuration: one FMA, its joint weight formed as prefix * W
    std::function<void(int)> leaves = [&](int k) {
      for (int g = 0; g < 3; ++g) {
        dig[k] = g;
        if (k < nu_ - 1) {
          leaves(k + 1);
          continue;
        }
        o_ << ind << "b" << last << "_" << g << " = __builtin_fma(" << P << ", " << sl_name("W", nu_ - 1, dig) << ", b"
           << last << "_" << g << ");\n";
      }
      if (k == nu_ - 1)
        o_ << ind << "asm volatile(\"\" : \"+v\"(b" << last << "_0), \"+v\"(b" << last << "_1), \"+v\"(b" << last
           << "_2), \"+v\"(" << P << "));\n";
    };
    leaves(k0);
    // the shallower super-leaf members: one FMA per combination of their digits
    for (int j = 0; j + 1 < sl_; ++j) {
      const int p = s_.unrolled[k0 + j];
      std::function<void(int)> bins = [&](int k) {
        for (int g = 0; g < 3; ++g) {
          dig[k] = g;
          if (k < k0 + j) {
            bins(k + 1);
            continue;
          }
          o_ << ind << "b" << p << "_" << g << " = __builtin_fma(" << P << ", " << sl_name(("WQ" + num(j)).c_str(), k0 + j, dig)
             << ", b" << p << "_" << g << ");\n";
        }
      };
      bins(k0);
      o_ << ind << "asm volatile(\"\" : \"+v\"(b" << p << "_0), \"+v\"(b" << p << "_1), \"+v\"(b" << p << "_2), \"+v\"("
         << P << "));\n";
    }
  }

  // Where prefix products are formed the instruction order is pinned (left alone, hipcc forms the
  // products of the whole unrolled tree ahead of their uses and spills).  pin_style_ 1: an empty asm
  // statement tying the values to registers at that point; 2: a scheduling barrier
  // (__builtin_amdgcn_sched_barrier: nothing is moved across, and no instruction is emitted — the asm
  // form costs an s_nop each); 0: none.
  int pin_style_ = 2;  // measured on MI355X: 10.59 ms (asm) -> 10.20 ms (barrier) per 4 M 10-member sites, 139.4 -> 136.8 ms at 15 members
  std::string pin(const std::string &operands, const std::string &ind) const {
    if (pin_style_ == 2) return ind + "__builtin_amdgcn_sched_barrier(0);\n";
    if (pin_style_ == 0) return "";
    return ind + "asm volatile(\"\" : " + operands + ");\n";
  }

  void level(int k, const std::string &P, std::vector<int> &dig, const std::string &ind) {
    const int p = s_.unrolled[k];
    if (sl_ >= 2 && k == nu_ - sl_) {
      superleaf(P, dig, ind);
      return;
    }
    if (k == nu_ - 1) {
      for (int g = 0; g < 3; ++g)
        o_ << ind << "b" << p << "_" << g << " = __builtin_fma(" << P << ", " << w_name(k, g, dig) << ", b" << p << "_" << g
           << ");\n";
      o_ << ind << "asm volatile(\"\" : \"+v\"(b" << p << "_0), \"+v\"(b" << p << "_1), \"+v\"(b" << p << "_2), \"+v\"(" << P
         << "));\n";
      return;
    }
    for (int g = 0; g < 3; ++g) {
      dig[k] = g;
      if (k == 0) {
        // one digit ahead: the LDS reads behind these statements return while the previous digit's part of the
        // tree is computed (built at the head of their own part they stalled it: 11.1 vs 10.3 ms per 4 M sites)
        if (lazy_ahead_) {
          if (g == 0) o_ << lazy0_stmt_[0];
          if (g < 2) o_ << lazy0_stmt_[g + 1];
        } else {
          o_ << lazy0_stmt_[g];
        }
        if (q0_incremental_) {  // the block total, in the order the eager form adds it
          const std::string w = w_name(0, g, dig), q = q_name(1, dig);
          if (g == 0) o_ << ind << "double Q0i = " << w << " * " << q << ";\n";
          else o_ << ind << "Q0i = __builtin_fma(" << w << ", " << q << ", Q0i);\n";
        }
      }
      const std::string pg = "p" + num(uid_++);
      o_ << ind << "double " << pg << " = " << P << " * " << w_name(k, g, dig) << ";\n"
         << ind << "b" << p << "_" << g << " = __builtin_fma(" << pg << ", " << q_name(k + 1, dig) << ", b" << p << "_" << g
         << ");\n"
         << pin("\"+v\"(" + pg + "), \"+v\"(b" + num(p) + "_" + num(g) + ")", ind);
      level(k + 1, pg, dig, ind);
      o_ << pin("\"+v\"(" + P + "), \"+v\"(b" + num(p) + "_" + num(g) + ")", ind);
    }
  }

  void block(const std::string &P, const std::string &acc_parent) {
    const std::string ind(6 + 2 * outer_.size(), ' ');
    o_ << ind << "{\n";
    const std::string in2 = ind + "  ";
    std::vector<int> dig(nu_, 0);
    o_ << in2 << "double Pb = " << P << ";\n";
    if (!acc_parent.empty() && !q0_incremental_) o_ << in2 << acc_parent << " += Pb * " << q_name(0, dig) << ";\n";
    level(0, "Pb", dig, in2);
    if (!acc_parent.empty() && q0_incremental_) o_ << in2 << acc_parent << " += Pb * Q0i;\n";
    o_ << ind << "}\n";
  }
};

}  // namespace

std::string enumgen_describe(const Model &m, int variant) {
  int cap = (variant >= 0 && variant < 2) ? 7 : 6;  // kEnumVariants; unknown yet (-1): the 6-member form
  if (const char *e = std::getenv("FAMSEQ_LANE_CAP")) cap = std::atoi(e);
  const Shape s = choose_shape(m, cap);
  std::string d = "looped members [";
  for (size_t k = 0; k < s.outer.size(); ++k) d += (k ? " " : "") + num(s.outer[k]);
  d += "], unrolled block [";
  for (size_t k = 0; k < s.unrolled.size(); ++k) d += (k ? " " : "") + num(s.unrolled[k]);
  int n = 1;
  for (size_t k = 0; k < s.unrolled.size(); ++k) n *= 3;
  return d + "] = " + num(n) + " configurations per step";
}

// One lane per site: workgroups of ONE wave, and no register cap (`__launch_bounds__(64, 1)`).  A wave that
// shares its workgroup with nobody waits at no real barrier, so the waves of a CU drift apart and one's memory phases
// overlap another's arithmetic (five members 0.724 -> 0.683 ms per 8 M sites, quads 0.570 -> 0.537); and where the
// arithmetic needs more than 256 registers the compiler now takes one wave per SIMD with its overflow in AGPRs —
// the ten-member kernel: 24 of them instead of 108 bytes of scratch per lane, an LDS row of 43 instead of 37 doubles
// (a quarter of the lanes per CU: no likelihood is re-read from global memory any more), 10.21 -> 9.78 ms per 4 M sites
// (profiles/r02c/exp_block_sizes_*.txt).  The lanes-per-site forms keep wide workgroups: a site's 81 lanes span waves.
int enumgen_block_threads(const Model &m, int group_digits) {
  if (const char *e = std::getenv("FAMSEQ_LANE_BT")) return std::atoi(e);  // tuning aid
  if (group_digits == 0) return 64;
  return m.n_members <= 10 ? 256 : 128;
}

// (asked of the call-path form: its LDS row has less room than the plain form's, so it may re-read members
// the plain form keeps in LDS — and it is the form that can be fed packed PLs, with no fp64 rows to read)
bool enumgen_reads_global_rows(const Model &m, int variant) {
  return enumgen_source(m, variant, 0, /*call_mode=*/true).find("lg[") != std::string::npos;
}

int enumgen_max_group_digits(const Model &m) {
  int cap = 6;
  if (const char *e = std::getenv("FAMSEQ_LANE_CAP")) cap = std::atoi(e);
  return std::min<int>(kEnumMaxGroupDigits, (int)choose_shape(m, cap).outer.size());
}

int enumgen_sites_per_chunk(const Model &m, int group_digits) {
  int g = 1;
  for (int k = 0; k < group_digits; ++k) g *= 3;
  return enumgen_block_threads(m, group_digits) / g;
}

namespace {

// Shell of the lanes-per-site mode (small batches): G = 3^d consecutive lanes share a site, each
// walks the digits (fx0, fx1, ...) of the d outermost looped members given by its position in the
// group, i.e. 1/G of the enumeration; the partial marginals meet in the lanes' LDS rows, the group
// adds them column by column in lane order, its first lane normalises and applies the failure rule.  A site's
// latency drops by G and G times as many lanes are busy, which is what a batch too small to give
// every lane of the chip a site of its own needs (one lane per site: 0.17 ms for anything up to 131 k
// 10-member sites).  I/O is a plain strided walk — this shell never sees a large batch.
std::string grouped_shell(const Model &m, const std::string &comment, const std::string &body, const std::string &reduce,
                          int bt, int min_waves, bool fence_single, int row_doubles, int group) {
  const int N = m.n_members, W3 = 3 * N, ROW = (row_doubles > 0 ? row_doubles : W3) | 1;
  std::ostringstream s;
  s << "// generated by famseq_amd/csrc for a " << N << "-member pedigree: " << comment << "\n"
    // (the in-process compiler, hiprtc, brings the device built-ins itself and has no include path for the header)
    << "#ifndef __HIPCC_RTC__\n#include <hip/hip_runtime.h>\n#endif\n#pragma clang fp contract(off)\n"
    << "#define W3 " << W3 << "\n#define ROW " << ROW << "\n#define BT " << bt << "\n#define G " << group << "\n#define SPC (BT / G)\n"
    << "#define LDS_BARRIER() asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\")\n"
    << kDiv3Text
    // rows of the groups' first lanes -> global, coalesced
    << "#define STAGE_OUT(Gp) { double *g_ = (Gp) + site0 * W3; \\\n"
    << "  for (int e = tid; e < ns * W3; e += BT) { const int s_ = e / W3; g_[e] = s_io[s_ * G * ROW + (e - s_ * W3)]; } }\n"
    << "extern \"C\" __global__ __launch_bounds__(BT, " << min_waves << ") void famseq_enum_lane(const double *__restrict__ lk_g,\n"
    << "    const unsigned char *__restrict__ flags_g, double *__restrict__ post_g, double *__restrict__ single_g,\n"
    << "    unsigned char *__restrict__ status_g, long n_sites, const double *__restrict__ tc_g, double lc) {\n"
    << "  __shared__ double s_io[BT * ROW];  // one padded row per lane\n"
    << "  __shared__ double s_tc[432];\n"
    << "  const int tid = threadIdx.x;\n"
    << "  for (int i = tid; i < 432; i += BT) s_tc[i] = tc_g[i];\n"
    << "  const int sidx = tid / G, sub = tid - sidx * G;  // site within the chunk, lane within the group\n"
    << "  const long chunks = (n_sites + SPC - 1) / SPC;\n"
    << "  const long q_wg = chunks / gridDim.x, r_wg = chunks - q_wg * gridDim.x;  // q or q + 1 chunks each: no idle workgroup\n"
    << "  const long c_lo = (long)blockIdx.x * q_wg + (blockIdx.x < r_wg ? blockIdx.x : r_wg), c_hi = c_lo + q_wg + (blockIdx.x < r_wg ? 1 : 0);\n"
    << "  const double kNaN = __builtin_nan(\"\");\n"
    << "  double *row = s_io + tid * ROW;\n"
    << "  for (long ch = c_lo; ch < c_hi; ++ch) {\n"
    << "    const long site0 = ch * SPC;\n"
    << "    const int ns = n_sites - site0 < SPC ? (int)(n_sites - site0) : SPC;\n"
    << "    const bool act = sidx < ns;  // lanes beyond the chunk's sites compute on its first site and store nothing\n"
    << "    const long site = site0 + (act ? sidx : 0);\n"
    << "    const double *lg = lk_g + site * W3;\n"
    << "    LDS_BARRIER();  // the previous chunk's rows have been stored; the table is in LDS\n"
    << "    const int fl = flags_g ? (flags_g[site] & 3) : 0;\n"
    << "    const double *tcf = s_tc + fl * 108;\n"
    << "    bool single_fail = false, full = false, bn_fail = false;\n";
  for (int p = 0; p < N; ++p)
    for (int gt = 0; gt < 3; ++gt) s << "    const double l" << p << "_" << gt << " = lg[" << 3 * p + gt << "];\n";
  s << single_posterior_statements(m, true, true, fence_single)
    << "    if (!act) full = false;\n"
    << "    if (single_fail) for (int k = 0; k < W3; ++k) row[k] = kNaN;\n"
    << "    LDS_BARRIER();\n"
    << "    if (single_g) { STAGE_OUT(single_g); }\n"
    << "    LDS_BARRIER();  // single rows are stored; sites that need the full computation overwrite theirs\n"
    << "    if (full && !single_fail) {\n"
    << body
    << "    }\n"
    << "    LDS_BARRIER();  // every lane's share of the marginals is in its row\n"
    // column sums over the group's rows, the columns dealt round the group's lanes (3N / G + 1 of them
    // each), every column in lane order 0..G-1 (bit-reproducible); the sums land in the first lane's row
    << "    if (full && !single_fail) {\n"
    << "      double *lead = row - sub * ROW;\n"
    << "#pragma unroll 1\n"
    << "      for (int k = sub; k < W3; k += G) {\n"
    << "        double t = lead[k];\n"
    << "#pragma unroll 1\n"
    << "        for (int j = 1; j < G; ++j) t += lead[j * ROW + k];\n"
    << "        lead[k] = t;\n      }\n    }\n"
    << "    LDS_BARRIER();\n"
    << "    if (full && !single_fail && sub == 0) {\n"
    << reduce
    << "      if (bn_fail) for (int k = 0; k < W3; ++k) row[k] = kNaN;\n"
    << "    }\n"
    << "    LDS_BARRIER();\n"
    << "    STAGE_OUT(post_g);\n"
    << "    if (status_g && act && sub == 0) status_g[site] = single_fail ? 1 : (!full ? 0x80 : (bn_fail ? 2 : 0));\n"
    << "  }\n}\n";
  return s.str();
}

}  // namespace

std::string enumgen_source(const Model &m, int variant, int group_digits, bool call_mode, bool call_ct_out) {
  int cap = (group_digits == 0 && variant < 2) ? 7 : 6;  // see kEnumVariants
  if (const char *e = std::getenv("FAMSEQ_LANE_CAP")) cap = std::atoi(e);  // tuning aid
  const Shape s = choose_shape(m, cap);
  if (s.unrolled.empty()) throw std::runtime_error("enumeration codegen: empty unrolled set");
  if (group_digits < 0 || group_digits > std::min<int>(kEnumMaxGroupDigits, (int)s.outer.size()))
    throw std::runtime_error("enumeration codegen: more group digits than looped members");
  const int bt = enumgen_block_threads(m, group_digits);
  // The lane's LDS row: 3N doubles padded to an odd count, plus — while two workgroups per CU still
  // fit in the 160 KB — room for the likelihoods of unrolled members whose tables are rebuilt inside
  // the loops (otherwise re-read from global memory there: L2 misses that show up as HBM traffic).
  int row_len = (3 * m.n_members) | 1;
  {
    int looped_tables = 0;  // unrolled members with a looped parent
    for (int p : s.unrolled)
      if (m.mother[p] >= 0 && (s.upos[m.mother[p]] < 0 || s.upos[m.father[p]] < 0)) ++looped_tables;
    const int used = 6 * (int)s.outer.size() <= row_len ? 6 * (int)s.outer.size() : 3 * (int)s.outer.size();
    const int want = (used + 3 * looped_tables) | 1;
    // odd, two workgroups per CU (the call-path form also keeps a byte per member and lane, and two small tables)
    const int fit = ((160 * 1024 / 2 - 432 * 8 - (call_mode ? bt * m.n_members + 256 + 2064 : 0)) / (bt * 8) - 1) | 1;
    if (want > row_len) row_len = std::min(want, std::max(row_len, fit));
  }
  // An experiment that is OFF (FAMSEQ_LANE_LATE=1 turns it on): the sum-product kernel's order of phases for
  // the enumeration too — the whole computation first (marginals to registers), the next chunk requested,
  // then the two outputs — instead of single posterior / store / enumeration / store: one barrier fewer and
  // the prefetch in flight through both output phases; the row keeps the likelihoods until the end, scratch
  // slots follow them.  Measured slower (8 M sites, tools/kernel_bench, two runs): trio 0.368-0.374 -> 0.385-0.392 ms,
  // quad 0.604-0.613 -> 0.640-0.655, 5 members 0.805-0.816 -> 0.827-0.829 (identical binaries differ by +-4 %
  // between runs on these boxes).
  bool late = false;
  if (const char *e = std::getenv("FAMSEQ_LANE_LATE")) late = std::atoi(e) != 0 && group_digits == 0;  // tuning aid
  int scratch_len = 0;
  if (late) {
    int looped_tables = 0;
    for (int p : s.unrolled)
      if (m.mother[p] >= 0 && (s.upos[m.mother[p]] < 0 || s.upos[m.father[p]] < 0)) ++looped_tables;
    const int fit = ((160 * 1024 / 2 - 432 * 8 - (call_mode ? bt * m.n_members + 256 + 2064 : 0)) / (bt * 8) - 1) | 1;
    const int w3 = 3 * m.n_members, room = fit - w3;
    if (room < 3 * (int)s.outer.size()) late = false;  // not even the looped members' accumulators fit behind the row
    else {
      const int used = 6 * (int)s.outer.size() <= room ? 6 * (int)s.outer.size() : 3 * (int)s.outer.size();
      scratch_len = std::min(room, used + 3 * looped_tables);
      row_len = (w3 + scratch_len) | 1;
    }
  }
  int group = 1;
  for (int k = 0; k < group_digits; ++k) group *= 3;
  std::string what = "3^N enumeration, lane per site, " + std::to_string(s.outer.size()) + " looped + " +
                     std::to_string(s.unrolled.size()) + " unrolled members, variant " + std::to_string(variant);
  if (group > 1)
    what = "3^N enumeration, " + std::to_string(group) + " lanes per site (" + std::to_string(group_digits) + " of " +
           std::to_string(s.outer.size()) + " looped members' digits on lanes), " + std::to_string(s.unrolled.size()) +
           " unrolled members, variant " + std::to_string(variant);
  // (the call-path form of a small pedigree's kernel: two waves per SIMD at least — with 512 registers to fill, its output stages'
  // batched loads took the five-member kernel from two waves to one, 0.156 -> 0.203 ms per 1 M sites; bounded, the variant
  // contest sees the spill and takes the leaner stage-out)
  int min_waves = group_digits == 0 ? (call_mode && m.n_members <= 6 ? 2 : 1) : bt / 128;
  if (const char *e = std::getenv("FAMSEQ_LANE_MINWAVES")) min_waves = std::atoi(e);  // tuning aid
  // Transmission entries through scalar loads, and the innermost loop's loads one step ahead (see Gen): the one-lane-per-site
  // forms of pedigrees that have looped members; the lanes-per-site forms keep the per-lane LDS table.  Round 3, measured
  // with tools/kernel_bench on one box (profiles/r03a/exp_scalar_tables.txt): ten members 10.40 -> 10.13-10.20 ms per 4 M sites,
  // fifteen 139.2 -> 138.4 ms per 262 k; the innermost loop loses 36 of its 39 LDS reads and 10 of its 11 waits (885 -> 881
  // instructions per 729 configurations) — the waits were a small part of what one wave per SIMD loses: at 1.21 instructions
  // per configuration in that loop and 1.33 overall the kernel runs at the issue rate a single wave sustains.
  bool scalar_t = true;
  int prefetch = 2;
  if (const char *e = std::getenv("FAMSEQ_LANE_ST")) scalar_t = std::atoi(e) != 0;  // tuning aid
  if (const char *e = std::getenv("FAMSEQ_LANE_PRE")) prefetch = std::atoi(e);     // tuning aid: 0 none, 1 table entries, 2 and LDS reads
  scalar_t = scalar_t && group_digits == 0 && !late && !s.outer.empty();
  Gen gen(m, s, late ? std::max(scratch_len, 1) : row_len, group_digits, late, scalar_t, prefetch);
  if (call_mode) what += ", call path";
  if (late)  // regs_l = false: the shell's compute-first flow; variant 0 / 1 as below
    return kernel_shell(m, "famseq_enum_lane", what + ", compute-first shell", gen.body(), bt, min_waves, /*regs_l=*/false, (variant & 1) != 0,
                        /*chrx_loop=*/false, row_len, call_mode, /*lane_body=*/true, call_ct_out);
  const bool fence_single = variant & 1;
  if (group > 1) {
    if (call_mode) throw std::runtime_error("enumeration codegen: the lanes-per-site form has no call path");
    const std::string body = gen.body();
    return grouped_shell(m, what, body, gen.reduce_body(), bt, min_waves, fence_single, row_len, group);
  }
  // regs_l: LDS-resident likelihoods measured 17% slower.  variant 0: the members of the single
  // posterior overlap, 1: fenced one from the other (fewer registers)
  return kernel_shell(m, "famseq_enum_lane", what, gen.body(), bt, min_waves, /*regs_l=*/true, fence_single,
                      /*chrx_loop=*/scalar_t, row_len, call_mode, /*lane_body=*/false, call_ct_out);
}

}  // namespace famseq
