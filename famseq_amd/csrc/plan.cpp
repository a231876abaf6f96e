// plan.cpp — builds the enumeration plan (see plan.h for the decomposition).
#include "plan.h"

#include <algorithm>
#include <sstream>
#include <stdexcept>

namespace famseq {

namespace {

int pow3(int e) {
  int r = 1;
  while (e-- > 0) r *= 3;
  return r;
}

size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

template <class T>
void json_array(std::ostringstream &o, const char *name, const std::vector<T> &v, bool comma = true) {
  o << "\"" << name << "\":[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << (long long)v[i];
  o << "]" << (comma ? "," : "");
}

}  // namespace

LdsLayout lds_layout(const Plan &p) {
  LdsLayout l{};
  const size_t bins = size_t(p.teams_per_block) * 3 * p.N;
  l.tc = align16(4 * 4 * 27 * sizeof(double));
  l.laneoff = align16(size_t(p.row_stride) * p.team_lanes * sizeof(uint32_t));
  l.lk = align16(bins * sizeof(double));
  l.flags = align16(size_t(p.teams_per_block) * 4 * sizeof(int32_t) + FAMSEQ_MAX_MEMBERS * sizeof(int32_t));
  l.red = align16(size_t(p.cols) * p.block_threads * sizeof(double));
  l.part = align16(bins * p.parts * sizeof(double));
  l.bins = align16(bins * sizeof(double));
  l.total = l.tc + l.laneoff + l.lk + l.flags + l.red + l.part + l.bins;
  return l;
}

Plan build_plan(const Model &m, const PlanOptions &opt) {
  const int N = m.n_members;
  if (N < 1 || N > FAMSEQ_MAX_MEMBERS) throw std::runtime_error("n_members out of range");
  std::vector<int> nchild(N, 0);
  for (int i = 0; i < N; ++i) {
    const int mo = m.mother[i], fa = m.father[i];
    if ((mo < 0) != (fa < 0)) throw std::runtime_error("member with exactly one parent");
    if (mo >= N || fa >= N || mo == i || fa == i) throw std::runtime_error("parent index out of range");
    if (mo >= 0) {
      nchild[mo]++;
      nchild[fa]++;
    }
  }
  Plan p;
  p.N = N;
  p.kind.resize(N);
  for (int i = 0; i < N; ++i) {
    const bool male = m.gender[i] == 1;
    p.kind[i] = m.mother[i] < 0 ? (male ? kFounderMale : kFounderFemale) : (male ? kChildMale : kChildFemale);
  }

  // ---- low members: childless, at most kMaxLow, taken from the end of the PED order
  std::vector<int> childless;
  for (int i = 0; i < N; ++i)
    if (nchild[i] == 0) childless.push_back(i);
  int L = std::min<int>(kMaxLow, childless.size());
  if (opt.low_members > 0) L = std::min(L, opt.low_members);
  if (L < 1) throw std::runtime_error("pedigree without a childless member (parent cycle)");
  p.L = L;
  p.low_member.assign(childless.end() - L, childless.end());
  std::vector<char> is_low(N, 0);
  for (int i : p.low_member) is_low[i] = 1;

  // ---- high members: J of them are iterated, A sit on lane digits
  std::vector<int> high;
  for (int i = 0; i < N; ++i)
    if (!is_low[i]) high.push_back(i);
  const int H = high.size();
  // Measured on MI355X (ped10): per-site fixed costs (cross-lane reduction, barriers) dominate with
  // wide teams, so by default a site gets at most 81 lanes and iterates over the rest.
  int A = std::min(H, 4);
  if (opt.fixed_digits >= 0) A = std::min(H, std::min(opt.fixed_digits, kMaxFixed));
  const int J = H - A;
  if (J > kIterLevels * kIterDigitsPerLevel) throw std::runtime_error("too many iterated members");
  p.A = A;
  p.J = J;
  p.team_lanes = pow3(A);
  // Which high members are iterated?  An iterated digit forces every factor that mentions it to
  // be re-evaluated at every step, and if it is the parent of a low member the low factors can
  // no longer be hoisted out of the step loop.  Greedy: add, J times, the member whose addition
  // keeps (B-list length, low dependence) cheapest.
  {
    std::vector<char> in_iter(N, 0);
    auto cost = [&]() {
      int nb = 0, lowdep = 0;
      for (int i = 0; i < N; ++i) {
        const bool dep = in_iter[i] || (m.mother[i] >= 0 && (in_iter[m.mother[i]] || in_iter[m.father[i]]));
        if (!dep) continue;
        if (is_low[i]) lowdep = 1; else nb++;
      }
      return 3 * nb + (lowdep ? 4 * L : 0);
    };
    for (int q = 0; q < J; ++q) {
      int best = -1, best_cost = 1 << 30;
      for (int i : high) {
        if (in_iter[i]) continue;
        in_iter[i] = 1;
        const int c = cost() * 64 + nchild[i];  // tie-break: fewer children
        in_iter[i] = 0;
        if (c < best_cost) best_cost = c, best = i;
      }
      in_iter[best] = 1;
    }
    for (int i : high) (in_iter[i] ? p.iter_member : p.fixed_member).push_back(i);
  }
  std::vector<int> fixed_pos(N, -1), iter_pos(N, -1);
  for (int q = 0; q < A; ++q) fixed_pos[p.fixed_member[q]] = q;
  for (int q = 0; q < J; ++q) iter_pos[p.iter_member[q]] = q;

  p.block_threads = p.team_lanes > 256 ? 768 : 256;
  if (opt.block_threads > 0) {
    if (opt.block_threads % 64 || opt.block_threads > 768 || opt.block_threads < p.team_lanes)
      throw std::runtime_error("block_threads must be a multiple of 64, <= 768 and >= 3^A");
    p.block_threads = opt.block_threads;
  }
  p.teams_per_block = p.block_threads / p.team_lanes;

  // ---- iter levels
  p.jlevels = (J + kIterDigitsPerLevel - 1) / kIterDigitsPerLevel;
  for (int l = 0; l < kIterLevels; ++l) {
    p.jd[l] = std::max(0, std::min(kIterDigitsPerLevel, J - l * kIterDigitsPerLevel));
    p.jn[l] = pow3(p.jd[l]);
  }

  // ---- slots: A list, B list, low list
  auto touches_iter = [&](int i) {
    if (iter_pos[i] >= 0) return true;
    return m.mother[i] >= 0 && (iter_pos[m.mother[i]] >= 0 || iter_pos[m.father[i]] >= 0);
  };
  std::vector<int> listA, listB;
  for (int i = 0; i < N; ++i) {
    if (is_low[i]) continue;
    (touches_iter(i) ? listB : listA).push_back(i);
  }
  p.nA = listA.size();
  p.nB = listB.size();
  p.slot_member = listA;
  p.slot_member.insert(p.slot_member.end(), p.low_member.begin(), p.low_member.end());
  p.slot_member.insert(p.slot_member.end(), listB.begin(), listB.end());
  p.n_slots = p.slot_member.size();
  p.row_stride = p.n_slots | 1;
  p.low_invariant = 1;
  for (int i : p.low_member)
    if (touches_iter(i)) p.low_invariant = 0;

  // ---- packed byte offsets: lane part per (lane, entry), step part per (level, step, step slot)
  p.laneoff.assign(size_t(p.team_lanes) * p.row_stride, 0);
  p.joff.assign(size_t(kIterLevels) * kIterTab * kStepSlots, 0);
  p.jdigits.assign(size_t(kIterLevels) * kIterTab, 0);
  for (int s = 0; s < p.n_slots; ++s) {
    const int i = p.slot_member[s];
    struct Dep { int who, tcoef, lkcoef; };
    std::vector<Dep> deps;
    if (!is_low[i]) deps.push_back({i, 9, 1});  // a low member's own digit is the unrolled loop index
    if (m.mother[i] >= 0) {
      deps.push_back({m.mother[i], 3, 0});
      deps.push_back({m.father[i], 1, 0});
    }
    for (int t = 0; t < p.team_lanes; ++t) {
      uint32_t tidx = p.kind[i] * 27, lkidx = 3 * i;
      for (const Dep &d : deps) {
        const int pos = fixed_pos[d.who];
        if (pos < 0) continue;
        const int dig = (t / pow3(pos)) % 3;
        tidx += d.tcoef * dig;
        lkidx += d.lkcoef * dig;
      }
      p.laneoff[size_t(t) * p.row_stride + s] = (8 * lkidx) << 16 | (8 * tidx);
    }
    // step record position: low member k -> k, B-list entry b -> L + b; A-list entries have none
    int rec = -1;
    if (s >= p.nA && s < p.nA + L) rec = s - p.nA;
    if (s >= p.nA + L) rec = L + (s - p.nA - L);
    if (rec < 0) continue;
    for (int l = 0; l < kIterLevels; ++l)
      for (int jl = 0; jl < p.jn[l]; ++jl) {
        uint32_t tidx = 0, lkidx = 0;
        for (const Dep &d : deps) {
          const int q = iter_pos[d.who];
          if (q < 0 || q / kIterDigitsPerLevel != l) continue;
          const int dig = (jl / pow3(q % kIterDigitsPerLevel)) % 3;
          tidx += d.tcoef * dig;
          lkidx += d.lkcoef * dig;
        }
        p.joff[(size_t(l) * kIterTab + jl) * kStepSlots + rec] = (8 * lkidx) << 16 | (8 * tidx);
      }
  }
  for (int l = 0; l < kIterLevels; ++l)
    for (int jl = 0; jl < p.jn[l]; ++jl) {
      uint16_t v = 0;
      for (int d = 0; d < p.jd[l]; ++d) v |= uint16_t(((jl / pow3(d)) % 3) << (2 * d));
      p.jdigits[size_t(l) * kIterTab + jl] = v;
    }

  // ---- reduction
  p.cols = 3 * L + 3 * J + 1;
  p.bin_kind.assign(N, 0);
  p.bin_index.assign(N, 0);
  for (int k = 0; k < L; ++k) p.bin_kind[p.low_member[k]] = 0, p.bin_index[p.low_member[k]] = k;
  for (int q = 0; q < J; ++q) p.bin_kind[p.iter_member[q]] = 1, p.bin_index[p.iter_member[q]] = q;
  for (int q = 0; q < A; ++q) p.bin_kind[p.fixed_member[q]] = 2, p.bin_index[p.fixed_member[q]] = q;
  p.parts = std::max(1, std::min(16, p.team_lanes / (3 * N)));
  p.lds_bytes = lds_layout(p).total;
  if (p.lds_bytes > 160 * 1024) throw std::runtime_error("plan needs more than 160 KiB of LDS");
  return p;
}

std::string Plan::json() const {
  std::ostringstream o;
  o << "{\"N\":" << N << ",\"L\":" << L << ",\"A\":" << A << ",\"J\":" << J << ",\"team_lanes\":" << team_lanes
    << ",\"block_threads\":" << block_threads << ",\"teams_per_block\":" << teams_per_block << ",\"nA\":" << nA
    << ",\"nB\":" << nB << ",\"n_slots\":" << n_slots << ",\"row_stride\":" << row_stride << ",\"low_invariant\":" << low_invariant << ",\"step_slots\":" << kStepSlots << ",\"jlevels\":" << jlevels << ",\"jn\":[" << jn[0] << ","
    << jn[1] << "," << jn[2] << "],\"jd\":[" << jd[0] << "," << jd[1] << "," << jd[2] << "],\"cols\":" << cols
    << ",\"parts\":" << parts << ",\"lds_bytes\":" << lds_bytes << ",\"iter_tab\":" << kIterTab << ",";
  json_array(o, "low_member", low_member);
  json_array(o, "fixed_member", fixed_member);
  json_array(o, "iter_member", iter_member);
  json_array(o, "slot_member", slot_member);
  json_array(o, "kind", kind);
  json_array(o, "bin_kind", bin_kind);
  json_array(o, "bin_index", bin_index);
  json_array(o, "laneoff", laneoff);
  json_array(o, "jdigits", jdigits);
  json_array(o, "joff", joff, false);
  o << "}";
  return o.str();
}

// Device image (32-bit words): [laneoff: team_lanes*row_stride][joff: 3*243*20][jdigits: 3*243]
// [member info: N words = kind | sequenced<<2 (filled by the caller) | bin_kind<<3 | bin_index<<5]
std::vector<uint32_t> Plan::device_image() const {
  std::vector<uint32_t> img;
  img.insert(img.end(), laneoff.begin(), laneoff.end());
  while (img.size() % 4) img.push_back(0);  // step records are fetched as 16-byte vectors
  img.insert(img.end(), joff.begin(), joff.end());
  for (uint16_t v : jdigits) img.push_back(v);
  for (int i = 0; i < N; ++i) img.push_back(uint32_t(kind[i]) | uint32_t(bin_kind[i]) << 3 | uint32_t(bin_index[i]) << 5);
  return img;
}

}  // namespace famseq
