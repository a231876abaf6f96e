// model.h — the library's own, size-independent copy of what `family` holds after setFam() + init()
// (/root/reference/src/file.cpp:1888-1927; family.cpp:78-127, 221-238).  Both public structs convert to it:
// famseq_model (fixed arrays, N <= FAMSEQ_MAX_MEMBERS: every engine) and famseq_pedigree (caller-owned
// arrays of any length: the reference's CPU code has no member cap — std::vector state throughout, odometer
// family.cpp:894-941, peeling :1126-1403 — so neither has the engine that serves its -method 2 domain here).
#ifndef FAMSEQ_MODEL_H_
#define FAMSEQ_MODEL_H_

#include <cstdint>
#include <cstring>
#include <vector>

#include "famseq_hip.h"

namespace famseq {

struct Model {
  int32_t n_members = 0;
  std::vector<int32_t> mother, father, gender;  // parent[i][0], parent[i][1] (index or -1); 1 = male
  std::vector<uint8_t> sequenced;
  double pcp2[27] = {}, pcp2Xf[27] = {}, pcp2Xm[27] = {};
  double genoProbN[3] = {}, genoProbK[3] = {}, genoProbXN[3] = {}, genoProbXK[3] = {};
  double lc = 1.0;

  Model() = default;
  Model(const famseq_model &m) {  // NOLINT: implicit on purpose, the fixed struct is the common caller
    const int n = m.n_members < 0 ? 0 : (m.n_members > FAMSEQ_MAX_MEMBERS ? FAMSEQ_MAX_MEMBERS : m.n_members);
    n_members = m.n_members;
    mother.assign(m.mother, m.mother + n);
    father.assign(m.father, m.father + n);
    gender.assign(m.gender, m.gender + n);
    sequenced.assign(m.sequenced, m.sequenced + n);
    tables(m);
  }
  Model(const famseq_pedigree &p) {  // NOLINT
    n_members = p.n_members;
    const int n = p.n_members < 0 ? 0 : p.n_members;
    if (p.mother && p.father && p.gender) {
      mother.assign(p.mother, p.mother + n);
      father.assign(p.father, p.father + n);
      gender.assign(p.gender, p.gender + n);
    } else {
      n_members = -1;  // rejected by validate_model
    }
    if (p.sequenced) sequenced.assign(p.sequenced, p.sequenced + n);
    else sequenced.assign(n, 1);
    tables(p);
  }

 private:
  template <class T>
  void tables(const T &s) {
    std::memcpy(pcp2, s.pcp2, sizeof pcp2);
    std::memcpy(pcp2Xf, s.pcp2Xf, sizeof pcp2Xf);
    std::memcpy(pcp2Xm, s.pcp2Xm, sizeof pcp2Xm);
    std::memcpy(genoProbN, s.genoProbN, sizeof genoProbN);
    std::memcpy(genoProbK, s.genoProbK, sizeof genoProbK);
    std::memcpy(genoProbXN, s.genoProbXN, sizeof genoProbXN);
    std::memcpy(genoProbXK, s.genoProbXK, sizeof genoProbXK);
    lc = s.lc;
  }
};

}  // namespace famseq
#endif
