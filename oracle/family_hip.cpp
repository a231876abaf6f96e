// oracle/family_hip.cpp — the link-time drop-in, compiled for real (TEST INFRASTRUCTURE ONLY).
//
// The reference chooses its implementation of `class family` at link time
// (/root/reference/src/makefile:4 links family.cpp, makefile.gpu:83-87 links family.cu).  This
// translation unit is the third choice a FamSeq maintainer would add: it defines
//     bool family::calPostProbBN(bool Known, int chrType)        family.h:375
// by forwarding the member's likelihood matrix to libfamseq_hip.so through the C ABI
// (include/famseq_hip.h) and filling postProb / postProbSingle / flagPB / flagPBS the way
// family.cpp:750-1124 leaves them.  oracle/Makefile builds _ref/FamSeq_hipref from the reference's
// OWN FamSeq.cpp, file.cpp, checkInput.cpp, normal.cpp and family.cpp (compiled where they lie),
// with that one symbol of family.o weakened (objcopy --weaken-symbol) so that this definition
// wins; everything else in class family — set_LK, the getters, init(), the tables — is the
// reference's code.  tests/test_dropin_gpu.py runs the result under the reference's own
// callGenoMVCF / callGenoLK (file.cpp:595->607->680, :1743->1751->1804) and diffs its output
// against the reference CLI's.  The reference's header is found through -I; nothing of it is copied.
//
// One launch per site (the reference's drivers call the operator per site) — this proves the
// boundary, not the throughput; the batched driver is bin/FamSeq.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "family.h"       // the reference's, via -I$(REF)/src
#include "famseq_hip.h"

namespace {

struct Bound {
  famseq_ctx *ctx = nullptr;
};

// `family` objects are passed BY VALUE into the drivers (file.h:55,64) and copied freely, so the
// device context cannot live in the object: it is looked up by everything the model depends on.
std::map<std::string, Bound> &cache() {
  static std::map<std::string, Bound> c;
  return c;
}

void put(std::string &k, const void *p, size_t n) { k.append(static_cast<const char *>(p), n); }

}  // namespace

bool family::calPostProbBN(bool Known, int chrType) {
  if (!flagLK) {  // family.cpp:752-756
    std::cout << "Likelihood has not been set. Please set likelihood first." << std::endl;
    return false;
  }
  const int n = (int)numInd;
  famseq_model m;
  std::memset(&m, 0, sizeof m);
  m.n_members = n;
  for (int i = 0; i < n; ++i) {
    m.mother[i] = parent[i].empty() ? -1 : parent[i][0];  // setRelation's result, family.cpp:329-330
    m.father[i] = parent[i].empty() ? -1 : parent[i][1];
    m.gender[i] = member[i].get_gender();
    m.sequenced[i] = 0;
  }
  for (size_t v = 0; v < mapV2P.size(); ++v)
    if (mapV2P[v] >= 0) m.sequenced[mapV2P[v]] = 1;  // the shortcut vote's member set, family.cpp:768-771
  for (int c = 0; c < 3; ++c)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        m.pcp2[9 * c + 3 * a + b] = pcp2[c](a, b);  // the tables this very object computed in init()
        m.pcp2Xf[9 * c + 3 * a + b] = pcp2Xf[c](a, b);
        m.pcp2Xm[9 * c + 3 * a + b] = pcp2Xm[c](a, b);
      }
  for (int g = 0; g < 3; ++g) {
    m.genoProbN[g] = genoProbN[g];
    m.genoProbK[g] = genoProbK[g];
    m.genoProbXN[g] = genoProbXN[g];
    m.genoProbXK[g] = genoProbXK[g];
  }
  m.lc = m_lc;
  std::string key;
  put(key, &m, sizeof m);
  Bound &b = cache()[key];
  if (!b.ctx) {
    char err[512] = "";
    b.ctx = famseq_create(&m, 0, err, sizeof err);
    if (!b.ctx) {  // family.cu:1153-1167: device trouble goes to stderr, the site is not calculated
      std::fprintf(stderr, "family_hip: famseq_create: %s\n", err);
      return false;
    }
  }
  std::vector<double> lk(3 * n), post(3 * n), single(3 * n);
  for (int i = 0; i < n; ++i)
    for (int g = 0; g < 3; ++g) lk[3 * i + g] = likelihood(i, g);
  const uint8_t flags = (Known ? FAMSEQ_FLAG_KNOWN : 0) | (chrType == 1 ? FAMSEQ_FLAG_CHRX : 0);
  uint8_t status = 0;
  const int rc = famseq_bn_batch(b.ctx, 1, lk.data(), &flags, post.data(), single.data(), &status);
  if (rc != 0) {
    std::fprintf(stderr, "family_hip: famseq_bn_batch: %s\n", famseq_last_error(b.ctx));
    return false;
  }
  if ((status & 3) == FAMSEQ_ST_SINGLE_FAIL) return false;  // calPostProbSingle returned false, family.cpp:758-763
  for (int i = 0; i < n; ++i)
    for (int g = 0; g < 3; ++g) postProbSingle(i, g) = single[3 * i + g];
  flagPBS = true;  // family.cpp:1497
  if ((status & 3) == FAMSEQ_ST_BN_FAIL) return false;  // a row sum <= 0, family.cpp:946 / :1111
  for (int i = 0; i < n; ++i)
    for (int g = 0; g < 3; ++g) postProb(i, g) = post[3 * i + g];
  flagPB = true;  // family.cpp:876 / :1122
  return true;
}
