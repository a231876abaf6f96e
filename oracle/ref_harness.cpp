/*
 * ref_harness.cpp — C entry points over the *compiled reference* `class family`
 * (TEST INFRASTRUCTURE ONLY).
 *
 * This file is our own code.  It includes the reference's public headers from
 * /root/reference/src (never copied into this repo) and is linked with the
 * reference's family.cpp compiled in place by oracle/Makefile.  The result,
 * oracle/_ref/libfamseq_ref.so, is git-ignored; it is used to
 *   (1) pin oracle/bn_oracle.c bit-for-bit (oracle/gen_golden.py), and
 *   (2) optionally serve as the "reference" CPU baseline in bench.py.
 *
 * Calls made (reference interface, family.h): family(mem, mRate) :225,
 * set_lc :356, init :232, set_mapV2P :353, set_LK :344, calPostProbBN :375,
 * calPostProbPeeling :368, get_postProb(false) :262, get_postProbSingle(false) :265,
 * get_flagPBS :295, get_pcp2/get_pcp2Xf/get_pcp2Xm :322-328.
 */
#include <cmath>
#include <cstdint>
#include <vector>

#include "family.h"

namespace {
struct RefFam {
  family fam;
  int n;
};

void fill(double *p, int n, double v) {
  for (int i = 0; i < n; i++) p[i] = v;
}
}  // namespace

extern "C" {

/* returns NULL if family::init() rejects the pedigree */
void *famref_create(int n, const int32_t *id, const int32_t *mid, const int32_t *fid,
                    const int32_t *gender, const uint8_t *sequenced, double mrate, double lc,
                    const double *gN, const double *gK, const double *gXN, const double *gXK) {
  std::vector<individual> mem;
  for (int i = 0; i < n; i++) mem.push_back(individual(id[i], gender[i], 0, 0, mid[i], fid[i], "s"));
  RefFam *h = new RefFam;
  h->n = n;
  h->fam = family(mem, mrate);
  if (gN) h->fam.set_genoProbN(std::vector<double>(gN, gN + 3));
  if (gK) h->fam.set_genoProbK(std::vector<double>(gK, gK + 3));
  if (gXN) h->fam.set_genoProbXN(std::vector<double>(gXN, gXN + 3));
  if (gXK) h->fam.set_genoProbXK(std::vector<double>(gXK, gXK + 3));
  h->fam.set_lc(lc);
  if (!h->fam.init()) {
    delete h;
    return 0;
  }
  std::vector<int> v2p;
  for (int i = 0; i < n; i++)
    if (!sequenced || sequenced[i]) v2p.push_back(i);
  h->fam.set_mapV2P(v2p);
  return h;
}

void famref_destroy(void *hh) { delete static_cast<RefFam *>(hh); }

/* [child*9 + mother*3 + father] */
void famref_tables(void *hh, double *pcp2, double *xf, double *xm) {
  RefFam *h = static_cast<RefFam *>(hh);
  std::vector<dMatrix<double> > a = h->fam.get_pcp2(), b = h->fam.get_pcp2Xf(), c = h->fam.get_pcp2Xm();
  for (int ch = 0; ch < 3; ch++)
    for (int mo = 0; mo < 3; mo++)
      for (int fa = 0; fa < 3; fa++) {
        pcp2[ch * 9 + mo * 3 + fa] = a[ch](mo, fa);
        xf[ch * 9 + mo * 3 + fa] = b[ch](mo, fa);
        xm[ch * 9 + mo * 3 + fa] = c[ch](mo, fa);
      }
}

/* method: 1 = calPostProbBN, 2 = calPostProbPeeling.
 * status: 0 ok; 1 single posterior failed; 2 BN/peeling failed.
 * bit7 is set when post == single bit-for-bit (the reference does not expose
 * which branch ran; equality is what the shortcut branch produces). */
int famref_site(void *hh, int method, const double *lk, int known, int chrx, double *post,
                double *single) {
  RefFam *h = static_cast<RefFam *>(hh);
  const int n = h->n;
  dMatrix<double> m(n, 3, 1);
  for (int i = 0; i < n; i++)
    for (int g = 0; g < 3; g++) m(i, g) = lk[3 * i + g];
  if (!h->fam.set_LK(m)) return -1;
  bool ok = method == 2 ? h->fam.calPostProbPeeling(known != 0, chrx) : h->fam.calPostProbBN(known != 0, chrx);
  if (!ok) {
    fill(post, 3 * n, NAN);
    if (!h->fam.get_flagPBS()) {
      fill(single, 3 * n, NAN);
      return 1;
    }
    dMatrix<double> s = h->fam.get_postProbSingle(false);
    for (int i = 0; i < n; i++)
      for (int g = 0; g < 3; g++) single[3 * i + g] = s(i, g);
    return 2;
  }
  dMatrix<double> p = h->fam.get_postProb(false), s = h->fam.get_postProbSingle(false);
  bool same = true;
  for (int i = 0; i < n; i++)
    for (int g = 0; g < 3; g++) {
      post[3 * i + g] = p(i, g);
      single[3 * i + g] = s(i, g);
      if (!(p(i, g) == s(i, g))) same = false;
    }
  return same ? 0x80 : 0;
}

}  // extern "C"
