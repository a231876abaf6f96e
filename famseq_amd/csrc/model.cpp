// model.cpp — one-time pedigree model setup behind the C ABI (host only, no device code).
//
// Replaces, for the BN path, what the reference does in
//   family::family(mem, mRate)   /root/reference/src/family.cpp:78-127   (default priors)
//   family::setPCP / calPCP2*    /root/reference/src/family.cpp:259-289, 383-550
//   family::setRelation          /root/reference/src/family.cpp:291-350
//   family::checkPed             /root/reference/src/family.cpp:204-219
// The tables are accumulated in the reference's order because the result is not
// bit-symmetric in (mother,father) and the posteriors are compared
// against the reference CPU output.
#include <cstring>
#include <initializer_list>

#include "famseq_hip.h"

namespace {

inline double &cell(double *T, int child, int mother, int father) { return T[child * 9 + mother * 3 + father]; }

// Genotype g carries alleles {lo[g], hi[g]}; a child built from alleles (a,b) has genotype a+b.
constexpr int lo[3] = {0, 0, 1};
constexpr int hi[3] = {0, 1, 1};

void autosome(double mu, double *T) {
  std::memset(T, 0, 27 * sizeof(double));
  if (mu == 0) {  // family.cpp:473-491: four equally likely haplotype pairs
    for (int m = 0; m < 3; ++m)
      for (int f = 0; f < 3; ++f) {
        const int kids[4] = {lo[m] + lo[f], lo[m] + hi[f], hi[m] + lo[f], hi[m] + hi[f]};
        for (int k : kids) cell(T, k, m, f) = cell(T, k, m, f) + 0.25;
      }
    return;
  }
  const double wrong = mu / 2;         // family.cpp:498, nAllele = 2
  const double right = (1 - mu) / 2;   // family.cpp:502
  for (int m = 0; m < 3; ++m)
    for (int f = 0; f < 3; ++f) {
      // Transmission weights per allele for the haplotype currently "selected" in each
      // parent.  The reference walks the four haplotype pairs by editing these two
      // vectors in place (family.cpp:501-545); with a homozygous parent the edits
      // overlap, so the sequence of writes is kept as is.
      double pm[2] = {wrong, wrong}, pf[2] = {wrong, wrong};
      auto deposit = [&] {
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b) cell(T, a + b, m, f) = cell(T, a + b, m, f) + pm[a] * pf[b];
      };
      pm[lo[m]] = right; pf[lo[f]] = right;                                         deposit();
      pf[lo[f]] = wrong; pf[hi[f]] = right;                                         deposit();
      pm[lo[m]] = wrong; pf[hi[f]] = wrong; pm[hi[m]] = right; pf[lo[f]] = right;   deposit();
      pf[lo[f]] = wrong; pf[hi[f]] = right;                                         deposit();
    }
}

void x_daughter(double mu, double *T) {  // family.cpp:383-416; a father is never het on X
  std::memset(T, 0, 27 * sizeof(double));
  const double u = 1.0 - mu;
  const double hom_same[3] = {u * u, 2 * mu * u, mu * mu};             // mother 0, father 0
  const double hom_diff[3] = {u * mu, u * u + mu * mu, u * mu};        // mother 0, father 2
  const double het_lo[3] = {u * u / 2 + mu * u / 2, mu * u + u * u / 2 + mu * mu / 2,
                            mu * mu / 2 + mu * u / 2};                  // mother 1, father 0
  for (int c = 0; c < 3; ++c) {
    cell(T, c, 0, 0) = hom_same[c];
    cell(T, c, 0, 2) = hom_diff[c];
    cell(T, c, 1, 0) = het_lo[c];
    cell(T, c, 1, 2) = het_lo[2 - c];
    cell(T, c, 2, 0) = hom_diff[c];
    cell(T, c, 2, 2) = hom_same[2 - c];
  }
}

void x_son(double mu, double *T) {  // family.cpp:418-445; hemizygous, never het
  std::memset(T, 0, 27 * sizeof(double));
  for (int f : {0, 2}) {
    cell(T, 0, 0, f) = 1 - mu; cell(T, 2, 0, f) = mu;
    cell(T, 0, 1, f) = 0.5;    cell(T, 2, 1, f) = 0.5;
    cell(T, 0, 2, f) = mu;     cell(T, 2, 2, f) = 1 - mu;
  }
}

}  // namespace

extern "C" void famseq_transmission_tables(double mrate, double *pcp2, double *pcp2Xf, double *pcp2Xm) {
  autosome(mrate, pcp2);
  x_daughter(mrate, pcp2Xf);
  x_son(mrate, pcp2Xm);
}

extern "C" int famseq_model_init(famseq_model *m, int32_t n, const int32_t *id, const int32_t *mother_id,
                                 const int32_t *father_id, const int32_t *gender, const uint8_t *sequenced,
                                 double mrate, double lc) {
  if (!m || !id || !mother_id || !father_id || !gender) return FAMSEQ_E_ARG;
  if (n < 1 || n > FAMSEQ_MAX_MEMBERS) return FAMSEQ_E_ARG;
  std::memset(m, 0, sizeof(*m));
  m->n_members = n;
  m->lc = lc;
  const double N[3] = {0.9985, 0.001, 0.0005}, K[3] = {0.45, 0.1, 0.45};
  const double XN[3] = {0.999, 0, 0.001}, XK[3] = {0.5, 0, 0.5};
  std::memcpy(m->genoProbN, N, sizeof N);
  std::memcpy(m->genoProbK, K, sizeof K);
  std::memcpy(m->genoProbXN, XN, sizeof XN);
  std::memcpy(m->genoProbXK, XK, sizeof XK);
  famseq_transmission_tables(mrate, m->pcp2, m->pcp2Xf, m->pcp2Xm);
  for (int i = 0; i < n; ++i) {
    m->gender[i] = gender[i];
    m->sequenced[i] = sequenced ? (sequenced[i] != 0) : 1;
    // setRelation scans every member without stopping: the last matching id wins
    int im = -1, ifa = -1;
    for (int j = 0; j < n; ++j) {
      if (mother_id[i] == id[j]) im = j;
      if (father_id[i] == id[j]) ifa = j;
    }
    if ((im < 0) != (ifa < 0)) return FAMSEQ_E_PED_HALF;
    m->mother[i] = im;
    m->father[i] = ifa;
  }
  for (int i = 0; i < n; ++i)
    if (m->mother[i] >= 0 && (m->gender[m->mother[i]] != 2 || m->gender[m->father[i]] != 1))
      return FAMSEQ_E_PED_SEX;
  return 0;
}

// get_postRlt (family.cpp:636-665): the arg-max with strict '<' from -1, i.e. the LOWEST genotype among equals.  Equals are
// common where they are exact in the reference — a child of a 0/0 x 0/1 couple without data, at mutation rate 0: 0.5 / 0.5 from
// identical terms in identical order — and only equal to rounding here (another order of the same sum: 1e-15 relative).  So that
// the call is the reference's there too, a genotype within kCallTie (relative) of the largest posterior counts as equal to it.
// The device kernels use the same rule (elim_codegen.cpp ARGMAX3, io_kernels.hip).
extern "C" void famseq_call_genotypes(const double *post, int64_t n_rows, int8_t *geno) {
  for (int64_t r = 0; r < n_rows; ++r) {
    const double *p = post + 3 * r;
    double best = -1;
    for (int g = 0; g < 3; ++g)
      if (best < p[g]) best = p[g];
    const double thr = best * (1.0 - 1e-12);
    geno[r] = best < 0 ? (int8_t)-1 : (p[0] >= thr ? (int8_t)0 : (p[1] >= thr ? (int8_t)1 : (int8_t)2));  // (a NaN row leaves best at -1)
  }
}

extern "C" int famseq_pedigree_init(famseq_pedigree *p, int32_t n, const int32_t *id, const int32_t *mother_id,
                                    const int32_t *father_id, const int32_t *gender, const uint8_t *sequenced, double mrate,
                                    double lc, int32_t *mother_idx, int32_t *father_idx) {
  if (!p || !id || !mother_id || !father_id || !gender || !mother_idx || !father_idx) return FAMSEQ_E_ARG;
  if (n < 1) return FAMSEQ_E_ARG;
  std::memset(p, 0, sizeof(*p));
  p->n_members = n;
  p->lc = lc;
  const double N[3] = {0.9985, 0.001, 0.0005}, K[3] = {0.45, 0.1, 0.45};
  const double XN[3] = {0.999, 0, 0.001}, XK[3] = {0.5, 0, 0.5};
  std::memcpy(p->genoProbN, N, sizeof N);
  std::memcpy(p->genoProbK, K, sizeof K);
  std::memcpy(p->genoProbXN, XN, sizeof XN);
  std::memcpy(p->genoProbXK, XK, sizeof XK);
  famseq_transmission_tables(mrate, p->pcp2, p->pcp2Xf, p->pcp2Xm);
  p->gender = gender;
  p->sequenced = sequenced;
  p->mother = mother_idx;
  p->father = father_idx;
  for (int i = 0; i < n; ++i) {
    // setRelation scans every member without stopping: the last matching id wins (family.cpp:291-350)
    int im = -1, ifa = -1;
    for (int j = 0; j < n; ++j) {
      if (mother_id[i] == id[j]) im = j;
      if (father_id[i] == id[j]) ifa = j;
    }
    if ((im < 0) != (ifa < 0)) return FAMSEQ_E_PED_HALF;
    mother_idx[i] = im;
    father_idx[i] = ifa;
  }
  for (int i = 0; i < n; ++i)  // checkPed (family.cpp:204-219)
    if (mother_idx[i] >= 0 && (gender[mother_idx[i]] != 2 || gender[father_idx[i]] != 1)) return FAMSEQ_E_PED_SEX;
  return 0;
}
