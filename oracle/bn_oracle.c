/*
 * bn_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY, see bn_oracle.h).
 *
 * Restates, in plain C and in the reference's own arithmetic order, the
 * `-method 1` path of wwylab/FamSeq v1.0.3:
 *   family::calPostProbBN      /root/reference/src/family.cpp:750-1124
 *   family::calPostProbSingle  /root/reference/src/family.cpp:1405-1499
 *   family::calPCP2S/Xf/Xm     /root/reference/src/family.cpp:383-550
 *   family::setRelation        /root/reference/src/family.cpp:291-350
 *   family::checkPed           /root/reference/src/family.cpp:204-219
 *   family ctor default priors /root/reference/src/family.cpp:78-127
 * Every product (lk*prior, 1e7*f0*f1*..., row sums left to right) is formed
 * in the same order as the reference so results are bit-identical when
 * compiled without FMA contraction (oracle/Makefile uses -ffp-contract=off).
 */
#include "bn_oracle.h"

#include <math.h>
#include <pthread.h>
#include <string.h>

/* ---- one-time tables -------------------------------------------------- */

/* autosomal table: family.cpp:447-550 with nAllele = 2.
 * Genotype code g -> unordered allele pair: 0->(0,0) 1->(0,1) 2->(1,1);
 * allele pair (k,l) -> genotype k+l.  T[child*9 + mother*3 + father]. */
static void table_autosome(double mu, double *T) {
  static const int al[3][2] = {{0, 0}, {0, 1}, {1, 1}};
  memset(T, 0, 27 * sizeof(double));
  if (mu == 0) { /* family.cpp:473-491 */
    for (int mo = 0; mo < 3; mo++)
      for (int fa = 0; fa < 3; fa++)
        for (int a = 0; a < 2; a++)
          for (int b = 0; b < 2; b++) {
            int c = al[mo][a] + al[fa][b];
            T[c * 9 + mo * 3 + fa] = T[c * 9 + mo * 3 + fa] + 0.25;
          }
    return;
  }
  const double slip = mu / (2 * (2 - 1)); /* family.cpp:498 */
  const double keep = (1 - mu) / 2;       /* family.cpp:502 */
  for (int mo = 0; mo < 3; mo++) {
    for (int fa = 0; fa < 3; fa++) {
      double pm[2] = {slip, slip}, pf[2] = {slip, slip};
      /* four haplotype passes, assignments in the reference's sequence
       * (family.cpp:501-545); each pass adds pm[k]*pf[l] for k,l in 0..1 */
      for (int pass = 0; pass < 4; pass++) {
        switch (pass) {
          case 0:
            pm[al[mo][0]] = keep;
            pf[al[fa][0]] = keep;
            break;
          case 1:
            pf[al[fa][0]] = slip;
            pf[al[fa][1]] = keep;
            break;
          case 2:
            pm[al[mo][0]] = slip;
            pf[al[fa][1]] = slip;
            pm[al[mo][1]] = keep;
            pf[al[fa][0]] = keep;
            break;
          default:
            pf[al[fa][0]] = slip;
            pf[al[fa][1]] = keep;
            break;
        }
        for (int k = 0; k < 2; k++)
          for (int l = 0; l < 2; l++) {
            double *cell = &T[(k + l) * 9 + mo * 3 + fa];
            *cell = *cell + pm[k] * pf[l];
          }
      }
    }
  }
}

/* daughter on chrX: family.cpp:383-416 (father genotype 1 impossible -> 0) */
static void table_x_daughter(double mu, double *T) {
  const double u = 1.0 - mu;
  memset(T, 0, 27 * sizeof(double));
#define XF(c, mo, fa) T[(c) * 9 + (mo) * 3 + (fa)]
  XF(0, 0, 0) = u * u;
  XF(1, 0, 0) = 2 * mu * u;
  XF(2, 0, 0) = mu * mu;

  XF(0, 0, 2) = u * mu;
  XF(1, 0, 2) = u * u + mu * mu;
  XF(2, 0, 2) = u * mu;

  XF(0, 1, 0) = u * u / 2 + mu * u / 2;
  XF(1, 1, 0) = mu * u + u * u / 2 + mu * mu / 2;
  XF(2, 1, 0) = mu * mu / 2 + mu * u / 2;

  XF(0, 1, 2) = mu * mu / 2 + mu * u / 2;
  XF(1, 1, 2) = mu * u + u * u / 2 + mu * mu / 2;
  XF(2, 1, 2) = u * u / 2 + mu * u / 2;

  XF(0, 2, 0) = u * mu;
  XF(1, 2, 0) = u * u + mu * mu;
  XF(2, 2, 0) = u * mu;

  XF(0, 2, 2) = mu * mu;
  XF(1, 2, 2) = 2 * mu * u;
  XF(2, 2, 2) = u * u;
#undef XF
}

/* son on chrX: family.cpp:418-445 (hemizygous; never het; father columns 0,2) */
static void table_x_son(double mu, double *T) {
  memset(T, 0, 27 * sizeof(double));
  for (int fa = 0; fa < 3; fa += 2) {
    T[0 * 9 + 0 * 3 + fa] = 1 - mu;
    T[2 * 9 + 0 * 3 + fa] = mu;
    T[0 * 9 + 1 * 3 + fa] = 0.5;
    T[2 * 9 + 1 * 3 + fa] = 0.5;
    T[0 * 9 + 2 * 3 + fa] = mu;
    T[2 * 9 + 2 * 3 + fa] = 1 - mu;
  }
}

void oracle_tables(double mu, double *pcp2, double *pcp2Xf, double *pcp2Xm) {
  table_autosome(mu, pcp2);
  table_x_daughter(mu, pcp2Xf);
  table_x_son(mu, pcp2Xm);
}

int oracle_model_init(oracle_model *m, int n, const int32_t *id, const int32_t *mid,
                      const int32_t *fid, const int32_t *gender, const uint8_t *sequenced,
                      double mu, double lc) {
  if (n < 1 || n > ORACLE_MAX_MEMBERS) return -3;
  memset(m, 0, sizeof(*m));
  m->n = n;
  m->lc = lc;
  /* default priors, family.cpp:91-109 */
  m->genoProbN[0] = 0.9985; m->genoProbN[1] = 0.001; m->genoProbN[2] = 0.0005;
  m->genoProbK[0] = 0.45;   m->genoProbK[1] = 0.1;   m->genoProbK[2] = 0.45;
  m->genoProbXN[0] = 0.999; m->genoProbXN[1] = 0;    m->genoProbXN[2] = 0.001;
  m->genoProbXK[0] = 0.5;   m->genoProbXK[1] = 0;    m->genoProbXK[2] = 0.5;
  oracle_tables(mu, m->pcp2, m->pcp2Xf, m->pcp2Xm);
  for (int i = 0; i < n; i++) {
    m->gender[i] = gender[i];
    m->sequenced[i] = sequenced ? sequenced[i] : 1;
  }
  /* setRelation: the id scan does not stop at the first hit (last match wins) */
  for (int i = 0; i < n; i++) {
    int im = -1, ifa = -1;
    for (int j = 0; j < n; j++) {
      if (mid[i] == id[j]) im = j;
      if (fid[i] == id[j]) ifa = j;
    }
    if ((im < 0) != (ifa < 0)) return -1; /* family.cpp:319-323 */
    m->mother[i] = im;
    m->father[i] = ifa;
  }
  /* checkPed */
  for (int i = 0; i < n; i++) {
    if (m->mother[i] >= 0) {
      if (m->gender[m->mother[i]] != 2) return -2;
      if (m->gender[m->father[i]] != 1) return -2;
    }
  }
  return 0;
}

/* ---- per-site path ---------------------------------------------------- */

static void fill_nan(double *p, int n) {
  for (int i = 0; i < n; i++) p[i] = NAN;
}

/* prior used for member i (family.cpp:1052-1062, :1475-1485) */
static const double *prior_for(const oracle_model *m, int i, int known, int chr_x) {
  if (chr_x && m->gender[i] == 1) return known ? m->genoProbXK : m->genoProbXN;
  return known ? m->genoProbK : m->genoProbN;
}

/* lk*prior, row-normalised; returns 0 on a row sum <= 0 (family.cpp:1426-1445) */
static int single_posterior(const oracle_model *m, const double *lk, int known, int chr_x,
                            double *out) {
  const int n = m->n;
  for (int i = 0; i < n; i++) {
    const double *pr = prior_for(m, i, known, chr_x);
    for (int g = 0; g < 3; g++) out[3 * i + g] = lk[3 * i + g] * pr[g];
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int g = 0; g < 3; g++) s = s + out[3 * i + g]; /* dMatrix.h:153-162 */
    if (s <= 0) return 0;
    for (int g = 0; g < 3; g++) out[3 * i + g] = out[3 * i + g] / s;
  }
  return 1;
}

uint8_t oracle_bn_site(const oracle_model *m, const double *lk, int known, int chr_x,
                       double *post, double *single) {
  const int n = m->n;
  if (!single_posterior(m, lk, known, chr_x, single)) { /* family.cpp:758-763 */
    fill_nan(single, 3 * n);
    fill_nan(post, 3 * n);
    return ORACLE_ST_SINGLE_FAIL;
  }

  /* shortcut vote over sequenced members, family.cpp:767-789 */
  int all_sharp = 1;
  for (int i = 0; i < n && all_sharp; i++) {
    if (!m->sequenced[i]) continue;
    double big = 0, sum = 0;
    for (int g = 0; g < 3; g++) {
      if (big < lk[3 * i + g]) big = lk[3 * i + g];
      sum = sum + lk[3 * i + g];
    }
    big = big / sum;
    if (big < m->lc) all_sharp = 0;
  }

  if (all_sharp) { /* family.cpp:793-878: same formula as the single posterior */
    if (!single_posterior(m, lk, known, chr_x, post)) {
      fill_nan(post, 3 * n);
      return ORACLE_ST_BN_FAIL;
    }
    return ORACLE_ST_OK | ORACLE_ST_SHORTCUT;
  }

  /* full enumeration, family.cpp:882-941 (autosome) / :990-1106 (chrX) */
  int geno[ORACLE_MAX_MEMBERS];
  double term[ORACLE_MAX_MEMBERS];
  for (int i = 0; i < n; i++) geno[i] = 0;
  for (int i = 0; i < 3 * n; i++) post[i] = 0;
  for (;;) {
    for (int i = 0; i < n; i++) {
      const int g = geno[i];
      if (m->mother[i] < 0) {
        term[i] = prior_for(m, i, known, chr_x)[g] * lk[3 * i + g];
      } else {
        const double *T = !chr_x ? m->pcp2 : (m->gender[i] == 1 ? m->pcp2Xm : m->pcp2Xf);
        term[i] = T[g * 9 + geno[m->mother[i]] * 3 + geno[m->father[i]]] * lk[3 * i + g];
      }
    }
    double w = 10000000;
    for (int i = 0; i < n; i++) w = w * term[i];
    for (int i = 0; i < n; i++) post[3 * i + geno[i]] = post[3 * i + geno[i]] + w;
    int k = 0;
    while (k < n) { /* odometer, member 0 fastest */
      if (++geno[k] == 3) {
        geno[k] = 0;
        k++;
      } else {
        break;
      }
    }
    if (k == n) break;
  }
  for (int i = 0; i < n; i++) { /* family.cpp:943-954 */
    double s = 0;
    for (int g = 0; g < 3; g++) s = s + post[3 * i + g];
    if (s <= 0) {
      fill_nan(post, 3 * n);
      return ORACLE_ST_BN_FAIL;
    }
    for (int g = 0; g < 3; g++) post[3 * i + g] = post[3 * i + g] / s;
  }
  return ORACLE_ST_OK;
}

int oracle_argmax3(const double *row) {
  double best = -1;
  int ind = -1;
  for (int g = 0; g < 3; g++)
    if (best < row[g]) {
      best = row[g];
      ind = g;
    }
  return ind;
}

/* ---- batch driver ----------------------------------------------------- */

typedef struct {
  const oracle_model *m;
  int64_t lo, hi;
  const double *lk;
  const uint8_t *flags;
  double *post, *single;
  uint8_t *status;
} shard_t;

static void *run_shard(void *arg) {
  shard_t *s = (shard_t *)arg;
  const int w = 3 * s->m->n;
  for (int64_t i = s->lo; i < s->hi; i++) {
    const int f = s->flags ? s->flags[i] : 0;
    s->status[i] = oracle_bn_site(s->m, s->lk + i * w, f & 1, (f >> 1) & 1, s->post + i * w,
                                  s->single + i * w);
  }
  return 0;
}

void oracle_bn_batch(const oracle_model *m, int64_t n_sites, const double *lk,
                     const uint8_t *flags, double *post, double *single, uint8_t *status,
                     int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  if (n_threads > n_sites) n_threads = n_sites > 0 ? (int)n_sites : 1;
  shard_t sh[256];
  pthread_t th[256];
  for (int t = 0; t < n_threads; t++) {
    sh[t].m = m;
    sh[t].lo = n_sites * t / n_threads;
    sh[t].hi = n_sites * (t + 1) / n_threads;
    sh[t].lk = lk;
    sh[t].flags = flags;
    sh[t].post = post;
    sh[t].single = single;
    sh[t].status = status;
  }
  if (n_threads == 1) {
    run_shard(&sh[0]);
    return;
  }
  for (int t = 0; t < n_threads; t++) pthread_create(&th[t], 0, run_shard, &sh[t]);
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], 0);
}
