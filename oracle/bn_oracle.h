/*
 * bn_oracle.h — CPU oracle for the FamSeq `-method 1` pedigree posterior.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (family::calPostProbBN and the one-time table setup it depends
 * on).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product library (libfamseq_hip.so) never links, loads or
 * calls anything in oracle/.
 *
 * Parity pin: validated bit-for-bit against the compiled reference
 * (oracle/_ref/libfamseq_ref.so, built by oracle/Makefile from the sources in
 * /root/reference/src) on every TestData pedigree x VCF site / LK row and on
 * synthetic autosome / chrX / failure cases; see oracle/gen_golden.py and
 * tests/test_oracle_golden.py.
 */
#ifndef BN_ORACLE_H_
#define BN_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_MEMBERS 20

/* status byte per site (same encoding as the product ABI) */
#define ORACLE_ST_OK 0
#define ORACLE_ST_SINGLE_FAIL 1 /* calPostProbSingle returned false (family.cpp:1437) */
#define ORACLE_ST_BN_FAIL 2     /* BN row sum <= 0 (family.cpp:946, :1111) */
#define ORACLE_ST_SHORTCUT 0x80 /* took the -LRC single-sample branch (family.cpp:793) */

typedef struct {
  int32_t n;                              /* numInd */
  int32_t mother[ORACLE_MAX_MEMBERS];     /* parent[i][0] or -1 (family.cpp:329) */
  int32_t father[ORACLE_MAX_MEMBERS];     /* parent[i][1] or -1 (family.cpp:330) */
  int32_t gender[ORACLE_MAX_MEMBERS];     /* 1 = male */
  uint8_t sequenced[ORACLE_MAX_MEMBERS];  /* member appears in mapV2P */
  double pcp2[27], pcp2Xf[27], pcp2Xm[27]; /* [child*9 + mother*3 + father] */
  double genoProbN[3], genoProbK[3], genoProbXN[3], genoProbXK[3];
  double lc; /* m_lc */
} oracle_model;

/* Transmission tables for mutation rate mu (family.cpp:383-550). */
void oracle_tables(double mu, double *pcp2, double *pcp2Xf, double *pcp2Xm);

/* ctor + init(): default priors (family.cpp:78-127), tables, setRelation
 * (family.cpp:291-350), checkPed (family.cpp:204-219).
 * returns 0 ok, -1 half-parented member, -2 parent of wrong sex, -3 bad n. */
int oracle_model_init(oracle_model *m, int n, const int32_t *id, const int32_t *mid,
                      const int32_t *fid, const int32_t *gender, const uint8_t *sequenced,
                      double mu, double lc);

/* One site: lk[n][3] in PED order. Returns the status byte.  On status 1 both
 * outputs are NaN-filled, on status 2 `post` is NaN-filled. */
uint8_t oracle_bn_site(const oracle_model *m, const double *lk, int known, int chr_x,
                       double *post, double *single);

/* Batch; flags bit0 = Known, bit1 = chrX.  n_threads >= 1 (sites are sharded
 * contiguously over pthreads). */
void oracle_bn_batch(const oracle_model *m, int64_t n_sites, const double *lk,
                     const uint8_t *flags, double *post, double *single, uint8_t *status,
                     int n_threads);

/* get_postRlt (family.cpp:636-665): arg-max with strict '<' from -1. */
int oracle_argmax3(const double *row);

#ifdef __cplusplus
}
#endif
#endif
