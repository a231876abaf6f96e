/* c_abi_check.c — the C ABI consumed from plain C99 (no C++, no Python): proves the header is
 * valid C, that every entry point links, and exercises the no-device error path or, on a GPU box,
 * one small batch.  Built and run by tests/test_c_abi.py. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "famseq_hip.h"

int main(void) {
  /* trio: father, mother, child */
  const int32_t id[3] = {1, 2, 3}, mid[3] = {0, 0, 2}, fid[3] = {0, 0, 1}, sex[3] = {1, 2, 2};
  famseq_model m;
  char err[256] = {0};
  if (famseq_model_init(&m, 3, id, mid, fid, sex, NULL, 1e-7, 1.0) != 0) return 10;
  if (m.mother[2] != 1 || m.father[2] != 0 || m.mother[0] != -1) return 11;
  const int32_t bad_fid[3] = {0, 0, 0};
  if (famseq_model_init(&m, 3, id, mid, bad_fid, sex, NULL, 1e-7, 1.0) != FAMSEQ_E_PED_HALF) return 12;
  famseq_model_init(&m, 3, id, mid, fid, sex, NULL, 1e-7, 1.0);

  double lk[2][3][3] = {{{1, 1e-3, 1e-9}, {1e-2, 1, 1e-4}, {1e-3, 1, 1e-2}}, {{1, 1e-20, 1e-30}, {1, 1e-20, 1e-30}, {1, 1e-20, 1e-30}}};
  uint8_t flags[2] = {0, FAMSEQ_FLAG_KNOWN}, status[2] = {9, 9};
  double post[2][3][3], single[2][3][3];

  famseq_ctx *plan_only = famseq_create(&m, -1, err, sizeof err);
  if (!plan_only) return 13;
  if (!strstr(famseq_plan_json(plan_only), "\"N\":3")) return 14;
  if (famseq_bn_batch(plan_only, 2, &lk[0][0][0], flags, &post[0][0][0], &single[0][0][0], status) != FAMSEQ_E_NODEVICE) return 15;
  if (!strstr(famseq_last_error(plan_only), "no CPU path")) return 16;
  famseq_destroy(plan_only);

  if (famseq_device_count() < 1) {
    famseq_ctx *c = famseq_create(&m, 0, err, sizeof err);
    if (c != NULL || err[0] == 0) return 17; /* must fail loudly, never fall back */
    printf("no device: %s\n", err);
    return 0;
  }
  famseq_ctx *c = famseq_create(&m, 0, err, sizeof err);
  if (!c) {
    fprintf(stderr, "%s\n", err);
    return 18;
  }
  if (famseq_bn_batch(c, 2, &lk[0][0][0], flags, &post[0][0][0], &single[0][0][0], status) != 0) return 19;
  for (int s = 0; s < 2; s++)
    for (int i = 0; i < 3; i++) {
      const double sum = post[s][i][0] + post[s][i][1] + post[s][i][2];
      if (!(fabs(sum - 1.0) < 1e-12)) return 20;
    }
  if (status[0] != FAMSEQ_ST_OK || status[1] != (FAMSEQ_ST_OK | FAMSEQ_ST_SHORTCUT)) return 21;
  int8_t gt[3];
  famseq_call_genotypes(&post[0][0][0], 3, gt);
  printf("gpu: child posterior %.6g %.6g %.6g, calls %d %d %d\n", post[0][2][0], post[0][2][1], post[0][2][2], gt[0], gt[1], gt[2]);
  if (gt[0] != 0 || gt[1] != 1) return 22;
  famseq_destroy(c);

  /* the size-independent model: a 22-member pedigree (two founders, twenty children) through famseq_pedigree —
   * beyond famseq_model's arrays, served by the sum-product engine only */
  {
    enum { NW = 22 };
    int32_t wid[NW], wmid[NW], wfid[NW], wsex[NW], wmo[NW], wfa[NW];
    for (int i = 0; i < NW; i++) wid[i] = i + 1, wmid[i] = i < 2 ? 0 : 2, wfid[i] = i < 2 ? 0 : 1, wsex[i] = i == 0 ? 1 : (i == 1 ? 2 : 1 + i % 2);
    famseq_pedigree wp;
    if (famseq_pedigree_init(&wp, NW, wid, wmid, wfid, wsex, NULL, 1e-7, 1.0, wmo, wfa) != 0) return 30;
    if (wp.mother[5] != 1 || wp.father[5] != 0 || wp.mother[0] != -1) return 31;
    famseq_ctx *w = famseq_create_pedigree(&wp, 0, err, sizeof err);
    if (!w) {
      fprintf(stderr, "%s\n", err);
      return 32;
    }
    if (famseq_set_option(w, "engine", FAMSEQ_ENGINE_ENUM) != FAMSEQ_E_ARG) return 33; /* 3^22 is no enumeration's business */
    double wlk[2][NW][3], wpost[2][NW][3], wsingle[2][NW][3];
    uint8_t wst[2] = {9, 9};
    for (int s2 = 0; s2 < 2; s2++)
      for (int i = 0; i < NW; i++) wlk[s2][i][0] = 1.0, wlk[s2][i][1] = 1e-2 * (1 + (i + s2) % 3), wlk[s2][i][2] = 1e-5;
    if (famseq_bn_batch(w, 2, &wlk[0][0][0], NULL, &wpost[0][0][0], &wsingle[0][0][0], wst) != 0) return 34;
    for (int s2 = 0; s2 < 2; s2++)
      for (int i = 0; i < NW; i++)
        if (!(fabs(wpost[s2][i][0] + wpost[s2][i][1] + wpost[s2][i][2] - 1.0) < 1e-12)) return 35;
    if (wst[0] != FAMSEQ_ST_OK || wst[1] != FAMSEQ_ST_OK) return 36;
    printf("gpu: 22-member pedigree through famseq_create_pedigree, child 3 posterior %.6g %.6g %.6g\n", wpost[0][2][0], wpost[0][2][1], wpost[0][2][2]);
    famseq_destroy(w);
  }
  return 0;
}
