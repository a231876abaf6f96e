/*
 * famseq_hip.h — C ABI of libfamseq_hip.so: the MI355X (gfx950) implementation of
 * FamSeq's `-method 1` Bayesian-network pedigree genotype posterior.
 *
 * Boundary being replaced (all cites into /root/reference/src):
 *   the per-site operator   bool family::calPostProbBN(bool Known,int chrType)  family.h:375,
 *                           family.cpp:750-1124 (CUDA twin family.cu:974-2305)
 *   fed by                  bool family::set_LK(const dMatrix<double>&)          family.h:344, family.cpp:698-748
 *   and drained by          get_postProb / get_postProbSingle / get_postRlt      family.h:262,265,308
 *   called from             file.cpp:595->607->680-682, :833->845->918-920, :1743->1751->1804-1806
 * The reference crosses host->device once per site (6 cudaMalloc + 5 H2D + 1 launch
 * + 1 D2H per site, family.cu:1152-1703).  This ABI re-cuts the same operator as a
 * BATCH of sites per call: plain pointers and sizes, no C++ or torch types.
 *
 * Data layout (both directions): [n_sites][n_members][3] fp64, row-major, members in
 * PED order, genotype order RR,RA,AA — i.e. dMatrix<double> N x 3 (dMatrix.h:141-151)
 * repeated per site.  flags[s]: bit0 = Known (ID != "."), bit1 = chrX (file.cpp:476-486).
 *
 * status[s] (replaces the bool return + flagPB/flagPBS, family.cpp:752-763, :946-949):
 *   0     ok, full enumeration
 *   0x80  ok, took the -LRC single-sample shortcut (family.cpp:767-878)
 *   1     calPostProbSingle failed: a lk*prior row sum <= 0 (family.cpp:1437); post and
 *         post_single are NaN-filled — the caller prints `:NA:NA:NA` (file.cpp:607-620)
 *   2     BN row sum <= 0 (family.cpp:946/:1111); post is NaN-filled, post_single valid
 *
 * There is NO CPU fallback: every compute entry point fails (negative return, message
 * in famseq_last_error) when no gfx950 device is usable.
 */
#ifndef FAMSEQ_HIP_H_
#define FAMSEQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Limit of the 3^N enumeration engines (and the size of famseq_model's arrays): the reference CUDA file's own,
 * geno[20], pTmp[60] (family.cu:771-772) — 3^20 = 3.5e9 configurations per site is past any use of an enumeration
 * anyway.  The sum-product engine (FAMSEQ_ENGINE_ELIM) has no member limit, like the reference's CPU code
 * (std::vector state, family.cpp:894-941; its -method 2 peeling, :1126-1403, :1501-1845): see famseq_pedigree. */
#define FAMSEQ_MAX_MEMBERS 20

#define FAMSEQ_ST_OK 0
#define FAMSEQ_ST_SINGLE_FAIL 1
#define FAMSEQ_ST_BN_FAIL 2
#define FAMSEQ_ST_SHORTCUT 0x80

#define FAMSEQ_FLAG_KNOWN 1
#define FAMSEQ_FLAG_CHRX 2

/* engines (famseq_set_option "engine") */
#define FAMSEQ_ENGINE_ENUM 0 /* 3^N joint-genotype enumeration: any pedigree (the reference's -method 1 algorithm) */
#define FAMSEQ_ENGINE_ELIM 1 /* exact sum-product on the pedigree's factor graph: same marginals in O(27 N) per
                                site (pedigrees with loops: conditioned on up to three members, x3 per member);
                                kernel generated and compiled per pedigree */

/* error codes (negative returns) */
#define FAMSEQ_E_ARG (-1)      /* bad argument / model rejected */
#define FAMSEQ_E_NODEVICE (-2) /* no usable gfx950 device, or ctx created without one */
#define FAMSEQ_E_HIP (-3)      /* HIP runtime error, see famseq_last_error */
#define FAMSEQ_E_PED_HALF (-10) /* member with exactly one known parent (family.cpp:319-323) */
#define FAMSEQ_E_PED_SEX (-11)  /* mother not female / father not male (family.cpp:204-219) */

/* Everything `family` holds after setFam()+init() that the path reads
 * (file.cpp:1888-1927; family.cpp:78-127, 221-238). */
typedef struct {
  int32_t n_members;                     /* numInd, 1..FAMSEQ_MAX_MEMBERS */
  int32_t mother[FAMSEQ_MAX_MEMBERS];    /* parent[i][0] or -1            family.cpp:329 */
  int32_t father[FAMSEQ_MAX_MEMBERS];    /* parent[i][1] or -1            family.cpp:330 */
  int32_t gender[FAMSEQ_MAX_MEMBERS];    /* 1 = male, anything else is treated as female (family.cpp:1054) */
  uint8_t sequenced[FAMSEQ_MAX_MEMBERS]; /* 1 if the member is in mapV2P (has a VCF/LK column); the
                                            shortcut vote runs over these only (family.cpp:768-771) */
  double pcp2[27];   /* autosome   [child*9 + mother*3 + father]  family.cpp:447-550 */
  double pcp2Xf[27]; /* chrX daughter                             family.cpp:383-416 */
  double pcp2Xm[27]; /* chrX son                                  family.cpp:418-445 */
  double genoProbN[3], genoProbK[3], genoProbXN[3], genoProbXK[3]; /* family.cpp:91-109 */
  double lc;         /* m_lc, -LRC                                family.cpp:253-257 */
} famseq_model;

/* The same for a pedigree of ANY size: the per-member arrays are the caller's ([n_members] each, copied by
 * famseq_create_pedigree; `sequenced` may be NULL = all sequenced).  Contexts of more than FAMSEQ_MAX_MEMBERS
 * members serve FAMSEQ_ENGINE_ELIM only — what the reference's -method 2 is recommended for ("family size > 7 and
 * loop-free", FamSeq_Manual.pdf p.1) — and answer FAMSEQ_E_ARG to a batch call while the engine is the enumeration. */
typedef struct {
  int32_t n_members;        /* numInd, >= 1 */
  const int32_t *mother;    /* [n_members] parent[i][0] or -1    family.cpp:329 */
  const int32_t *father;    /* [n_members] parent[i][1] or -1    family.cpp:330 */
  const int32_t *gender;    /* [n_members] 1 = male              family.cpp:1054 */
  const uint8_t *sequenced; /* [n_members] or NULL               family.cpp:768-771 */
  double pcp2[27], pcp2Xf[27], pcp2Xm[27];
  double genoProbN[3], genoProbK[3], genoProbXN[3], genoProbXK[3];
  double lc;
} famseq_pedigree;

typedef struct famseq_ctx famseq_ctx;

/* ---- one-time model setup (replaces family ctor + init()) ---------------- */

/* calPCP2S / calPCP2Xf / calPCP2Xm for mutation rate `mrate`, bit-faithful to the
 * reference's accumulation order.  Each output is 27 doubles. */
void famseq_transmission_tables(double mrate, double *pcp2, double *pcp2Xf, double *pcp2Xm);

/* PED columns -> model: default priors, tables for `mrate`, setRelation, checkPed.
 * `sequenced` may be NULL (all sequenced).  Returns 0 or FAMSEQ_E_*. */
int famseq_model_init(famseq_model *m, int32_t n_members, const int32_t *id, const int32_t *mother_id,
                      const int32_t *father_id, const int32_t *gender, const uint8_t *sequenced,
                      double mrate, double lc);

/* The same for famseq_pedigree: fills the priors, the tables and lc of *p, and the caller's mother_idx /
 * father_idx arrays ([n_members] each) with setRelation's indices, and points p->mother / p->father at them;
 * p->gender and p->sequenced point at the arguments (which must outlive p's use).  Returns 0 or FAMSEQ_E_*. */
int famseq_pedigree_init(famseq_pedigree *p, int32_t n_members, const int32_t *id, const int32_t *mother_id,
                         const int32_t *father_id, const int32_t *gender, const uint8_t *sequenced, double mrate,
                         double lc, int32_t *mother_idx, int32_t *father_idx);

/* ---- context -------------------------------------------------------------- */

int famseq_device_count(void); /* visible HIP devices; 0 when there is none */

/* Validates the model, builds the enumeration plan and (device_id >= 0) binds one GPU:
 * uploads constants, creates streams and staging buffers.  One ctx = one device = one
 * caller thread (multi-GPU = one process/ctx per device, sites sharded by the caller).
 * device_id < 0 builds a plan-only ctx (for inspection; compute calls return
 * FAMSEQ_E_NODEVICE).  On failure returns NULL with a message in err. */
famseq_ctx *famseq_create(const famseq_model *model, int device_id, char *err, size_t errlen);
/* ... from a size-independent model.  Up to FAMSEQ_MAX_MEMBERS members: exactly famseq_create.  Beyond: no
 * enumeration plan is built, the context starts on FAMSEQ_ENGINE_ELIM (generated and compiled here; fails with a
 * message when the pedigree's loops need more than three conditioned members). */
famseq_ctx *famseq_create_pedigree(const famseq_pedigree *pedigree, int device_id, char *err, size_t errlen);
void famseq_destroy(famseq_ctx *ctx);
const char *famseq_last_error(famseq_ctx *ctx);

/* Integer knobs; must be set before the first batch call.  Keys:
 *   "fixed_digits"  A: high members mapped onto lanes (team = 3^A lanes), 0..6
 *   "low_members"   L: childless members enumerated in the unrolled register loop, 1..5
 *   "block_threads" workgroup size (multiple of 64, <= 768)
 *   "grid_blocks"   persistent grid size (0 = auto: CUs x resident blocks)
 *   "chunk_sites"   host-staging chunk of famseq_bn_batch (0 = auto)
 *   "enum_impl"     which enumeration kernel serves FAMSEQ_ENGINE_ENUM: 0 = the team-per-site kernel
 *                   compiled into the library (any batch size, any pedigree, no compiler needed); 1 = the
 *                   kernel generated and compiled for this pedigree, one lane per site unless "group_digits"
 *                   says otherwise; -1 (default) = the generated kernel, lanes per site chosen by batch size,
 *                   for batches of >= "lane_min_sites" (256) sites when it can be built, the team kernel
 *                   otherwise (tiny calls never wait for a compile)
 *   "tune"          1 = time the candidates of this pedigree's generated kernels on this device (where a static rule
 *                   picks a variant: the enumeration kernel's 7- or 6-member unrolled block, the sum-product kernel's
 *                   fence variant and where it keeps the likelihoods) on synthetic rows, a few milliseconds each, and keep the winners' indices as notes
 *                   in the kernel cache — every later context for the pedigree starts from them; compiles every
 *                   candidate (seconds each), needs a device; famseq_plan_json "tune" reports what was measured
 *   "pick_lane", "pick_elim"  a variant index measured elsewhere, kept as the note "tune" would leave (how build() ships
 *                   its table of measured picks); works on a plan-only context
 *   "prebuild_lane", "prebuild_elim"  compile that variant of the pedigree's kernel into the cache without loading it
 *                   (a plan-only context will do): how a build host prepares every candidate "tune" races
 *   "group_digits"  the generated kernel's lanes per site, 3^d: d = 0 one lane per site (large batches),
 *                   d = 1..4 lanes-per-site mode for batches too small to give every lane of the chip a
 *                   site (each lane of a group enumerates one combination of the d outermost looped
 *                   members' genotypes; the group sums through LDS); -1 (default) = chosen per call from
 *                   the batch size.  At most the number of looped members of the pedigree's enumeration
 *                   (famseq_plan_json "enum_group_digits_max"; 0 for pedigrees of up to 6 members)
 *   "phase_clock_report"  measuring aid: with FAMSEQ_PHASE_CLOCK=1 in the environment the generated kernels carry cycle marks between
 *                   their phases; this prints the plain kernels' shares (wave cycles per phase) on stderr and clears them
 *   "call_kernels"  1 = build the generated kernels' fused call-path forms now (famseq_bn_call_batch would on
 *                   its first call)
 *   "engine"        FAMSEQ_ENGINE_ENUM (default) or FAMSEQ_ENGINE_ELIM; selecting ELIM generates the
 *                   kernel for this pedigree, compiles it (libhiprtc in-process; cached on disk) and fails with
 *                   FAMSEQ_E_ARG on a pedigree whose loops need more than three conditioned members
 * Returns 0 or FAMSEQ_E_ARG. */
int famseq_set_option(famseq_ctx *ctx, const char *key, int64_t value);

/* JSON description of the plan and launch geometry (valid until the next call on ctx). */
const char *famseq_plan_json(famseq_ctx *ctx);

/* ---- the operator ---------------------------------------------------------- */

/* Host buffers (pageable, or pinned for the full rate of the link).  Blocking.  Works in chunks
 * through three event-chained stages (copy in / compute / copy out, one HIP stream each) so that both
 * directions of the host link are busy at once.  post_single and status may be NULL.  Returns 0 or a
 * negative error.  (The host link bounds this entry point: 722 B/site at N = 10; see
 * famseq_bn_call_batch for the compact path.) */
int famseq_bn_batch(famseq_ctx *ctx, int64_t n_sites, const double *lk, const uint8_t *flags,
                    double *post, double *post_single, uint8_t *status);

/* The same over several GPUs of one node from one process: ctxs[g] (one per device, created by the
 * caller for the same model) gets the contiguous site range [g*S/G, (g+1)*S/G) and its own host
 * thread; results land in the caller's arrays in site order.  No collective: sites are independent
 * (family.cpp:791 zeroes all state per site).  Returns 0, or the first error any
 * ctx reported (famseq_last_error on that ctx has the text). */
int famseq_bn_batch_sharded(famseq_ctx *const *ctxs, int n_ctx, int64_t n_sites, const double *lk, const uint8_t *flags,
                            double *post, double *post_single, uint8_t *status);

/* Device buffers already resident in HBM on ctx's device.  Enqueues on `stream`
 * (a hipStream_t; NULL = the default stream) and returns without synchronising.
 * d_post_single and d_status may be NULL. */
int famseq_bn_batch_device(famseq_ctx *ctx, int64_t n_sites, const double *d_lk, const uint8_t *d_flags,
                           double *d_post, double *d_post_single, uint8_t *d_status, void *stream);

/* Device-resident form of famseq_bn_batch_sharded, for a process that holds several GPUs (the
 * reference has no multi-GPU: family.cu:1152 binds device 0): shard g is n_sites[g] sites whose
 * arrays are already resident on ctxs[g]'s device.  One host thread per ctx enqueues the shard on
 * that ctx's own compute stream and waits for it; blocking; no collective and no copy between
 * devices.  d_flags / d_post_single / d_status may be NULL, and so may their entries.  Returns 0 or
 * the first error any ctx reported.  (Several ctxs may name the same device: rehearsal on one GPU.) */
int famseq_bn_batch_device_sharded(famseq_ctx *const *ctxs, int n_ctx, const int64_t *n_sites,
                                   const double *const *d_lk, const uint8_t *const *d_flags, double *const *d_post,
                                   double *const *d_post_single, uint8_t *const *d_status);

/* Fused call path (SURVEY.md 8(f) rows N2 + N4): what the drivers print per sequenced sample, computed on
 * the device, so that only 49*n_seq bytes per site come back instead of 48*N.
 *   input   either lk  [n_sites][N][3] fp64 (as famseq_bn_batch)
 *           or     pl16 [n_sites][n_seq][3] uint16: integer PL/GL magnitudes in VCF column order, turned
 *                  into likelihoods on the device with the reference's pow(10,-|x|/10) (file.cpp:588-590,
 *                  table filled by the host libm); 0xFFFF,0xFFFF,0xFFFF = sample missing at this site
 *                  (likelihood {1,1,1}, file.cpp:794-809); members outside seq_members are {1,1,1} (:565)
 *   seq_members[n_seq]  PED index of each sequenced sample in VCF column order (mapV2P[i] >= 0)
 *   gpp, fpp [n_sites][n_seq][3]  fabs(-10*log10(p)) of the single / BN posterior, +inf -> 99999
 *                                 (file.cpp:696-745); NaN where the site failed
 *   fgt      [n_sites][n_seq]     arg-max genotype 0/1/2 (family.cpp:636-665), -1 where the site failed
 * Any of gpp / fpp / fgt / status may be NULL.  Blocking; chunks are pipelined like famseq_bn_batch.
 * Batches served by a generated kernel run ONE kernel per chunk — its call-path form unpacks the PLs into
 * the LDS rows and turns the posterior rows into GPP / FPP / FGT while storing them; the fp64 likelihoods and
 * posteriors never exist in HBM.  The compiled-in team kernel and the lanes-per-site mode (tiny batches)
 * keep the separate unpack / posterior / Phred stages; both ways give the same bits. */
int famseq_bn_call_batch(famseq_ctx *ctx, int64_t n_sites, const double *lk, const uint16_t *pl16,
                         const uint8_t *flags, const int32_t *seq_members, int32_t n_seq, double *gpp, double *fpp,
                         int8_t *fgt, uint8_t *status);

/* The same call with the outputs as TEXT (SURVEY.md 8(f) rows N1 + N2): what the reference's drivers append to every
 * sample column of an output line (file.cpp:696-745) — the GPP triple, the FPP triple and the called genotype,
 * "g0,g1,g2:f0,f1,f2:0/1\t", every number as C++ `ostream << double` prints it (printf's %g: six significant digits of the
 * exact binary value, round-half-even) — formatted on the device by one more streaming kernel over the called outputs
 * while they are in HBM (csrc/io_kernels.hip text_call_kernel; csrc/g6_core.h is the digit code it shares with the host
 * formatter).  The sixty %g conversions per ten-member site were the slowest stage of the command line (0.74 of its 0.88 s
 * loop per 3 M sites on 16 host threads); with this entry the host copies one record per sample.
 *   text [n_sites][n_seq][FAMSEQ_TEXT_STRIDE]  one record per (site, sequenced sample) in VCF column order: the characters
 *        from byte 0, their count (<= 76) in byte FAMSEQ_TEXT_STRIDE - 1, zeros between.  Records of a site whose
 *        status has bit 0 or 1 set hold no number to print (the drivers write ":NA:NA:NA" there, file.cpp:607-620).
 * Other arguments, chunking and kernels as famseq_bn_call_batch; 80 n_seq + 1 bytes per site come back. */
#define FAMSEQ_TEXT_STRIDE 80
int famseq_bn_call_text_batch(famseq_ctx *ctx, int64_t n_sites, const double *lk, const uint16_t *pl16,
                              const uint8_t *flags, const int32_t *seq_members, int32_t n_seq, char *text, uint8_t *status);

/* The call path on DEVICE buffers already resident on ctx's device: enqueues on `stream` (a hipStream_t; NULL = the default stream)
 * and returns without synchronising; nothing crosses the host link.  Exactly one of d_lk / d_pl16; seq_members is a HOST array
 * (uploaded when it changes); any of d_gpp / d_fpp / d_fgt / d_status / d_text may be NULL (d_text: 16-byte aligned,
 * [n_sites][n_seq][FAMSEQ_TEXT_STRIDE]).  The same kernels as famseq_bn_call_batch / famseq_bn_call_text_batch; batches they do not
 * serve fused go through scratch rows this context keeps (grown on demand). */
int famseq_bn_call_batch_device(famseq_ctx *ctx, int64_t n_sites, const double *d_lk, const uint16_t *d_pl16,
                                const uint8_t *d_flags, const int32_t *seq_members, int32_t n_seq, double *d_gpp, double *d_fpp,
                                int8_t *d_fgt, uint8_t *d_status, char *d_text, void *stream);

/* Diagnostic / test aid: the device formatter alone.  values[n] (host) -> out[n][16] (host): the characters of each
 * value as the text kernel prints a GPP / FPP number from byte 0, their count in byte 15; "nan" for anything outside
 * the formatter's domain, 0 and [1e-16, 999999.5). */
int famseq_format_probe(famseq_ctx *ctx, int64_t n, const double *values, char *out);

/* Diagnostic for measurement (bench.py): runs the posterior kernels' traffic shape — read one fp64 array of
 * n_doubles, write two — as a bare elementwise kernel on `stream` and returns without synchronising.  What a
 * device's memory system sustains for that shape differs between MI355X devices by 10-20 %; timing this next to
 * the kernels says how much of what is attainable THERE they reach.  Arrays must be 16-byte aligned. */
int famseq_stream_probe(famseq_ctx *ctx, int64_t n_doubles, const double *d_in, double *d_out1, double *d_out2, void *stream);

/* Page-locked host memory for the buffers handed to famseq_bn_batch / famseq_bn_call_batch: with pinned
 * buffers both directions of the host link run at their full rate at once (the callers of the reference's
 * operator own their buffers, file.cpp:565; this is how to own fast ones without linking HIP).  Returns NULL
 * when there is no device or no memory; famseq_free_pinned(NULL) is a no-op. */
void *famseq_alloc_pinned(size_t bytes);
void famseq_free_pinned(void *p);

/* get_postRlt (family.cpp:636-665) for one N x 3 posterior row block: arg-max with strict '<' starting from -1, so ties
 * resolve to the lowest genotype — here with posteriors within 1e-12 (relative) of the largest counting as ties: the
 * reference's exact ties (a 0.5 / 0.5 child at mutation rate 0) are ties to rounding only in another order of summation,
 * and the call should be the reference's there too.  The device kernels (fgt above, the text records) use the same rule. */
void famseq_call_genotypes(const double *post, int64_t n_rows, int8_t *geno);

#ifdef __cplusplus
}
#endif
#endif /* FAMSEQ_HIP_H_ */
