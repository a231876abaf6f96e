// fma_issue.hip — how fast can ONE wave (and two) issue independent v_fma_f64 on a gfx950 SIMD, by instruction form?
// The enumeration kernel's inner block is 27 x `S[c] = fma(p, W[c], S[c])` per prefix p at two waves per SIMD, and a pure
// stream of such FMAs reaches only 0.83 of the nominal rate at two waves (0.67 at one).  Is that the form?
//   fmac     v_fmac_f64 S, p, W         (VOP2, 4 bytes; what hipcc emits)
//   fma3     v_fma_f64 S, p, W, S       (VOP3, 8 bytes)
//   nop      fmac with an s_nop 0 after each (is there an issue slot to spare?)
//   distinct every FMA its own multiplicand pair (no shared p)
//   pk32     v_pk_fma_f32 beside it for scale (same lanes x 2)
// Chip-wide rates from HIP events, 27 accumulators per wave, 1 / 2 / 3 / 4 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

#define K 27
enum Form { FMAC = 0, FMA3 = 1, NOP = 2, DISTINCT = 3 };

template <int FORM>
__global__ __launch_bounds__(256) void stream(double *out, double a, double b, int iters) {
  double S[K], W[K];
#pragma unroll
  for (int k = 0; k < K; ++k) S[k] = threadIdx.x + k, W[k] = b + k * 1e-3;
  double p = a + threadIdx.x * 1e-9;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (FORM == FMAC) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(S[k]) : "v"(p), "v"(W[k]));
        else if (FORM == FMA3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(S[k]) : "v"(p), "v"(W[k]));
        else if (FORM == NOP) asm volatile("v_fmac_f64 %0, %1, %2\n\ts_nop 0" : "+v"(S[k]) : "v"(p), "v"(W[k]));
        else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(S[k]) : "v"(W[(k + 1) % K]), "v"(W[k]));
      }
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(p) : "v"(W[r]));  // a new prefix per 27, as in the kernel
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += S[k];
  out[threadIdx.x + (size_t)blockIdx.x * blockDim.x] = s + p;
}

static const char *fname(int f) { return f == FMAC ? "fmac" : f == FMA3 ? "fma3" : f == NOP ? "fmac+s_nop" : "distinct"; }

template <int FORM>
static void rate(int waves_per_simd, double *out) {
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  stream<FORM><<<blocks, 256>>>(out, 0.999, 1e-3, 50);
  (void)hipEventRecord(e0);
  stream<FORM><<<blocks, 256>>>(out, 0.999, 1e-3, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double lanes = (double)blocks * 256 * iters * 4 * (K + 1);
  printf("%-11s %d wave(s)/SIMD: %6.2f T fp64 lane-ops/s (nominal 39.3)  %.1f ms\n", fname(FORM), waves_per_simd, lanes / ms * 1e-9, ms);
  fflush(stdout);
}

int main() {
  double *out;
  (void)hipMalloc(&out, 8 * 256 * 4096);
  for (int w : {1, 2, 3, 4}) {
    rate<FMAC>(w, out);
    rate<FMA3>(w, out);
    rate<NOP>(w, out);
    rate<DISTINCT>(w, out);
  }
  (void)hipFree(out);
  return 0;
}
