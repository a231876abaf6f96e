// coexec.hip — do the fp64 vector pipe and the fp64 matrix pipe of a gfx950 SIMD run at the same time?
// Waves 0-3 of a 512-thread workgroup (one per SIMD) run role A, waves 4-7 (their SIMD partners) role B;
// roles: V = stream of independent v_fma_f64 (8 chains), M4 = v_mfma_f64_4x4x4_4b (8 accumulators),
// M16 = v_mfma_f64_16x16x4 (4 accumulators), idle = exits at once.  Wall time from HIP events, every CU busy;
// rates in 1e12 fp64 FMA lane-operations per second (nominal vector peak 39.3, matrix peak 39.3).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
enum Role { IDLE = 0, V = 1, M4 = 2, M16 = 3 };

__device__ __forceinline__ double role_v(int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  return s;
}

__device__ __forceinline__ double role_m4(int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x + k;
  a += threadIdx.x * 1e-9;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x[k], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  return s;
}

__device__ __forceinline__ double role_m16(int iters, double a, double b) {
  d4 x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (d4){1.0 * threadIdx.x, 2.0, 3.0, 4.0 + k};
  a += threadIdx.x * 1e-9;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x[k], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += x[k].x + x[k].y + x[k].z + x[k].w;
  return s;
}

// FMA lane-operations one wave performs per iteration of its role
static double ops_per_iter(int role) {
  switch (role) {
    case V: return 16.0 * 8 * 64;
    case M4: return 4.0 * 8 * 256;
    case M16: return 2.0 * 4 * 1024;
  }
  return 0;
}

__global__ __launch_bounds__(512) void mix(double *out, int role_a, int role_b, int iters_a, int iters_b, double a, double b) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? role_a : role_b, iters = wave < 4 ? iters_a : iters_b;
  double s = 0;
  if (role == V) s = role_v(iters, a, b);
  else if (role == M4) s = role_m4(iters, a, b);
  else if (role == M16) s = role_m16(iters, a, b);
  else return;
  out[threadIdx.x + (size_t)blockIdx.x * blockDim.x] = s;
}

static const char *name(int r) { return r == V ? "V" : r == M4 ? "M4x4x4" : r == M16 ? "M16x16x4" : "idle"; }

static void run(int role_a, int role_b, int wg_per_cu, double *out) {
  // iterations sized so that each role alone would take about the same time (both pipes nominally 16 FMA/clk/SIMD)
  const double target = 2.0e9;  // lane-ops per wave
  const int ia = role_a ? (int)(target / ops_per_iter(role_a)) : 0, ib = role_b ? (int)(target / ops_per_iter(role_b)) : 0;
  const int blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  mix<<<blocks, 512>>>(out, role_a, role_b, ia / 50 + 1, ib / 50 + 1, 0.999, 1e-3);
  (void)hipEventRecord(e0);
  mix<<<blocks, 512>>>(out, role_a, role_b, ia, ib, 0.999, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double oa = (double)blocks * 4 * ia * ops_per_iter(role_a), ob = (double)blocks * 4 * ib * ops_per_iter(role_b);
  printf("%-9s + %-9s  %d workgroup(s)/CU: %7.2f ms   A %.2f T/s  B %.2f T/s  together %.2f T FMA lane-ops/s\n", name(role_a),
         name(role_b), wg_per_cu, ms, oa / ms * 1e-9, ob / ms * 1e-9, (oa + ob) / ms * 1e-9);
  fflush(stdout);
}

int main() {
  double *out;
  (void)hipMalloc(&out, 8 * 512 * 2048);
  const int pairs[][2] = {{V, IDLE}, {V, V}, {M4, IDLE}, {M4, M4}, {M16, IDLE}, {M16, M16}, {V, M4}, {V, M16}};
  for (auto &p : pairs) run(p[0], p[1], 1, out);
  run(V, M4, 2, out);
  run(V, V, 2, out);
  (void)hipFree(out);
  return 0;
}
