// fp64_latency.hip — issue/latency of v_fma_f64 on one wave: K independent dependency chains,
// cycles per FMA from s_memtime.  (How many independent accumulators does a wave need?)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void chains(double *out, long long *cyc, double a, double b) {
  double x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = threadIdx.x + k;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K>
void run(int waves_per_simd) {
  double *out;
  long long *cyc;
  hipMalloc(&out, 8 * 64 * 4096);
  hipMalloc(&cyc, 8 * 4096);
  const int blocks = 256 * 4 * waves_per_simd;  // 64-lane blocks: one wave each
  chains<K><<<blocks, 64>>>(out, cyc, 0.999, 1e-3);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("K=%d chains, %d wave(s)/SIMD: %.2f clock ticks per FMA per wave (s_memtime ticks at 100 MHz: x24 for 2.4 GHz cycles)\n", K,
         waves_per_simd, (double)h[0] / (256.0 * 16 * K));
  hipFree(out);
  hipFree(cyc);
}

// chip-wide sustained rate: every SIMD busy with `waves` waves of K-chain FMAs, wall time from HIP events
template <int K>
__global__ void sustained(double *out, double a, double b, int iters) {
  double x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = threadIdx.x + k;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  out[threadIdx.x + (size_t)blockIdx.x * blockDim.x] = s;
}

static void rate(int waves_per_simd) {
  double *out;
  (void)hipMalloc(&out, 8 * 256 * 4096);
  const int blocks = 256 * waves_per_simd, iters = 20000;  // 256-thread blocks: one wave per SIMD each
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  sustained<8><<<blocks, 256>>>(out, 0.999, 1e-3, 100);
  (void)hipEventRecord(a);
  sustained<8><<<blocks, 256>>>(out, 0.999, 1e-3, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double lanes = (double)blocks * 256 * iters * 16 * 8;
  printf("sustained, %d wave(s)/SIMD on all 256 CUs: %.2f T fp64 FMA lane-ops/s (nominal 39.3), %.1f ms\n", waves_per_simd,
         lanes / ms * 1e-9, ms);
  (void)hipFree(out);
}

int main() {
  rate(1);
  rate(2);
  rate(4);
  for (int w : {1, 2}) {
    run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w); run<8>(w);
  }
  return 0;
}
