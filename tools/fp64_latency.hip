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

int main() {
  for (int w : {1, 2}) {
    run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w); run<8>(w);
  }
  return 0;
}
