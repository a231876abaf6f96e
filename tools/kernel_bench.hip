// kernel_bench.hip — times one generated kernel (a .hsaco from the JIT cache, or a hand-edited
// variant compiled with `hipcc --genco`) on synthetic likelihood rows, outside the library.
// A tuning aid: lets several variants of a generated source be compared in one GPU session.
//   kernel_bench FILE.hsaco ENTRY N_MEMBERS BLOCK_THREADS BLOCKS_PER_CU [N_SITES=4000000] [LC=1.0]
// The factor table is filled with a constant (timing only: the arithmetic does not depend on the
// values), the likelihoods with a fixed pseudo-random pattern in (0,1].
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void fill(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z ^= z >> 31;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29;
    p[i] = ((z >> 11) + 1) * (1.0 / 9007199254740992.0);
  }
}

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s FILE.hsaco ENTRY N_MEMBERS BLOCK_THREADS BLOCKS_PER_CU [N_SITES] [LC]\n", argv[0]);
    return 2;
  }
  const char *file = argv[1], *entry = argv[2];
  const int n = atoi(argv[3]), bt = atoi(argv[4]), bpc = atoi(argv[5]);
  const long n_sites = argc > 6 ? atol(argv[6]) : 4000000;
  double lc = argc > 7 ? atof(argv[7]) : 1.0;
  hipModule_t mod;
  hipFunction_t fn;
  CHECK(hipModuleLoad(&mod, file));
  CHECK(hipModuleGetFunction(&fn, mod, entry));
  int occ = 0;
  CHECK(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, bt, 0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const size_t w = (size_t)n_sites * 3 * n;
  double *lk, *post, *single, *tc;
  unsigned char *status;
  CHECK(hipMalloc(&lk, w * 8));
  CHECK(hipMalloc(&post, w * 8));
  CHECK(hipMalloc(&single, w * 8));
  CHECK(hipMalloc(&status, n_sites));
  CHECK(hipMalloc(&tc, 432 * 8));
  fill<<<2048, 256>>>(lk, w);
  std::vector<double> h(432, 0.25);
  CHECK(hipMemcpy(tc, h.data(), 432 * 8, hipMemcpyHostToDevice));
  const unsigned char *flags = nullptr;
  long ns = n_sites;
  const long chunks = (n_sites + bt - 1) / bt;
  const int per_cu = bpc > 0 ? bpc : occ;
  const unsigned grid = (unsigned)(chunks < (long)prop.multiProcessorCount * per_cu ? chunks : (long)prop.multiProcessorCount * per_cu);
  void *args[] = {&lk, &flags, &post, &single, &status, &ns, &tc, &lc};
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CHECK(hipModuleLaunchKernel(fn, grid, 1, 1, bt, 1, 1, 0, 0, args, nullptr));
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) CHECK(hipModuleLaunchKernel(fn, grid, 1, 1, bt, 1, 1, 0, 0, args, nullptr));
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  std::vector<unsigned char> st(16);
  CHECK(hipMemcpy(st.data(), status, 16, hipMemcpyDeviceToHost));
  const double bytes = (double)n_sites * (72.0 * n + 2);
  printf("%-44s occ %d/CU grid %u  %8.4f ms  %6.2f Gsites/s  %.3f of 8 TB/s  status[0]=%d\n", file, occ, grid, ms,
         n_sites / ms * 1e-6, bytes / ms * 1e-6 / 8000, st[0]);
  return 0;
}
