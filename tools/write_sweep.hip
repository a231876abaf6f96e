// write_sweep.hip — what does an MI355X give a WRITE stream, by store shape?  The posterior kernels write two bytes for
// every byte they read, and a write-only stream of 16-byte non-temporal stores reached only 0.51 of the 8 TB/s peak
// (tools/io_ceiling.hip) against 0.78 for reads.  This sweeps the shapes a kernel can choose between:
//   width    4 / 8 / 16 bytes per lane and store
//   policy   plain, nt (non-temporal), sc1, sc0 sc1 (written through inline asm)
//   order    grid-stride (every wave-instruction of the grid lands next to its neighbours'), or one contiguous
//            range per workgroup, or one contiguous range per wave
//   grid     workgroups per CU
//   arrays   one output array or two
// usage: write_sweep [GB total = 4.8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
enum Policy { PLAIN = 0, NT = 1, SC1 = 2, SC01 = 3 };
enum Order { STRIDE = 0, PER_WG = 1, PER_WAVE = 2 };

template <int POLICY>
__device__ __forceinline__ void st16(f4 *p, f4 v) {
  if (POLICY == PLAIN) *p = v;
  else if (POLICY == NT) __builtin_nontemporal_store(v, p);
  else if (POLICY == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ void st8(f2 *p, f2 v) {
  if (POLICY == PLAIN) *p = v;
  else if (POLICY == NT) __builtin_nontemporal_store(v, p);
  else if (POLICY == SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ __forceinline__ void st4(float *p, float v) {
  if (POLICY == PLAIN) *p = v;
  else if (POLICY == NT) __builtin_nontemporal_store(v, p);
  else if (POLICY == SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// n = number of WIDTH-byte elements per array; every element written exactly once
template <int WIDTH, int POLICY, int ORDER, int ARRAYS>
__global__ __launch_bounds__(256) void wr(char *o1, char *o2, size_t n) {
  size_t lo, hi, step;
  if (ORDER == STRIDE) {
    lo = blockIdx.x * (size_t)256 + threadIdx.x, hi = n, step = (size_t)gridDim.x * 256;
  } else if (ORDER == PER_WG) {
    const size_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n, lo += threadIdx.x, step = 256;
  } else {
    const size_t waves = (size_t)gridDim.x * 4, w = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
    const size_t per = ((n + waves - 1) / waves + 63) / 64 * 64;
    lo = w * per, hi = lo + per < n ? lo + per : n, lo += threadIdx.x & 63, step = 64;
  }
  for (size_t i = lo; i < hi; i += step) {
    if (WIDTH == 16) {
      st16<POLICY>((f4 *)o1 + i, f4{1, 2, 3, 4});
      if (ARRAYS == 2) st16<POLICY>((f4 *)o2 + i, f4{5, 6, 7, 8});
    } else if (WIDTH == 8) {
      st8<POLICY>((f2 *)o1 + i, f2{1, 2});
      if (ARRAYS == 2) st8<POLICY>((f2 *)o2 + i, f2{5, 6});
    } else {
      st4<POLICY>((float *)o1 + i, 1.f);
      if (ARRAYS == 2) st4<POLICY>((float *)o2 + i, 5.f);
    }
  }
}

// the kernels' real shape: read one array, write two (16 bytes per lane), by policy and order
template <int POLICY, int ORDER>
__global__ __launch_bounds__(256) void rw(const f4 *__restrict__ in, f4 *o1, f4 *o2, size_t n) {
  size_t lo, hi, step;
  if (ORDER == STRIDE) {
    lo = blockIdx.x * (size_t)256 + threadIdx.x, hi = n, step = (size_t)gridDim.x * 256;
  } else {
    const size_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n, lo += threadIdx.x, step = 256;
  }
  for (size_t i = lo; i < hi; i += step) {
    const f4 v = in[i];
    st16<POLICY>(o1 + i, v * 2.f);
    st16<POLICY>(o2 + i, v + 1.f);
  }
}

// the same with U loads issued before the 2U stores of a tile (more reads in flight per lane), tiles dealt grid-stride
template <int POLICY, int U, int BLOCK, bool LOAD_NT>
__global__ __launch_bounds__(BLOCK) void rwu(const f4 *__restrict__ in, f4 *o1, f4 *o2, size_t n) {
  const size_t tile = (size_t)BLOCK * U, tiles = n / tile;
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const size_t base = t * tile + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = LOAD_NT ? __builtin_nontemporal_load(in + base + (size_t)u * BLOCK) : in[base + (size_t)u * BLOCK];
#pragma unroll
    for (int u = 0; u < U; ++u) st16<POLICY>(o1 + base + (size_t)u * BLOCK, v[u] * 2.f);
#pragma unroll
    for (int u = 0; u < U; ++u) st16<POLICY>(o2 + base + (size_t)u * BLOCK, v[u] + 1.f);
  }
}

static char *g_o1, *g_o2, *g_in;
static size_t g_bytes_per_array;
static hipEvent_t g_e0, g_e1;

template <typename F>
static float best_ms(F launch) {
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(g_e0));
    launch();
    CHECK(hipEventRecord(g_e1));
    CHECK(hipEventSynchronize(g_e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, g_e0, g_e1));
    if (rep && ms < best) best = ms;
  }
  return best;
}

static const char *pname(int p) { return p == PLAIN ? "plain" : p == NT ? "nt" : p == SC1 ? "sc1" : "sc0sc1"; }
static const char *oname(int o) { return o == STRIDE ? "grid-stride" : o == PER_WG ? "range/workgroup" : "range/wave"; }

template <int WIDTH, int POLICY, int ORDER, int ARRAYS>
static void one(int wg_per_cu) {
  const size_t total = g_bytes_per_array * 2;              // same bytes whether into one array or two
  const size_t n = total / ARRAYS / WIDTH;
  const int blocks = 256 * wg_per_cu;
  const float ms = best_ms([&] { wr<WIDTH, POLICY, ORDER, ARRAYS><<<blocks, 256>>>(g_o1, ARRAYS == 2 ? g_o2 : g_o1, n); });
  printf("write %2d B/lane %-6s %-15s %2d wg/CU %d array(s): %7.3f ms  %.2f TB/s  (%.3f of 8)\n", WIDTH, pname(POLICY),
         oname(ORDER), wg_per_cu, ARRAYS, ms, total / ms * 1e-9, total / ms * 1e-9 / 8);
  fflush(stdout);
}

template <int POLICY, int ORDER>
static void one_rw(int wg_per_cu) {
  const size_t n = g_bytes_per_array / 16;
  const int blocks = 256 * wg_per_cu;
  const float ms = best_ms([&] { rw<POLICY, ORDER><<<blocks, 256>>>((const f4 *)g_in, (f4 *)g_o1, (f4 *)g_o2, n); });
  const double total = 3.0 * g_bytes_per_array;
  printf("read 1 : write 2  %-6s %-15s %2d wg/CU: %7.3f ms  %.2f TB/s  (%.3f of 8)\n", pname(POLICY), oname(ORDER), wg_per_cu, ms,
         total / ms * 1e-9, total / ms * 1e-9 / 8);
  fflush(stdout);
}

template <int POLICY, int U, int BLOCK, bool LOAD_NT>
static void one_rwu(int wg_per_cu) {
  const size_t n = g_bytes_per_array / 16;
  const int blocks = 256 * wg_per_cu;
  const float ms = best_ms([&] { rwu<POLICY, U, BLOCK, LOAD_NT><<<blocks, BLOCK>>>((const f4 *)g_in, (f4 *)g_o1, (f4 *)g_o2, n); });
  const double total = 3.0 * g_bytes_per_array;
  printf("read 1 : write 2  %-6s loads %-5s tile %2d x %3d lanes, %2d wg/CU: %7.3f ms  %.2f TB/s  (%.3f of 8)\n", pname(POLICY),
         LOAD_NT ? "nt" : "plain", U, BLOCK, wg_per_cu, ms, total / ms * 1e-9, total / ms * 1e-9 / 8);
  fflush(stdout);
}

template <int POLICY, bool LOAD_NT>
static void rwu_grid() {
  for (int w : {1, 2, 4}) {
    one_rwu<POLICY, 1, 256, LOAD_NT>(w);
    one_rwu<POLICY, 2, 256, LOAD_NT>(w);
    one_rwu<POLICY, 4, 256, LOAD_NT>(w);
    one_rwu<POLICY, 8, 256, LOAD_NT>(w);
    one_rwu<POLICY, 16, 256, LOAD_NT>(w);
  }
  for (int w : {1, 2}) {
    one_rwu<POLICY, 2, 512, LOAD_NT>(w);
    one_rwu<POLICY, 4, 512, LOAD_NT>(w);
    one_rwu<POLICY, 8, 512, LOAD_NT>(w);
    one_rwu<POLICY, 4, 1024, LOAD_NT>(w);
  }
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 4.8;
  g_bytes_per_array = (size_t)(gb * 1e9 / 2) / 65536 * 65536;
  CHECK(hipMalloc(&g_o1, g_bytes_per_array * 2));  // one array of everything, or two halves
  g_o2 = g_o1 + g_bytes_per_array;
  CHECK(hipMalloc(&g_in, g_bytes_per_array));
  CHECK(hipMemset(g_in, 0, g_bytes_per_array));
  CHECK(hipEventCreate(&g_e0));
  CHECK(hipEventCreate(&g_e1));
  printf("%.2f GB written per launch\n", g_bytes_per_array * 2e-9);
  if (argc > 2 && !strcmp(argv[2], "rwu")) {
    rwu_grid<NT, false>();
    rwu_grid<PLAIN, false>();
    rwu_grid<NT, true>();
    return 0;
  }
  // width x policy, grid-stride, 8 workgroups per CU, two arrays
  one<16, PLAIN, STRIDE, 2>(8);
  one<16, NT, STRIDE, 2>(8);
  one<16, SC1, STRIDE, 2>(8);
  one<16, SC01, STRIDE, 2>(8);
  one<8, PLAIN, STRIDE, 2>(8);
  one<8, NT, STRIDE, 2>(8);
  one<4, PLAIN, STRIDE, 2>(8);
  one<4, NT, STRIDE, 2>(8);
  // one array
  one<16, PLAIN, STRIDE, 1>(8);
  one<16, NT, STRIDE, 1>(8);
  one<4, PLAIN, STRIDE, 1>(8);
  // order
  one<16, PLAIN, PER_WG, 2>(8);
  one<16, NT, PER_WG, 2>(8);
  one<16, PLAIN, PER_WAVE, 2>(8);
  one<16, NT, PER_WAVE, 2>(8);
  // grid
  for (int w : {1, 2, 4, 16, 32}) one<16, NT, STRIDE, 2>(w);
  for (int w : {1, 2, 4, 16, 32}) one<16, PLAIN, STRIDE, 2>(w);
  for (int w : {2, 4}) one<16, NT, PER_WG, 2>(w);
  // the real shape
  for (int w : {2, 4, 8, 16}) one_rw<PLAIN, STRIDE>(w);
  for (int w : {2, 4, 8, 16}) one_rw<NT, STRIDE>(w);
  one_rw<SC1, STRIDE>(8);
  one_rw<PLAIN, PER_WG>(8);
  one_rw<NT, PER_WG>(8);
  one_rw<NT, PER_WG>(2);
  return 0;
}
