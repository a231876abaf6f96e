// io_ceiling.hip — what does MI355X give a kernel with the posterior kernels' traffic shape
// (read one fp64 row per site, write two)?  Times bare I/O skeletons with HIP events so that the
// generated kernels' staging (elim_codegen.cpp kernel_shell) can be compared against a ceiling
// measured in the same access pattern rather than against a copy.
//   ew8 / ew16 / ew16nt   elementwise streaming, no LDS: 8 or 16 B per lane, non-temporal or not
//   ew8c                  elementwise, one contiguous range per workgroup
//   lds8c / lds8i         kernel_shell's structure (stage in, row per lane, stage out twice) with
//                         contiguous chunk ranges per workgroup / interleaved chunks
//   lds16c / lds16i       the same with 16 B per lane on the global side
// usage: io_ceiling [n_members=10] [n_sites=8000000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ __launch_bounds__(256) void ew8(const double *__restrict__ in, double *__restrict__ o1, double *__restrict__ o2,
                                           size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double v = in[i];
    o1[i] = v * 2;
    o2[i] = v + 1;
  }
}

__global__ __launch_bounds__(256) void rd16(const double2 *__restrict__ in, double *__restrict__ o, size_t n) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = in[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) o[0] = acc;
}

__global__ __launch_bounds__(256) void wr16(double2 *__restrict__ o1, double2 *__restrict__ o2, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    o1[i] = double2{1.0, 2.0};
    o2[i] = double2{3.0, 4.0};
  }
}

__global__ __launch_bounds__(256) void ew16(const double2 *__restrict__ in, double2 *__restrict__ o1,
                                            double2 *__restrict__ o2, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = in[i];
    o1[i] = double2{v.x * 2, v.y * 2};
    o2[i] = double2{v.x + 1, v.y + 1};
  }
}

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void ew16nt(const v2d *__restrict__ in, v2d *__restrict__ o1, v2d *__restrict__ o2,
                                              size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v2d v = __builtin_nontemporal_load(in + i);
    __builtin_nontemporal_store(v * 2, o1 + i);
    __builtin_nontemporal_store(v + 1, o2 + i);
  }
}

__global__ __launch_bounds__(256) void ew8c(const double *__restrict__ in, double *__restrict__ o1,
                                            double *__restrict__ o2, size_t n) {
  const size_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = in[i];
    o1[i] = v * 2;
    o2[i] = v + 1;
  }
}

// kernel_shell's structure.  W3 doubles per site, BT sites per chunk, padded LDS rows.
template <int W3, int BT, bool INTERLEAVED, int VEC>
__global__ __launch_bounds__(BT, 2) void lds_shell(const double *__restrict__ in, double *__restrict__ o1,
                                                   double *__restrict__ o2, long n_sites) {
  constexpr int ROW = W3 | 1;
  __shared__ double s_io[BT * ROW];
  const int tid = threadIdx.x;
  const long chunks = (n_sites + BT - 1) / BT;
  const long per_wg = (chunks + gridDim.x - 1) / gridDim.x;
  long c_lo, c_hi, c_step;
  if (INTERLEAVED) {
    c_lo = blockIdx.x, c_hi = chunks, c_step = gridDim.x;
  } else {
    c_lo = (long)blockIdx.x * per_wg, c_hi = c_lo + per_wg < chunks ? c_lo + per_wg : chunks, c_step = 1;
  }
  double *row = s_io + tid * ROW;
  for (long ch = c_lo; ch < c_hi; ch += c_step) {
    const long site0 = ch * BT;
    const int ns = n_sites - site0 < BT ? (int)(n_sites - site0) : BT;
    const int nel = ns * W3;
    __syncthreads();
    if (VEC == 1) {
#pragma unroll
      for (int k = 0; k < W3; ++k) {
        const int e = tid + k * BT;
        if (e < nel) s_io[(e / W3) * ROW + e % W3] = in[site0 * W3 + e];
      }
    } else {
      const v2d *src = (const v2d *)(in + site0 * W3);
#pragma unroll
      for (int k = 0; k < (W3 + 1) / 2; ++k) {
        const int e2 = tid + k * BT, e = 2 * e2;
        if (e < nel) {
          const v2d v = src[e2];
          s_io[(e / W3) * ROW + e % W3] = v.x;
          s_io[((e + 1) / W3) * ROW + (e + 1) % W3] = v.y;
        }
      }
    }
    __syncthreads();
    double l[W3];
#pragma unroll
    for (int k = 0; k < W3; ++k) l[k] = row[k];
    double acc = 0;
#pragma unroll
    for (int k = 0; k < W3; ++k) acc += l[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W3; ++k) row[k] = l[k] / acc;
    __syncthreads();
    auto stage_out = [&](double *__restrict__ dst) {
      if (VEC == 1) {
#pragma unroll
        for (int k = 0; k < W3; ++k) {
          const int e = tid + k * BT;
          if (e < nel) dst[site0 * W3 + e] = s_io[(e / W3) * ROW + e % W3];
        }
      } else {
        v2d *d2 = (v2d *)(dst + site0 * W3);
#pragma unroll
        for (int k = 0; k < (W3 + 1) / 2; ++k) {
          const int e2 = tid + k * BT, e = 2 * e2;
          if (e < nel) {
            v2d v;
            v.x = s_io[(e / W3) * ROW + e % W3];
            v.y = s_io[((e + 1) / W3) * ROW + (e + 1) % W3];
            d2[e2] = v;
          }
        }
      }
    };
    stage_out(o2);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W3; ++k) row[k] = l[k] * acc;
    __syncthreads();
    stage_out(o1);
  }
}


#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// The generated kernels' shell as of round 1: 16-B global accesses, LDS-only barriers, next chunk
// prefetched into registers.  WAVEPRIV: every wave stages its own 64 sites (no workgroup barrier).
// WORK: dependent fp64 FMAs per element between the phases (stands in for the posterior's arithmetic).
template <int W3, int BT, bool PREFETCH, bool WAVEPRIV, int WORK, bool NT>
__global__ __launch_bounds__(BT, 2) void shell2(const double *__restrict__ in, double *__restrict__ o1,
                                                double *__restrict__ o2, long n_sites) {
  constexpr int ROW = W3 | 1;
  constexpr int G = WAVEPRIV ? 64 : BT;  // sites staged together
  constexpr int K2 = (W3 * G / 2 + G - 1) / G;
  __shared__ double s_all[BT * ROW];
  const int tid = WAVEPRIV ? (threadIdx.x & 63) : threadIdx.x;
  double *s_io = s_all + (WAVEPRIV ? (threadIdx.x >> 6) * 64 * ROW : 0);
  const long groups = n_sites / G;  // benchmark sizes are multiples of BT
  const long gid = WAVEPRIV ? (long)blockIdx.x * (BT / 64) + (threadIdx.x >> 6) : blockIdx.x;
  const long ngr = WAVEPRIV ? (long)gridDim.x * (BT / 64) : gridDim.x;
  const long per = (groups + ngr - 1) / ngr;
  const long c_lo = gid * per, c_hi = c_lo + per < groups ? c_lo + per : groups;
  double *row = s_io + tid * ROW;
  v2d pre[K2];
  bool have_pre = false;
  auto sync = [&] { if (WAVEPRIV) WAVE_SYNC(); else LDS_BARRIER(); };
  for (long ch = c_lo; ch < c_hi; ++ch) {
    const long site0 = ch * G;
    sync();
    if (!have_pre) {
      const v2d *src = (const v2d *)(in + site0 * W3);
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        const int e2 = tid + k * G;
        if (2 * e2 < W3 * G) pre[k] = NT ? __builtin_nontemporal_load(src + e2) : src[e2];
      }
    }
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      const int e = 2 * (tid + k * G);
      if (e < W3 * G) {
        s_io[(e / W3) * ROW + e % W3] = pre[k].x;
        s_io[((e + 1) / W3) * ROW + (e + 1) % W3] = pre[k].y;
      }
    }
    sync();
    double l[W3];
#pragma unroll
    for (int k = 0; k < W3; ++k) l[k] = row[k];
    double acc = 0;
#pragma unroll
    for (int k = 0; k < W3; ++k) acc += l[k];
    for (int w = 0; w < WORK; ++w) {
#pragma unroll
      for (int k = 0; k < W3; ++k) l[k] = l[k] * 0.999 + acc * 1e-9;
    }
    sync();
#pragma unroll
    for (int k = 0; k < W3; ++k) row[k] = l[k] / acc;
    sync();
    auto stage_out = [&](double *__restrict__ dst) {
      v2d *d2 = (v2d *)(dst + site0 * W3);
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        const int e2 = tid + k * G, e = 2 * e2;
        if (e < W3 * G) {
          v2d v;
          v.x = s_io[(e / W3) * ROW + e % W3];
          v.y = s_io[((e + 1) / W3) * ROW + (e + 1) % W3];
          if (NT) __builtin_nontemporal_store(v, d2 + e2); else d2[e2] = v;
        }
      }
    };
    stage_out(o2);
    have_pre = PREFETCH && ch + 1 < c_hi;
    if (have_pre) {
      const v2d *src = (const v2d *)(in + (site0 + G) * W3);
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        const int e2 = tid + k * G;
        if (2 * e2 < W3 * G) pre[k] = NT ? __builtin_nontemporal_load(src + e2) : src[e2];
      }
    }
    sync();
#pragma unroll
    for (int k = 0; k < W3; ++k) row[k] = l[k] * acc;
    sync();
    stage_out(o1);
  }
}



template <class F>
static void time_it(const char *name, size_t bytes, F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("%-28s %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 8000);
  fflush(stdout);
}

template <int W3>
static void run(long n_sites) {
  const size_t n = (size_t)n_sites * W3;
  double *in, *o1, *o2;
  CHECK(hipMalloc(&in, n * 8));
  CHECK(hipMalloc(&o1, n * 8));
  CHECK(hipMalloc(&o2, n * 8));
  CHECK(hipMemset(in, 0x3f, n * 8));
  const size_t bytes = 3 * n * 8;
  printf("W3=%d, %ld sites: %.2f GB read + %.2f GB written per launch\n", W3, n_sites, n * 8e-9, n * 16e-9);
  for (int g : {2048, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "ew8 grid %d", g);
    time_it(nm, bytes, [&] { ew8<<<g, 256>>>(in, o1, o2, n); });
    snprintf(nm, sizeof nm, "ew16 grid %d", g);
    time_it(nm, bytes, [&] { ew16<<<g, 256>>>((const double2 *)in, (double2 *)o1, (double2 *)o2, n / 2); });
    snprintf(nm, sizeof nm, "ew16nt grid %d", g);
    time_it(nm, bytes, [&] { ew16nt<<<g, 256>>>((const v2d *)in, (v2d *)o1, (v2d *)o2, n / 2); });
  }
  time_it("ew8c grid 2048", bytes, [&] { ew8c<<<2048, 256>>>(in, o1, o2, n); });
  time_it("read only 16 B/lane", bytes / 3, [&] { rd16<<<4096, 256>>>((const double2 *)in, o1, n / 2); });
  time_it("write only 16 B/lane", bytes / 3 * 2, [&] { wr16<<<4096, 256>>>((double2 *)o1, (double2 *)o2, n / 2); });
  constexpr int BT = 256;
  int nb = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lds_shell<W3, BT, false, 1>, BT, 0));
  printf("lds_shell: %d workgroups of %d per CU\n", nb, BT);
  const int grid = 256 * nb;
  time_it("lds8c", bytes, [&] { lds_shell<W3, BT, false, 1><<<grid, BT>>>(in, o1, o2, n_sites); });
  time_it("lds8i", bytes, [&] { lds_shell<W3, BT, true, 1><<<grid, BT>>>(in, o1, o2, n_sites); });
  time_it("lds16c", bytes, [&] { lds_shell<W3, BT, false, 2><<<grid, BT>>>(in, o1, o2, n_sites); });
  time_it("lds16i", bytes, [&] { lds_shell<W3, BT, true, 2><<<grid, BT>>>(in, o1, o2, n_sites); });
#define S2(P, WP, WK, NT) \
  time_it("shell2 pre=" #P " wavepriv=" #WP " work=" #WK " nt=" #NT, bytes, \
          [&] { shell2<W3, BT, P, WP, WK, NT><<<grid, BT>>>(in, o1, o2, n_sites); })
  S2(false, false, 0, false);
  S2(true, false, 0, false);
  S2(false, true, 0, false);
  S2(true, true, 0, false);
  S2(true, true, 0, true);
  S2(true, false, 0, true);
  S2(true, false, 8, false);
  S2(true, true, 8, false);
  S2(true, false, 32, false);
  S2(true, true, 32, false);
  time_it("lds8c 64/wg", bytes, [&] { lds_shell<W3, 64, false, 1><<<grid * 4, 64>>>(in, o1, o2, n_sites); });
  time_it("lds16i 64/wg", bytes, [&] { lds_shell<W3, 64, true, 2><<<grid * 4, 64>>>(in, o1, o2, n_sites); });
  CHECK(hipFree(in));
  CHECK(hipFree(o1));
  CHECK(hipFree(o2));
}

int main(int argc, char **argv) {
  const int n_members = argc > 1 ? atoi(argv[1]) : 10;
  const long n_sites = argc > 2 ? atol(argv[2]) : 8000000;
  if (n_members == 10)
    run<30>(n_sites);
  else if (n_members == 5)
    run<15>(n_sites);
  else {
    fprintf(stderr, "n_members must be 5 or 10\n");
    return 2;
  }
  return 0;
}
