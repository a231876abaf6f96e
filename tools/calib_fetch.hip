// calib_fetch.hip — calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths bn_enum_kernel uses (8 B per lane loads/stores), as MI355X_MICROARCH.md asks:
// "calibrate on a known byte count in your own access pattern before trusting an absolute".
//   read8 : every lane loads one double (8 B/lane), grid-stride, known total bytes
//   read16: every lane loads one double2 (16 B/lane)
//   write8: every lane stores one double
// Run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and WRITE_SIZE in a second pass)
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void read8(const double *p, size_t n, double *out) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678) out[0] = acc;
}
__global__ void read16(const double2 *p, size_t n, double *out) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void write8(double *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}

int main() {
  const size_t bytes = size_t(1) << 30;  // 1 GiB > 256 MiB Infinity Cache
  double *a, *o;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
  hipMemset(a, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    read8<<<2048, 256>>>(a, bytes / 8, o);
    read16<<<2048, 256>>>((const double2 *)a, bytes / 16, o);
    write8<<<2048, 256>>>(a, bytes / 8);
  }
  hipDeviceSynchronize();
  printf("each kernel touches %zu bytes\n", bytes);
  return 0;
}
