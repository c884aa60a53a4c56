// hbm_counter_calib.hip -- kernels that move a KNOWN number of bytes, for calibrating the rocprofv3 counters bench.py's
// roofline.traffic is made of (VERDICT r05 #7b: "calibrate WRITE_SIZE with a known-size write kernel").
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_counter_calib.hip -o build/hbm_counter_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dir> -o p --output-format csv -- build/hbm_counter_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o p --output-format csv -- build/hbm_counter_calib
//   python tools/ubench/hbm_counter_calib.py <dir_write> <dir_fetch>      -> JSON: counter units per byte actually moved
//
// Kernels (each launched 3 times on a fresh 256 MiB buffer, far beyond the 256 MB MALL only in sum -- the counters sit at the
// L2 <-> fabric boundary, so what leaves L2 is what is counted):
//   calib_write16   every lane stores 16 B, fully coalesced, the whole buffer          (256 MiB written, nothing read)
//   calib_write8    every lane stores ONE double, coalesced                            (256 MiB written)
//   calib_write1    every lane stores ONE byte, consecutive lanes consecutive bytes    (64 MiB written: byte rows like the engine's u8 state)
//   calib_write_row every WORKGROUP writes a 186-byte row at a 186-byte stride         (the engine's [env x nl] u8 rows: partial 64 B / 128 B lines at
//                                                                                       both ends of every row)
//   calib_read16    every lane loads 16 B, coalesced, result kept alive by a rare store (256 MiB read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_write16(uint4* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void calib_write8(double* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (double)i;
}
__global__ void calib_write1(unsigned char* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (unsigned char)i;
}
__global__ void calib_write_row(unsigned char* p, int row) {
  unsigned char* r = p + (size_t)blockIdx.x * row;
  for (int k = threadIdx.x; k < row; k += blockDim.x) r[k] = (unsigned char)k;
}
__global__ void calib_read16(const uint4* p, size_t n, unsigned* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint4 v = p[i]; if (v.x == 0xDEADBEEFu && v.y == 0x12345678u) *sink = v.z; }
}

int main() {
  const size_t BYTES = (size_t)256 << 20;
  void* buf = nullptr; unsigned* sink = nullptr;
  CHECK(hipMalloc(&buf, BYTES)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 0, BYTES));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    { const size_t n = BYTES / 16; hipLaunchKernelGGL(calib_write16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint4*)buf, n); }
    CHECK(hipDeviceSynchronize());
    { const size_t n = BYTES / 8; hipLaunchKernelGGL(calib_write8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (double*)buf, n); }
    CHECK(hipDeviceSynchronize());
    { const size_t n = BYTES / 4; hipLaunchKernelGGL(calib_write1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned char*)buf, n); }
    CHECK(hipDeviceSynchronize());
    { const int row = 186; const unsigned rows = (unsigned)((BYTES / 4) / row); hipLaunchKernelGGL(calib_write_row, dim3(rows), dim3(64), 0, 0, (unsigned char*)buf, row); }
    CHECK(hipDeviceSynchronize());
    { const size_t n = BYTES / 16; hipLaunchKernelGGL(calib_read16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint4*)buf, n, sink); }
    CHECK(hipDeviceSynchronize());
  }
  printf("bytes: write16 %zu write8 %zu write1 %zu write_row %zu read16 %zu\n", BYTES, BYTES, BYTES / 4, (size_t)((BYTES / 4) / 186) * 186, BYTES);
  return 0;
}
