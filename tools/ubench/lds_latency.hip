// Micro-benchmarks of the LDS / f64 primitives the step kernel is built from (one wave per workgroup, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_latency.hip -o build/lds_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define N_IT 256

__global__ void __launch_bounds__(64) k_bench(long long* out, int mode, int stride) {
  __shared__ double lds[4096];
  __shared__ unsigned idx[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) { lds[i] = 1.0 + i * 1e-9; idx[i] = (unsigned)((i * 17 + 5) & 4095); }
  __syncthreads();
  long long t0 = clock64(), w0 = wall_clock64();
  double acc = 0.0;
  unsigned p = (unsigned)lane;
  if (mode == 0) {            // dependent u32 LDS loads (pointer chase): latency of ds_read_b32
    for (int i = 0; i < N_IT; ++i) p = idx[p];
    acc = p;
  } else if (mode == 1) {     // dependent f64 loads: index from the loaded double
    double v = lds[lane];
    for (int i = 0; i < N_IT; ++i) { v = lds[((unsigned)(v * 64.0) + lane) & 4095]; }
    acc = v;
  } else if (mode == 2) {     // independent ds_add_f64, distinct addresses (stride in doubles)
    for (int i = 0; i < N_IT; ++i) __hip_atomic_fetch_add(&lds[(lane * stride + i) & 4095], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if (mode == 3) {     // ds_add_f64, groups of `stride` adjacent lanes hit the same address
    for (int i = 0; i < N_IT; ++i) __hip_atomic_fetch_add(&lds[((lane / stride) * 7 + i) & 4095], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if (mode == 4) {     // atomic add then dependent read of the same location (round trip incl. atomic)
    for (int i = 0; i < N_IT; ++i) {
      __hip_atomic_fetch_add(&lds[lane], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += lds[lane ^ 1];
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else if (mode == 5) {     // dependent f64 fma chain
    double v = lds[lane];
    for (int i = 0; i < N_IT; ++i) v = fma(v, 1.0000001, 1e-9);
    acc = v;
  } else if (mode == 6) {     // 4 independent fma chains (throughput)
    double a = lds[lane], b = lds[lane + 64], c = lds[lane + 128], d = lds[lane + 192];
    for (int i = 0; i < N_IT; ++i) { a = fma(a, 1.0000001, 1e-9); b = fma(b, 1.0000001, 1e-9); c = fma(c, 1.0000001, 1e-9); d = fma(d, 1.0000001, 1e-9); }
    acc = a + b + c + d;
  } else if (mode == 7) {     // ds_read_b128 throughput: independent 16-byte loads
    const double2* l2 = (const double2*)lds;
    double2 s = {0, 0};
    for (int i = 0; i < N_IT; ++i) { double2 v = l2[(lane * stride + i * 64) & 2047]; s.x += v.x; s.y += v.y; }
    acc = s.x + s.y;
  } else if (mode == 8) {     // plain store + wait + dependent load (LDS round trip write->read)
    for (int i = 0; i < N_IT; ++i) {
      lds[lane] = acc;
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc += lds[(lane + 1) & 63];
    }
  } else if (mode == 9) {     // readfirstlane of an LDS load (uniform scalar from LDS): the old level-boundary pattern
    unsigned q = 0;
    for (int i = 0; i < N_IT; ++i) { q = __builtin_amdgcn_readfirstlane(idx[(q + i) & 4095]); }
    acc = q;
  } else if (mode == 10) {    // f64 reciprocal + 2 Newton steps, dependent
    double v = lds[lane];
    for (int i = 0; i < N_IT; ++i) { double r = __builtin_amdgcn_rcp(v); r = fma(fma(-v, r, 1.0), r, r); r = fma(fma(-v, r, 1.0), r, r); v = r + 0.5; }
    acc = v;
  } else if (mode == 11) {    // sincos f64, dependent
    double v = lds[lane] * 0.1;
    for (int i = 0; i < N_IT; ++i) { double s, c; sincos(v, &s, &c); v = s * 0.3 + c * 0.01; }
    acc = v;
  } else if (mode == 12) {    // global load latency (L2 hits): pointer chase through a small global table
    const unsigned* g = (const unsigned*)(out + 1024);
    for (int i = 0; i < N_IT; ++i) p = g[p & 1023];
    acc = p;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (acc == 12345.678) out[1000] = 1;   // keep results alive
  if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  long long* d;
  hipMalloc(&d, 16384 * sizeof(long long));
  hipMemset(d, 0, 16384 * sizeof(long long));
  {
    unsigned h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (unsigned)((i * 17 + 5) & 1023);
    hipMemcpy(d + 1024, h, sizeof h, hipMemcpyHostToDevice);
  }
  struct { int mode, stride; const char* name; } T[] = {
    {0, 0, "dependent ds_read_b32 (pointer chase)"}, {1, 0, "dependent ds_read_b64 + cvt"},
    {2, 1, "ds_add_f64 distinct addr stride 1"}, {2, 4, "ds_add_f64 distinct addr stride 4 doubles"},
    {3, 1, "ds_add_f64 conflicts x1"}, {3, 2, "ds_add_f64 same addr x2"}, {3, 4, "ds_add_f64 same addr x4"}, {3, 8, "ds_add_f64 same addr x8"},
    {3, 64, "ds_add_f64 same addr x64"},
    {4, 0, "atomic + wait + read + wait"}, {5, 0, "dependent v_fma_f64"}, {6, 0, "4 independent v_fma_f64 (per 4)"},
    {7, 1, "ds_read_b128 independent stride 1"}, {7, 2, "ds_read_b128 independent stride 2 (32 B blocks)"},
    {8, 0, "store + wait + load round trip"}, {9, 0, "LDS load -> readfirstlane dependent"},
    {10, 0, "rcp f64 + 2 Newton, dependent"}, {11, 0, "sincos f64 dependent"}, {12, 0, "global load pointer chase (L2 hit)"}};
  for (auto& t : T) {
    long long h[2];
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, d, t.mode, t.stride);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f clk/iter   (%lld clk, %lld x 10ns wall -> %.0f MHz)\n", t.name, (double)h[0] / N_IT, h[0], h[1],
           h[1] ? (double)h[0] / (h[1] * 0.01) : 0.0);
  }
  return 0;
}
