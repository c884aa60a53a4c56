// one code shape of the four-word fast-decoupled kernels -- the terms of ALL rounds of lines requested
// at once, 96 registers in flight -- gave wrong flows on the GPU only.  Its ISA differs from the shipped two-round shape in
// exactly three instructions (tools/dev/one_kernel.hip, -DDEV_W=4 -DDEV_NT=0): one `global_load_dwordx4 a[0:3], ...` (a VMEM
// load that targets AGPRs) and two `ds_add_f64 vaddr, a[n:n+1]` (LDS f64 atomics whose DATA operand is an AGPR pair).  This
// micro-test runs that pair of instructions in isolation -- many loads in flight, a counted s_waitcnt, a divergent guard around
// the atomics as in the kernel -- next to the same sequence through VGPRs, and compares both with the host's sums.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/agpr_operands.hip -o build/agpr_operands ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool AGPR>
__global__ void __launch_bounds__(64) k(const double* __restrict__ src, const int* __restrict__ slot, const int* __restrict__ on,
                                        double* __restrict__ out, int rounds, int nslots) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < nslots; i += 64) lds[i] = 0.0;
  __syncthreads();
  const double* base = src + (size_t)blockIdx.x * rounds * 64 * 2;
  for (int r = 0; r < rounds; ++r) {
    const double* p = base + ((size_t)r * 64 + lane) * 2;
    const int s0 = slot[(blockIdx.x * rounds + r) * 64 + lane];
    const bool live = on[(blockIdx.x * rounds + r) * 64 + lane] != 0;
    // filler loads in flight in front of and behind the one under test (the kernel holds ~30)
    double f0 = base[(lane * 2 + 1) % (rounds * 128)], f1 = base[(lane * 2 + 3) % (rounds * 128)];
    if (AGPR) {
      d2 v;
      __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=a"(v) : "v"(p) : "memory");
      double f2 = base[(lane * 2 + 5) % (rounds * 128)];
      __asm__ volatile("s_waitcnt vmcnt(1)" ::: "memory");       // the load under test is done, one younger load still in flight
      if (live) {
        const unsigned a0 = (unsigned)(s0 * 8), a1 = (unsigned)(((s0 + 7) % nslots) * 8);
        __asm__ volatile("ds_add_f64 %0, %1" :: "v"(a0), "a"(v.x) : "memory");
        __asm__ volatile("ds_add_f64 %0, %1" :: "v"(a1), "a"(v.y) : "memory");
      }
      f0 += f2;
    } else {
      d2 v;
      __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
      double f2 = base[(lane * 2 + 5) % (rounds * 128)];
      __asm__ volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (live) {
        const unsigned a0 = (unsigned)(s0 * 8), a1 = (unsigned)(((s0 + 7) % nslots) * 8);
        __asm__ volatile("ds_add_f64 %0, %1" :: "v"(a0), "v"(v.x) : "memory");
        __asm__ volatile("ds_add_f64 %0, %1" :: "v"(a1), "v"(v.y) : "memory");
      }
      f0 += f2;
    }
    if (f0 + f1 == 1.2345e300) lds[0] = f0;      // keep the fillers alive
  }
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < nslots; i += 64) out[(size_t)blockIdx.x * nslots + i] = lds[i];
}

int main() {
  const int B = 2048, rounds = 12, nslots = 509;
  const size_t n = (size_t)B * rounds * 64;
  double* h = (double*)malloc(n * 2 * sizeof(double));
  int *hs = (int*)malloc(n * sizeof(int)), *ho = (int*)malloc(n * sizeof(int));
  srand(12345);
  for (size_t i = 0; i < n * 2; ++i) h[i] = (double)(rand() % 2000001 - 1000000) / 1024.0;      // exactly representable: sums are order independent
  for (size_t i = 0; i < n; ++i) { hs[i] = rand() % nslots; ho[i] = (rand() % 4) != 0; }
  double *d, *o; int *ds, *dn;
  hipMalloc(&d, n * 2 * sizeof(double)); hipMalloc(&o, (size_t)B * nslots * sizeof(double)); hipMalloc(&ds, n * sizeof(int)); hipMalloc(&dn, n * sizeof(int));
  hipMemcpy(d, h, n * 2 * sizeof(double), hipMemcpyHostToDevice);
  hipMemcpy(ds, hs, n * sizeof(int), hipMemcpyHostToDevice);
  hipMemcpy(dn, ho, n * sizeof(int), hipMemcpyHostToDevice);
  double* want = (double*)calloc((size_t)B * nslots, sizeof(double));
  for (int b = 0; b < B; ++b) for (int r = 0; r < rounds; ++r) for (int l = 0; l < 64; ++l) {
    const size_t i = ((size_t)b * rounds + r) * 64 + l;
    if (!ho[i]) continue;
    want[(size_t)b * nslots + hs[i]] += h[i * 2];
    want[(size_t)b * nslots + (hs[i] + 7) % nslots] += h[i * 2 + 1];
  }
  double* got = (double*)malloc((size_t)B * nslots * sizeof(double));
  int bad_total = 0;
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(o, 0, (size_t)B * nslots * sizeof(double));
    if (variant) hipLaunchKernelGGL(k<true>, dim3(B), dim3(64), nslots * sizeof(double), 0, d, ds, dn, o, rounds, nslots);
    else hipLaunchKernelGGL(k<false>, dim3(B), dim3(64), nslots * sizeof(double), 0, d, ds, dn, o, rounds, nslots);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    hipMemcpy(got, o, (size_t)B * nslots * sizeof(double), hipMemcpyDeviceToHost);
    size_t bad = 0; double worst = 0.0;
    for (size_t i = 0; i < (size_t)B * nslots; ++i) { const double e = fabs(got[i] - want[i]); if (e != 0.0) { ++bad; if (e > worst) worst = e; } }
    printf("%s data operands: %zu of %zu sums differ from the host (worst %.3g)\n", variant ? "AGPR" : "VGPR", bad, (size_t)B * nslots, worst);
    bad_total += bad != 0;
  }
  return bad_total ? 1 : 0;
}
