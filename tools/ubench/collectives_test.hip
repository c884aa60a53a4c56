// Unit test of the wave collectives of ppn_solve.inc (DPP based) against host results.  GPU box only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "../../include/ppn.h"
#include "../../pypownet_amd/csrc/ppn_device.h"
#include "../../pypownet_amd/csrc/ppn_solve.inc"

__global__ void __launch_bounds__(64) k_test(const int* in, const double* din, int* out, double* dout, int m) {
  __shared__ u16 a[600];
  const int lane0 = threadIdx.x;
  out[lane0] = wave_sum_i(in[lane0]);
  out[64 + lane0] = wave_min_i(in[lane0]);
  out[128 + lane0] = wave_incl_scan_i(in[lane0], lane0);
  dout[lane0] = wave_maxnan_d(din[lane0]);
  for (int i = lane0; i < m; i += 64) a[i] = (u16)(in[i % 64] & 7);
  __syncthreads();
  const int tot = scan_u16(a, m, lane0);
  __syncthreads();
  for (int i = lane0; i <= m; i += 64) out[192 + i] = a[i];
  if (lane0 == 0) out[1900] = tot;
  // divergent producer followed by a collective (the way the kernels use them)
  int v = 0;
  if (lane0 & 1) { for (int k = 0; k < (lane0 & 7); ++k) v += in[(lane0 + k) & 63] & 3; }
  out[1000 + lane0] = wave_sum_i(v);
  out[1100 + lane0] = v;
}

int main() {
  int bad = 0;
  for (int trial = 0; trial < 50; ++trial) {
    int h[64]; double hd[64];
    srand(trial + 1);
    for (int i = 0; i < 64; ++i) { h[i] = (rand() % 2001) - 1000; hd[i] = (rand() % 1000) * 0.37; }
    if (trial % 5 == 1) hd[rand() % 64] = NAN;
    const int m = 1 + rand() % 500;
    int *din_i, *dout_i; double *din_d, *dout_d;
    hipMalloc(&din_i, 64 * 4); hipMalloc(&dout_i, 2000 * 4); hipMalloc(&din_d, 64 * 8); hipMalloc(&dout_d, 64 * 8);
    hipMemcpy(din_i, h, 256, hipMemcpyHostToDevice); hipMemcpy(din_d, hd, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, din_i, din_d, dout_i, dout_d, m);
    int o[2000]; double od[64];
    hipMemcpy(o, dout_i, sizeof o, hipMemcpyDeviceToHost); hipMemcpy(od, dout_d, sizeof od, hipMemcpyDeviceToHost);
    int sum = 0, mn = 0x7fffffff; double mx = 0.0; bool nan = false;
    for (int i = 0; i < 64; ++i) { sum += h[i]; if (h[i] < mn) mn = h[i]; if (hd[i] != hd[i]) nan = true; else if (hd[i] > mx) mx = hd[i]; }
    int run = 0;
    for (int i = 0; i < 64; ++i) {
      run += h[i];
      if (o[i] != sum) { if (!bad++) printf("trial %d sum lane %d: %d vs %d\n", trial, i, o[i], sum); }
      if (o[64 + i] != mn) { if (!bad++) printf("trial %d min lane %d: %d vs %d\n", trial, i, o[64 + i], mn); }
      if (o[128 + i] != run) { if (!bad++) printf("trial %d scan lane %d: %d vs %d\n", trial, i, o[128 + i], run); }
      if (nan ? (od[i] == od[i]) : (od[i] != mx)) { if (!bad++) printf("trial %d maxnan lane %d: %g vs %g nan=%d\n", trial, i, od[i], mx, (int)nan); }
    }
    run = 0;
    for (int i = 0; i < m; ++i) { if (o[192 + i] != run) { if (!bad++) printf("trial %d scan_u16[%d] of %d: %d vs %d\n", trial, i, m, o[192 + i], run); } run += h[i % 64] & 7; }
    if (o[192 + m] != run || o[1900] != run) { if (!bad++) printf("trial %d scan total %d %d vs %d\n", trial, o[192 + m], o[1900], run); }
    int s2 = 0;
    for (int i = 0; i < 64; ++i) s2 += o[1100 + i];
    for (int i = 0; i < 64; ++i) if (o[1000 + i] != s2) { if (!bad++) printf("trial %d divergent sum lane %d: %d vs %d\n", trial, i, o[1000 + i], s2); }
    hipFree(din_i); hipFree(dout_i); hipFree(din_d); hipFree(dout_d);
  }
  printf(bad ? "FAILED (%d mismatches)\n" : "collectives OK\n", bad);
  return bad != 0;
}
