// ppn_sincos (ppn_solve.inc) on the GPU against the same source compiled for the host (bit for bit) and against long-double libm
// (ulp), plus its cost next to the library sincos.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/sincos_test.hip -o build/sincos_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/ppn.h"
#include "../../pypownet_amd/csrc/ppn_device.h"
#include "../../pypownet_amd/csrc/ppn_solve.inc"

// the host twin: the body of ppn_sincos, compiled by the host pass of this file (same builtins)
static void host_sincos(double x, double* sn, double* cs) {
  const double k = __builtin_rint(x * 0.63661977236758134308);
  const double hi = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  const double r = __builtin_fma(-k, 6.12323399573676603587e-17, hi);
  const double t = __builtin_fma(-k, 6.12323399573676603587e-17, hi - r);
  const double z = r * r, v = z * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double a = __builtin_fma(-v, ps, 0.5 * t);
  const double b = __builtin_fma(z, a, -t);
  const double s = r - __builtin_fma(v, 1.66666666666666324348e-01, b);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double rt = r * t;
  const double c = w + (((1.0 - w) - hz) + __builtin_fma(z, z * pc, -rt));
  const int q = (int)k & 3;
  const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}
__global__ void k_eval(const double* x, double* s, double* c, double* ls, double* lc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ppn_sincos(x[i], &s[i], &c[i]); sincos(x[i], &ls[i], &lc[i]); }
}
template <int WHICH> __global__ void k_time(double x0, double* out, long long* clk) {
  double x = x0 + 1e-3 * threadIdx.x, acc = 0.0;
  const long long t0 = clock64();
  for (int it = 0; it < 256; ++it) {      // one dependent chain, like the head of a Newton iteration
    double s, c;
    if (WHICH) ppn_sincos(x, &s, &c); else sincos(x, &s, &c);
    acc += s; x = c * 3.0;
  }
  const long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[0] = (t1 - t0) / 256;
}
static double ulp_of(double x) { x = fabs(x); int e; frexp(x, &e); return ldexp(1.0, e - 53); }
int main() {
  std::vector<double> x;
  srand48(7);
  for (int i = 0; i < 4000000; ++i) x.push_back((drand48() * 2 - 1) * 3.2);
  for (int i = 0; i < 500000; ++i) x.push_back((drand48() * 2 - 1) * 10.0);
  for (int k = -6; k <= 6; ++k) {
    double b = k * M_PI_2, y = b;
    for (int j = 0; j < 500; ++j) { x.push_back(y); y = nextafter(y, 1e9); }
    y = b; for (int j = 0; j < 500; ++j) { x.push_back(y); y = nextafter(y, -1e9); }
    for (int j = 0; j < 5000; ++j) x.push_back(b + (drand48() * 2 - 1) * ldexp(1.0, -(int)(drand48() * 60)));
  }
  for (int j = 0; j < 50000; ++j) x.push_back((drand48() * 2 - 1) * ldexp(1.0, -(int)(drand48() * 300)));
  x.push_back(0.0); x.push_back(NAN); x.push_back(INFINITY); x.push_back(1e6); x.push_back(-12345.678);
  const int n = (int)x.size();
  double *dx, *ds, *dc, *dls, *dlc;
  if (hipMalloc(&dx, n * 8) || hipMalloc(&ds, n * 8) || hipMalloc(&dc, n * 8) || hipMalloc(&dls, n * 8) || hipMalloc(&dlc, n * 8)) return 2;
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, 0, dx, ds, dc, dls, dlc, n);
  std::vector<double> s(n), c(n), ls(n), lc(n);
  (void)hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ls.data(), dls, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(lc.data(), dlc, n * 8, hipMemcpyDeviceToHost);
  long diff = 0; double ms = 0, mc = 0, mls = 0, mlc = 0;
  for (int i = 0; i < n; ++i) {
    double hs, hc; host_sincos(x[i], &hs, &hc);
    if (memcmp(&hs, &s[i], 8) || memcmp(&hc, &c[i], 8)) { if (!(hs != hs && s[i] != s[i])) ++diff; }
    if (!(fabs(x[i]) <= 1e300) || fabs(x[i]) > 10.0) continue;
    const long double rs = sinl((long double)x[i]), rc = cosl((long double)x[i]);
    const double us = ulp_of((double)rs), uc = ulp_of((double)rc);
    ms = fmax(ms, fabs((double)((long double)s[i] - rs)) / us); mc = fmax(mc, fabs((double)((long double)c[i] - rc)) / uc);
    mls = fmax(mls, fabs((double)((long double)ls[i] - rs)) / us); mlc = fmax(mlc, fabs((double)((long double)lc[i] - rc)) / uc);
  }
  printf("%d arguments: device != host twin in %ld; max ulp error of ppn_sincos: sin %.3f cos %.3f (library sincos on the device: %.3f / %.3f)\n", n, diff, ms, mc, mls, mlc);
  printf("sincos(1e6): ppn %.17g %.17g  library %.17g %.17g\n", s[n - 2], c[n - 2], ls[n - 2], lc[n - 2]);
  printf("sincos(nan), sincos(inf): %g %g, %g %g\n", s[n - 4], c[n - 4], s[n - 3], c[n - 3]);
  double* dout; long long* dclk; long long clk[2] = {0, 0};
  if (hipMalloc(&dout, 64 * 8) || hipMalloc(&dclk, 8)) return 2;
  hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(64), 0, 0, 0.3, dout, dclk); (void)hipMemcpy(&clk[0], dclk, 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_time<1>, dim3(1), dim3(64), 0, 0, 0.3, dout, dclk); (void)hipMemcpy(&clk[1], dclk, 8, hipMemcpyDeviceToHost);
  printf("dependent chain, clocks per call: library sincos %lld, ppn_sincos %lld\n", clk[0], clk[1]);
  return diff ? 1 : 0;
}
