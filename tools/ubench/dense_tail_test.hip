// Unit test of the register-resident dense tail (tail_gj_steps, ppn_solve.inc: Gauss-Jordan sweep, column blocks per DPP row) on the GPU: random
// diagonally dominant systems of `rows` <= M rows, identity rows marked as in the solver, against Gaussian elimination on
// the host.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPPN_TAIL_BUSES=8] tools/ubench/dense_tail_test.hip -o build/dense_tail_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../include/ppn.h"
#include "../../pypownet_amd/csrc/ppn_device.h"
#include "../../pypownet_amd/csrc/ppn_solve.inc"

constexpr int M = PPN_TAIL_ROWS;
// lane = 16 q + r: row r, columns 4q .. 4q + 3 (the layout of dense_tail, ppn_solve.inc)
__global__ void k_tail(const double* A, const double* b, double* x, int rows, unsigned idrows) {
  const int lane0 = threadIdx.x;
  const int sys = blockIdx.x;
  const int r = lane0 & 15, q = lane0 >> 4;
  double ta[4]; double ty, rs = 1.0;
  for (int j = 0; j < 4; ++j) { const int c = 4 * q + j; ta[j] = (r < rows && c < rows && c < M) ? A[((size_t)sys * M + r) * M + c] : 0.0; }
  ty = r < rows ? b[(size_t)sys * M + r] : 0.0;
  tail_gj_steps<0, M>(ta, ty, rs, rows, idrows, lane0);
  if (lane0 < rows) x[(size_t)sys * M + lane0] = ty * rs;
}

// timing: the same sweep 64 times back to back on one wave (results chained through ty)
__global__ void k_tail_time(const double* A, const double* b, double* x, long long* clk, int rows, unsigned idrows) {
  const int lane0 = threadIdx.x;
  const int r = lane0 & 15, q = lane0 >> 4;
  double a0[4]; double ty, rs = 1.0, y0;
  for (int j = 0; j < 4; ++j) { const int c = 4 * q + j; a0[j] = (r < rows && c < rows && c < M) ? A[(size_t)r * M + c] : 0.0; }
  y0 = r < rows ? b[r] : 0.0;
  ty = y0;
  const long long t0 = clock64();
  for (int rep = 0; rep < 64; ++rep) {
    double ta[4];
    for (int j = 0; j < 4; ++j) ta[j] = a0[j] + 1e-300 * ty;
    ty = y0 + 1e-300 * ty;
    rs = 1.0;
    tail_gj_steps<0, M>(ta, ty, rs, rows, idrows, lane0);
    ty *= rs;
  }
  const long long t1 = clock64();
  if (lane0 < rows) x[lane0] = ty;
  if (lane0 == 0) clk[0] = (t1 - t0) / 64;
}

int main() {
  {
    std::vector<double> A((size_t)M * M, 0.0), b(M, 0.3);
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) A[(size_t)i * M + j] = (i == j) ? M + 1.0 : 0.25 + 0.01 * ((i * 7 + j * 3) % 5);
    double *dA, *db, *dx; long long* dc; long long c = 0;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dx, M * 8); hipMalloc(&dc, 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
    for (int rows = M; rows >= M - 2; rows -= 2) for (unsigned idr : {0u, 0x2u | 0x80u}) {
      hipLaunchKernelGGL(k_tail_time, dim3(1), dim3(64), 0, 0, dA, db, dx, dc, rows, idr);
      hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      printf("timing: M %d rows %2d identity rows 0x%04x: %lld clocks per Gauss-Jordan sweep\n", M, rows, idr, c);
    }
    hipFree(dA); hipFree(db); hipFree(dx); hipFree(dc);
  }
  const int NS = 64;
  int bad = 0;
  for (int rows = 2; rows <= M; rows += 2) {
    for (unsigned pat = 0; pat < 3; ++pat) {
      unsigned idrows = 0;
      if (pat == 1) idrows = 0x2u | 0x80u;                 // a PV bus (no Q row) at bus 0, another at bus 3
      if (pat == 2) idrows = 0x3u | (rows > 6 ? 0x20u : 0u);   // the reference bus at bus 0 (both rows), a PV bus at bus 2
      idrows &= (1u << rows) - 1u;
      std::vector<double> A((size_t)NS * M * M, 0.0), b((size_t)NS * M, 0.0), x((size_t)NS * M, 0.0), ref((size_t)NS * M, 0.0);
      srand(17 + rows * 3 + pat);
      for (int s = 0; s < NS; ++s) {
        for (int i = 0; i < rows; ++i) {
          const bool idi = (idrows >> i) & 1u;
          for (int j = 0; j < rows; ++j) {
            const bool idj = (idrows >> j) & 1u;
            double v = (double)rand() / RAND_MAX - 0.5;
            if (i == j) v += (v < 0 ? -1.0 : 1.0) * rows;
            A[((size_t)s * M + i) * M + j] = (idi || idj) ? 0.0 : v;      // identity rows and their columns hold zeros (solver convention)
          }
          b[(size_t)s * M + i] = idi ? 0.0 : (double)rand() / RAND_MAX - 0.5;
        }
        // host reference
        std::vector<double> a((size_t)rows * rows), y(rows);
        for (int i = 0; i < rows; ++i) { for (int j = 0; j < rows; ++j) a[i * rows + j] = A[((size_t)s * M + i) * M + j]; y[i] = b[(size_t)s * M + i]; if ((idrows >> i) & 1u) a[i * rows + i] = 1.0; }
        for (int t = 0; t < rows; ++t) for (int i = t + 1; i < rows; ++i) { const double f = a[i * rows + t] / a[t * rows + t]; for (int j = t; j < rows; ++j) a[i * rows + j] -= f * a[t * rows + j]; y[i] -= f * y[t]; }
        for (int t = rows - 1; t >= 0; --t) { double v = y[t]; for (int j = t + 1; j < rows; ++j) v -= a[t * rows + j] * ref[(size_t)s * M + j]; ref[(size_t)s * M + t] = v / a[t * rows + t]; }
      }
      double *dA, *db, *dx;
      hipMalloc(&dA, A.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dx, x.size() * 8);
      hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
      hipMemset(dx, 0, x.size() * 8);
      hipLaunchKernelGGL(k_tail, dim3(NS), dim3(64), 0, 0, dA, db, dx, rows, idrows);
      hipMemcpy(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost);
      hipFree(dA); hipFree(db); hipFree(dx);
      double worst = 0.0;
      for (int s = 0; s < NS; ++s) for (int i = 0; i < rows; ++i) {
        if ((idrows >> i) & 1u) continue;          // (the unknown of an identity row does not exist)
        const double e = fabs(x[(size_t)s * M + i] - ref[(size_t)s * M + i]);
        if (!(e <= worst)) worst = e;
      }
      printf("M %d rows %2d identity rows 0x%04x: max |x - ref| = %.3e %s\n", M, rows, idrows, worst, worst < 1e-12 ? "ok" : "FAILED");
      if (!(worst < 1e-12)) ++bad;
    }
  }
  return bad ? 1 : 0;
}
