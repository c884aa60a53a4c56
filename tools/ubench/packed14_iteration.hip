// packed14_iteration.hip -- micro-benchmark for VERDICT r04 #5: ONE Newton iteration of FOUR 14-bus systems per wavefront, one DPP
// row (16 lanes) per system, against the 11 k cycles per iteration and environment of the one-word step kernel
// (profiles/r05_phase_profile_default14_b1024.txt).  Break-even for the packed form: 11 k / 4 = 2.75 k cycles per environment-iteration.
//
// Layout: lane = 16 g + r plays bus r (r < 14) of system g.  A lane holds row r of the dense Ybus (14 complex = 28 registers) and BOTH
// Jacobian rows of its bus -- the P_r equation and the Q_r equation over the 28 unknowns (angle_0..13, relative magnitude_0..13): 56
// registers -- plus their right-hand sides.  Everything another bus of the same system holds arrives through DPP row_newbcast:T (lane T
// of the own row of 16) as the source operand of the multiply-add itself; nothing goes through LDS.
//   evaluation      V_j of the 14 buses broadcast one by one, I_i = sum Y_ij V_j, S_i = V_i conj(I_i), T_ij = V_i conj(Y_ij V_j)
//                   -> Jacobian rows (the main kernel's formulas: magnitude unknowns are relative corrections)
//   linear solve    Gauss-Jordan in bus-major pivot order (angle_k, magnitude_k), no pivoting: pivot row k broadcast column by column,
//                   every lane sweeps its two rows; identity rows for unknowns that do not exist (reference bus, |V| of a PV bus)
//   update
// Checked against a host double-precision Newton on the same four systems (same formulas): voltages after the timed iterations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/packed14_iteration.hip -o build/packed14_iteration && build/packed14_iteration
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define NB 14
#define NSYS 4

template <int T> __device__ __forceinline__ double bcast16(double v) {
  double o;
  __asm__ volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(T));
  return o;
}
// acc += bcast_T(src) * mul   (src written long ago: no wait states needed)
template <int T> __device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul) {
  __asm__ volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
}
template <int T> __device__ __forceinline__ void fmac_bcast_self(double& acc, double mul) {
  __asm__ volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mul), "n"(T));
}
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

struct Lane {
  double yre[NB], yim[NB];     // row of Ybus
  double jp[2 * NB], jq[2 * NB];   // Jacobian rows of the P and the Q equation: columns [angle_0..13 | magnitude_0..13]
  double bp, bq;               // right-hand sides
  double vm, va, psp, qsp;
  int nv;                      // unknowns of this bus: 0 reference / idle lane, 1 PV, 2 PQ
};

// ---- evaluation: column j of the Jacobian rows and the contribution of bus j to S_i ------------------------------------------------
template <int J> __device__ __forceinline__ void eval_col(Lane& L, double vr, double vi, double& sr, double& si, int r) {
  if constexpr (J < NB) {
    const double vrj = bcast16<J>(vr), vij = bcast16<J>(vi);
    const double yr = L.yre[J], yi = L.yim[J];
    const double ar = yr * vrj - yi * vij, ai = yr * vij + yi * vrj;      // Y_ij V_j
    const double tr = vr * ar + vi * ai, ti = vi * ar - vr * ai;          // T_ij = V_i conj(Y_ij V_j)
    sr += tr; si += ti;
    // off-diagonal blocks (dP/da_j, dP/dVm_j |V_j|; dQ/da_j, dQ/dVm_j |V_j|) = (ti, tr; -tr, ti); the diagonal is finished by the caller
    L.jp[J] = ti; L.jp[NB + J] = tr;
    L.jq[J] = -tr; L.jq[NB + J] = ti;
    eval_col<J + 1>(L, vr, vi, sr, si, r);
  }
}

// ---- Gauss-Jordan step for pivot (bus K, part P): P = 0 the angle unknown / P row, P = 1 the magnitude unknown / Q row --------------
template <int K, int P, int C> __device__ __forceinline__ void sweep_cols(Lane& L, double fp, double fq) {
  // columns still alive: angle_{K+1..13}, magnitude_{K+P..13}  -> indices C over [0, 28)
  if constexpr (C < 2 * NB) {
    constexpr bool alive = (C < NB) ? (C > K) : (C - NB >= K + P);
    if constexpr (alive) {
      if constexpr (P == 0) { fmac_bcast<K>(L.jq[C], L.jp[C], fq); fmac_bcast_self<K>(L.jp[C], fp); }
      else { fmac_bcast<K>(L.jp[C], L.jq[C], fp); fmac_bcast_self<K>(L.jq[C], fq); }
    }
    sweep_cols<K, P, C + 1>(L, fp, fq);
  }
}
template <int K, int P> __device__ __forceinline__ void gj_step(Lane& L, int r) {
  constexpr int C = K + NB * P;
  const double piv = bcast16<K>(P == 0 ? L.jp[C] : L.jq[C]);
  const double rinv = fast_rcp(piv);
  // the pivot row itself is not swept; every other row r' takes  row' -= (a(r', C) / piv) * pivot row
  const bool own = (r == K);
  double fp = -(L.jp[C] * rinv), fq = -(L.jq[C] * rinv);
  if (own) { if (P == 0) fp = 0.0; else fq = 0.0; }
  sweep_cols<K, P, 0>(L, fp, fq);
  if constexpr (P == 0) { fmac_bcast<K>(L.bq, L.bp, fq); fmac_bcast_self<K>(L.bp, fp); }
  else { fmac_bcast<K>(L.bp, L.bq, fp); fmac_bcast_self<K>(L.bq, fq); }
}
template <int K> __device__ __forceinline__ void gj_all(Lane& L, int r) {
  if constexpr (K < NB) {
    gj_step<K, 0>(L, r);
    gj_step<K, 1>(L, r);
    gj_all<K + 1>(L, r);
  }
}

__device__ __forceinline__ void newton_iteration(Lane& L, int r, double* norm_out) {
  double sn, cs;
  sincos(L.va, &sn, &cs);
  const double vr = L.vm * cs, vi = L.vm * sn;
  double sr = 0.0, si = 0.0;
  eval_col<0>(L, vr, vi, sr, si, r);
  // diagonal blocks: T_ii was written as an off-diagonal entry of column r; finish it (the main kernel's pass 2)
  double tii_i = 0.0, tii_r = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) if (j == r) { tii_i = L.jp[j]; tii_r = L.jp[NB + j]; }
  const int ni = L.nv;
  const double pm = sr - L.psp, qm = si - L.qsp;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j == r) {
      L.jp[j] = (ni >= 1) ? tii_i - si : 1.0;  L.jp[NB + j] = (ni == 2) ? (tii_r + sr) : 0.0;
      L.jq[j] = (ni == 2) ? sr - tii_r : 0.0;  L.jq[NB + j] = (ni == 2) ? (tii_i + si) : 1.0;
    } else {
      // rows of unknowns that do not exist are identity rows
      if (ni < 1) { L.jp[j] = 0.0; L.jp[NB + j] = 0.0; }
      if (ni < 2) { L.jq[j] = 0.0; L.jq[NB + j] = 0.0; }
    }
  }
  L.bp = (ni >= 1) ? -pm : 0.0;
  L.bq = (ni == 2) ? -qm : 0.0;
  *norm_out = fmax(fabs(L.bp), fabs(L.bq));
  gj_all<0>(L, r);
  // x = rhs / pivot (the pivots stayed on the diagonal)
  double dpiv = 1.0, qpiv = 1.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) if (j == r) { dpiv = L.jp[j]; qpiv = L.jq[NB + j]; }
  const double da = L.bp * fast_rcp(dpiv), dv = L.bq * fast_rcp(qpiv);
  L.va += da;
  L.vm = fma(L.vm, dv, L.vm);
}

__global__ void __launch_bounds__(64) packed_kernel(const double* yre, const double* yim, const double* psp, const double* qsp,
                                                    const int* nv, double* vm_out, double* va_out, double* norms, long long* cycles, int n_it) {
  const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
  const int sys = blockIdx.x * NSYS + g;
  Lane L;
  const bool live = r < NB;
  const int rc = live ? r : 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) { L.yre[j] = live ? yre[(sys * NB + rc) * NB + j] : 0.0; L.yim[j] = live ? yim[(sys * NB + rc) * NB + j] : 0.0; }
  L.psp = live ? psp[sys * NB + rc] : 0.0; L.qsp = live ? qsp[sys * NB + rc] : 0.0;
  L.nv = live ? nv[sys * NB + rc] : 0;
  L.vm = 1.0; L.va = 0.0;
  double nrm = 0.0;
  // (idle lanes 14, 15 of a row play a reference bus: identity rows, nothing changes)
  const long long t0 = clock64();
  for (int it = 0; it < n_it; ++it) newton_iteration(L, r, &nrm);
  const long long t1 = clock64();
  if (live) { vm_out[sys * NB + r] = L.vm; va_out[sys * NB + r] = L.va; norms[sys * NB + r] = nrm; }
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

// ---- host reference: the same Newton iteration with a dense Gauss-Jordan in the same pivot order -------------------------------------
static void host_newton(const double* yre, const double* yim, const double* psp, const double* qsp, const int* nv, double* vm, double* va, int n_it) {
  for (int it = 0; it < n_it; ++it) {
    double A[2 * NB][2 * NB + 1] = {{0}};
    double vr[NB], vi[NB];
    for (int i = 0; i < NB; ++i) { vr[i] = vm[i] * cos(va[i]); vi[i] = vm[i] * sin(va[i]); }
    for (int i = 0; i < NB; ++i) {
      double sr = 0, si = 0, tiir = 0, tiii = 0;
      for (int j = 0; j < NB; ++j) {
        const double ar = yre[i * NB + j] * vr[j] - yim[i * NB + j] * vi[j], ai = yre[i * NB + j] * vi[j] + yim[i * NB + j] * vr[j];
        const double tr = vr[i] * ar + vi[i] * ai, ti = vi[i] * ar - vr[i] * ai;
        sr += tr; si += ti;
        A[2 * i][j] = ti; A[2 * i][NB + j] = tr; A[2 * i + 1][j] = -tr; A[2 * i + 1][NB + j] = ti;
        if (j == i) { tiir = tr; tiii = ti; }
      }
      const int ni = nv[i];
      for (int j = 0; j < NB; ++j) if (j != i) {
        if (ni < 1) { A[2 * i][j] = 0; A[2 * i][NB + j] = 0; }
        if (ni < 2) { A[2 * i + 1][j] = 0; A[2 * i + 1][NB + j] = 0; }
      }
      A[2 * i][i] = ni >= 1 ? tiii - si : 1.0; A[2 * i][NB + i] = ni == 2 ? tiir + sr : 0.0;
      A[2 * i + 1][i] = ni == 2 ? sr - tiir : 0.0; A[2 * i + 1][NB + i] = ni == 2 ? tiii + si : 1.0;
      A[2 * i][2 * NB] = ni >= 1 ? -(sr - psp[i]) : 0.0;
      A[2 * i + 1][2 * NB] = ni == 2 ? -(si - qsp[i]) : 0.0;
    }
    for (int k = 0; k < NB; ++k) for (int p = 0; p < 2; ++p) {
      const int row = 2 * k + p, col = k + NB * p;
      const double rinv = 1.0 / A[row][col];
      for (int r2 = 0; r2 < 2 * NB; ++r2) if (r2 != row) {
        const double f = -A[r2][col] * rinv;
        for (int c = 0; c <= 2 * NB; ++c) if (c != col) A[r2][c] += f * A[row][c];
        A[r2][col] = 0.0;
      }
    }
    for (int i = 0; i < NB; ++i) {
      va[i] += A[2 * i][2 * NB] / A[2 * i][i];
      vm[i] += vm[i] * (A[2 * i + 1][2 * NB] / A[2 * i + 1][NB + i]);
    }
  }
}

int main(int argc, char** argv) {
  const int n_blocks = argc > 1 ? atoi(argv[1]) : 1, n_it = argc > 2 ? atoi(argv[2]) : 4;
  const int nsys = n_blocks * NSYS;
  std::vector<double> yre(nsys * NB * NB, 0.0), yim(nsys * NB * NB, 0.0), psp(nsys * NB), qsp(nsys * NB);
  std::vector<int> nv(nsys * NB);
  srand(7);
  auto rnd = []() { return rand() / (double)RAND_MAX; };
  for (int s = 0; s < nsys; ++s) {
    double* R = &yre[s * NB * NB]; double* I = &yim[s * NB * NB];
    auto line = [&](int a, int b) {
      const double rr = 0.01 + 0.04 * rnd(), x = 0.05 + 0.2 * rnd(), den = rr * rr + x * x, g = rr / den, bb = -x / den, ch = 0.02 * rnd();
      R[a * NB + a] += g; I[a * NB + a] += bb + ch; R[b * NB + b] += g; I[b * NB + b] += bb + ch;
      R[a * NB + b] -= g; I[a * NB + b] -= bb; R[b * NB + a] -= g; I[b * NB + a] -= bb;
    };
    for (int i = 0; i < NB; ++i) line(i, (i + 1) % NB);          // a ring and six chords: 20 lines, as IEEE-14
    for (int c = 0; c < 6; ++c) { const int a = rand() % NB, b = (a + 2 + rand() % (NB - 4)) % NB; if (a != b) line(a, b); }
    for (int i = 0; i < NB; ++i) {
      nv[s * NB + i] = i == 0 ? 0 : (i < 3 ? 1 : 2);
      psp[s * NB + i] = (i < 3 ? 0.4 : -0.15) * (0.5 + rnd());
      qsp[s * NB + i] = i < 3 ? 0.0 : -0.05 * (0.5 + rnd());
    }
  }
  double *d_yre, *d_yim, *d_psp, *d_qsp, *d_vm, *d_va, *d_nrm; int* d_nv; long long* d_cyc;
  hipMalloc(&d_yre, yre.size() * 8); hipMalloc(&d_yim, yim.size() * 8); hipMalloc(&d_psp, psp.size() * 8); hipMalloc(&d_qsp, qsp.size() * 8);
  hipMalloc(&d_vm, nsys * NB * 8); hipMalloc(&d_va, nsys * NB * 8); hipMalloc(&d_nrm, nsys * NB * 8); hipMalloc(&d_nv, nv.size() * 4);
  hipMalloc(&d_cyc, n_blocks * 8);
  hipMemcpy(d_yre, yre.data(), yre.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_yim, yim.data(), yim.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_psp, psp.data(), psp.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_qsp, qsp.data(), qsp.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_nv, nv.data(), nv.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(packed_kernel, dim3(n_blocks), dim3(64), 0, 0, d_yre, d_yim, d_psp, d_qsp, d_nv, d_vm, d_va, d_nrm, d_cyc, n_it);
  hipDeviceSynchronize();
  std::vector<double> vm(nsys * NB), va(nsys * NB), nrm(nsys * NB);
  std::vector<long long> cyc(n_blocks);
  hipMemcpy(vm.data(), d_vm, vm.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(va.data(), d_va, va.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(nrm.data(), d_nrm, nrm.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0.0, worst_nrm = 0.0;
  for (int s = 0; s < nsys; ++s) {
    double hvm[NB], hva[NB];
    for (int i = 0; i < NB; ++i) { hvm[i] = 1.0; hva[i] = 0.0; }
    host_newton(&yre[s * NB * NB], &yim[s * NB * NB], &psp[s * NB], &qsp[s * NB], &nv[s * NB], hvm, hva, n_it);
    for (int i = 0; i < NB; ++i) { worst = fmax(worst, fmax(fabs(hvm[i] - vm[s * NB + i]), fabs(hva[i] - va[s * NB + i]))); worst_nrm = fmax(worst_nrm, nrm[s * NB + i]); }
  }
  double mean = 0; for (long long c : cyc) mean += (double)c; mean /= n_blocks;
  printf("packed 14-bus Newton: %d wavefront(s) x 4 systems, %d iterations: %.0f cycles per wavefront-iteration = %.0f per system-iteration "
         "(one-word step kernel: ~11000 per environment-iteration; break-even 2750)\n", n_blocks, n_it, mean / n_it, mean / n_it / 4.0);
  printf("max |V - host Newton| after %d iterations: %.3e; largest mismatch entering the last iteration: %.3e\n", n_it, worst, worst_nrm);
  return worst < 1e-9 ? 0 : 1;
}
