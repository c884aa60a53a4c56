/*
 * ORACLE (test infrastructure, not product code): plain-C, scalar, single-threaded-per-environment
 * restatement of pypownet's load-flow hot path, used (a) as a second, independent checker of the HIP
 * engine at sizes where the numpy/scipy oracle is too slow and (b) as the "port" CPU baseline of bench.py.
 *
 * What it restates (reference file:line; PYPOWER 5.1.4 is the un-vendored dependency of requirements.txt:9,
 * its algorithm is restated from the published MATPOWER/PYPOWER sources):
 *   orc_solve()            pypower runpf/rundcpf as called at pypownet/grid.py:227-229:
 *                          ext2int, bustypes, makeYbus, makeSbus, newtonpf | makeB+fdpf(XB) | makeBdc+dcpf,
 *                          pfsoln, int2ext; bus types as Grid._synchronize_bus_types (grid.py:141-209)
 *   orc_advance()          Game.load_entries_from_next_timestep/_from_timestep_id (game.py:405-501),
 *                          Grid.load_timestep_injections incl. quirk q9 (grid.py:266-311)
 *   orc_cascade()          Game._compute_loadflow_cascading (game.py:503-589)
 *   orc_apply_action()     Game._verify_illegal_action/apply_action + repair (game.py:591-753, 809-854)
 *   orc_step()/orc_process_game_over()/orc_reset()   game.py:799-885, 762-797, 255-340
 *
 * It deliberately shares NO code and no data structures with pypownet_amd/csrc: buses are numbered like
 * PYPOWER (kept rows in row order), matrices are CSR rows built by insertion, the unknowns are scalar
 * (theta_i, V_i) and the LU is a row-wise sparse elimination with a dense accumulator (no bitsets, no
 * blocks).  It is pinned against oracle/pf_np.py (itself pinned by the reference's known answers K1-K9) in
 * tests/test_oracle_c.py.
 *
 * It exports the same entry points as include/ppn.h under the prefix orc_ so that the test-suite can drive
 * both implementations with one harness.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/ppn.h"

typedef double complex cplx;

#define MAXDEG 48

typedef struct {
  int n;            /* dimension */
  int* cnt;         /* entries per row */
  int* col;         /* [n * MAXDEG] */
  double* val;      /* [n * MAXDEG] */
} RowMat;           /* real sparse matrix, rows built by insertion */

typedef struct {
  int nS, nP, nL, nl, nrows, alen, ntopo;
  double baseMVA;
  double *gs, *bs, *kv, *vm0, *va0;
  int *gen_sub, *load_sub, *or_sub, *ex_sub, *sub_load, *elem_sub, *sub_pos;
  double *qmax, *qmin, *qg0;
  double *r, *x, *b, *tap, *shift;
  uint8_t* status0;
  int slack_row;
  double* limits;
  ppn_rules R;
  /* chronics */
  int n_slots;
  int *cT;
  float **pp, **pv, **lp, **lq, **ppp, **pvp, **lpp, **lqp, **mt, **hz;
  int **ids;
} OCase;

typedef struct {
  double *vm, *va, *pg, *qg, *vg, *pd, *qd, *pf, *qf, *pt, *qt, *amps;
  uint8_t *pn, *ln, *on, *en, *st, *btype, *lev;
  int *rec, *lcd, *ncd, *soft;
  int id, src;      /* index of the environment (seed of its chronic draws), outcome of the last solve of the step's cascade */
  unsigned draws;   /* chronics drawn so far (PPN_LOOP_RANDOM) */
  int done, dead, succ, flag, ill, depth, nsolve, niter, slot, row, nlc, npc, epoch;
  int nstep;            /* Game.step calls executed since orc_reset (PPN_F_N_STEPS) */
  int illn[3], actsw[2];   /* IllegalActionException contents as counts; node / line switches of the action after the step */
  double min_vm;    /* test diagnostic: smallest |V| of an active bus over the solves of the last step (the last
                       iterate of a failed one included: a solve that reaches V = 0 exactly fails on the NaN it produces) */
} OEnv;

struct orc_engine {
  OCase c;
  int batch;
  OEnv* env;
  OEnv* sim;
  char err[256];
};
typedef struct orc_engine orc_engine;


/* Per-thread bump arena for the per-solve scratch (malloc/calloc of ~1 MB per solve would dominate the run). */
#define ARENA_BYTES ((size_t)24 << 20)
static __thread char* t_arena = NULL;
static __thread size_t t_off = 0;
static void* a_alloc(size_t bytes, int zero) {
  if (!t_arena) t_arena = (char*)malloc(ARENA_BYTES);
  bytes = (bytes + 63) & ~(size_t)63;
  if (t_off + bytes > ARENA_BYTES) { fprintf(stderr, "oracle arena exhausted\n"); abort(); }
  void* p = t_arena + t_off;
  t_off += bytes;
  if (zero) memset(p, 0, bytes);
  return p;
}
#define malloc(n) a_alloc((n), 0)
#define calloc(n, m) a_alloc((size_t)(n) * (size_t)(m), 1)
#define free(p) ((void)(p))

/* ------------------------------------------------------------------------------------------------------ */
static void rm_init(RowMat* m, int n) {
  m->n = n;
  m->cnt = (int*)calloc(n, sizeof(int));
  m->col = (int*)malloc((size_t)n * MAXDEG * sizeof(int));
  m->val = (double*)malloc((size_t)n * MAXDEG * sizeof(double));
}
static void rm_free(RowMat* m) { free(m->cnt); free(m->col); free(m->val); }
static int rm_add(RowMat* m, int i, int j, double v) {
  int* c = m->col + (size_t)i * MAXDEG;
  double* x = m->val + (size_t)i * MAXDEG;
  for (int k = 0; k < m->cnt[i]; ++k) if (c[k] == j) { x[k] += v; return 0; }
  if (m->cnt[i] >= MAXDEG) return -1;
  c[m->cnt[i]] = j; x[m->cnt[i]] = v; m->cnt[i]++;
  return 0;
}

/* Sparse LU, row-wise with a dense accumulator, no pivoting.  Rows are eliminated in the order perm[0..n).
 * On success L (unit lower, by rows, columns in elimination positions) and U are stored in lu / ucol.  */
typedef struct {
  int n;
  int *lcnt, *lcol; double* lval;    /* strictly lower part of each row (in elimination positions) */
  int *ucnt, *ucol; double* uval;    /* upper part incl. diagonal first */
  int* perm; int* pos;
} SpLU;
#define LUDEG 96

static void lu_init(SpLU* f, int n) {
  f->n = n;
  f->lcnt = (int*)calloc(n, sizeof(int)); f->ucnt = (int*)calloc(n, sizeof(int));
  f->lcol = (int*)malloc((size_t)n * LUDEG * sizeof(int)); f->ucol = (int*)malloc((size_t)n * LUDEG * sizeof(int));
  f->lval = (double*)malloc((size_t)n * LUDEG * sizeof(double)); f->uval = (double*)malloc((size_t)n * LUDEG * sizeof(double));
  f->perm = (int*)malloc(n * sizeof(int)); f->pos = (int*)malloc(n * sizeof(int));
}
static void lu_free(SpLU* f) { free(f->lcnt); free(f->ucnt); free(f->lcol); free(f->ucol); free(f->lval); free(f->uval); free(f->perm); free(f->pos); }

/* returns 0 ok, -1 capacity */
static int lu_factor(SpLU* f, const RowMat* a, const int* perm) {
  const int n = a->n;
  double* w = (double*)calloc(n, sizeof(double));
  char* mark = (char*)calloc(n, 1);
  int* cols = (int*)malloc(n * sizeof(int));
  int rc = 0;
  for (int p = 0; p < n; ++p) { f->perm[p] = perm[p]; f->pos[perm[p]] = p; }
  for (int p = 0; p < n && !rc; ++p) {
    const int i = perm[p];
    int m = 0;
    for (int k = 0; k < a->cnt[i]; ++k) {
      const int q = f->pos[a->col[(size_t)i * MAXDEG + k]];
      w[q] += a->val[(size_t)i * MAXDEG + k];
      if (!mark[q]) { mark[q] = 1; int t = m++; while (t > 0 && cols[t - 1] > q) { cols[t] = cols[t - 1]; --t; } cols[t] = q; }
    }
    int s = 0;
    while (s < m && cols[s] < p) {
      const int k = cols[s];
      const double lik = w[k] / f->uval[(size_t)k * LUDEG];
      w[k] = lik;
      for (int u = 1; u < f->ucnt[k]; ++u) {
        const int j = f->ucol[(size_t)k * LUDEG + u];
        if (!mark[j]) { mark[j] = 1; int t = m++; while (t > s + 1 && cols[t - 1] > j) { cols[t] = cols[t - 1]; --t; } cols[t] = j; }
        w[j] -= lik * f->uval[(size_t)k * LUDEG + u];
      }
      ++s;
    }
    if (s > LUDEG || m - s > LUDEG || m - s < 1 || cols[s] != p) { rc = -1; }
    if (!rc) {
      f->lcnt[p] = s;
      for (int t = 0; t < s; ++t) { f->lcol[(size_t)p * LUDEG + t] = cols[t]; f->lval[(size_t)p * LUDEG + t] = w[cols[t]]; }
      f->ucnt[p] = m - s;
      for (int t = s; t < m; ++t) { f->ucol[(size_t)p * LUDEG + t - s] = cols[t]; f->uval[(size_t)p * LUDEG + t - s] = w[cols[t]]; }
    }
    for (int t = 0; t < m; ++t) { w[cols[t]] = 0.0; mark[cols[t]] = 0; }
  }
  free(w); free(mark); free(cols);
  return rc;
}

/* x := A^-1 b  (b, x indexed by ORIGINAL row/col index) */
static void lu_solve(const SpLU* f, const double* b, double* x) {
  const int n = f->n;
  double* y = (double*)malloc(n * sizeof(double));
  for (int p = 0; p < n; ++p) {
    double acc = b[f->perm[p]];
    for (int t = 0; t < f->lcnt[p]; ++t) acc -= f->lval[(size_t)p * LUDEG + t] * y[f->lcol[(size_t)p * LUDEG + t]];
    y[p] = acc;
  }
  for (int p = n - 1; p >= 0; --p) {
    double acc = y[p];
    for (int t = 1; t < f->ucnt[p]; ++t) acc -= f->uval[(size_t)p * LUDEG + t] * y[f->ucol[(size_t)p * LUDEG + t]];
    y[p] = acc / f->uval[(size_t)p * LUDEG];
  }
  for (int p = 0; p < n; ++p) x[f->perm[p]] = y[p];
  free(y);
}

/* ------------------------------------------------------------------------------------------------------ */
/* Diagnostic for DESIGN.md ("is there a PROVABLE early exit from a Newton solve that will not converge?"): over the
 * Newton solves that run out of their iterations, how many reach a state from which the remaining iterations are decided --
 * an update that is exactly zero (a fixed point with a mismatch above the tolerance), or an iterate that repeats an earlier one bit
 * for bit (a cycle).  Counted only when ORC_CYCLE_STATS is set in the environment; read with orc_debug_cycle_stats. */
static long long g_cyc[6];      /* Newton solves, ran out of iterations, of those: zero update seen, exact repeat seen, NaN/inf mismatch seen, at iteration (sum) */
static int g_cyc_on = -1;
/* One runpf.  Returns 0 converged, 1 diverged, 2 not connexe (exception path of grid.py:228-231).           */
typedef struct { int n; int* deg; int* col; cplx* val; } YMat;

static int y_add(YMat* y, int i, int j, cplx v) {
  for (int k = 0; k < y->deg[i]; ++k) if (y->col[(size_t)i * MAXDEG + k] == j) { y->val[(size_t)i * MAXDEG + k] += v; return 0; }
  if (y->deg[i] >= MAXDEG) return -1;
  y->col[(size_t)i * MAXDEG + y->deg[i]] = j; y->val[(size_t)i * MAXDEG + y->deg[i]] = v; y->deg[i]++;
  return 0;
}

static void branch_y(double r, double x, double b, double tap, double shift_deg, cplx* yff, cplx* yft, cplx* ytf, cplx* ytt) {
  const cplx ys = 1.0 / (r + I * x);
  const cplx t = ((tap != 0.0) ? tap : 1.0) * cexp(I * M_PI / 180.0 * shift_deg);
  *ytt = ys + I * b / 2.0;
  *yff = *ytt / (t * conj(t));
  *yft = -ys / conj(t);
  *ytf = -ys / t;
}

static int connected(int n, const YMat* y) {
  int* stack = (int*)malloc(n * sizeof(int));
  char* seen = (char*)calloc(n, 1);
  int top = 0, cnt = 0;
  stack[top++] = 0; seen[0] = 1;
  while (top) {
    const int i = stack[--top]; ++cnt;
    for (int k = 0; k < y->deg[i]; ++k) { const int j = y->col[(size_t)i * MAXDEG + k]; if (!seen[j]) { seen[j] = 1; stack[top++] = j; } }
  }
  free(stack); free(seen);
  return cnt == n;
}

static int orc_solve(const OCase* c, OEnv* e, int* iters) {
  const int nS = c->nS, nrows = c->nrows, nl = c->nl, nP = c->nP, nL = c->nL;
  const ppn_rules* R = &c->R;
  *iters = 0;
  t_off = 0;   /* release the previous solve's scratch */
  /* isolated rows (grid.py:197-204) */
  char* touched = (char*)calloc(nrows, 1);
  for (int l = 0; l < nl; ++l) if (e->st[l]) { touched[c->or_sub[l] + e->on[l] * nS] = 1; touched[c->ex_sub[l] + e->en[l] * nS] = 1; }
  /* bus types (grid.py:141-174) */
  char* hasgen = (char*)calloc(nrows, 1);
  char* genon = (char*)calloc(nrows, 1);
  for (int g = 0; g < nP; ++g) { const int r = c->gen_sub[g] + e->pn[g] * nS; hasgen[r] = 1; if (e->vg[g] > 0.0) genon[r] = 1; }
  int target = c->slack_row;
  if (!touched[target]) {
    target = -1;
    for (int g = 0; g < nP; ++g) { const int r = c->gen_sub[g] + e->pn[g] * nS; if (r != c->slack_row) { target = r; break; } }
  }
  for (int r = 0; r < nrows; ++r)
    e->btype[r] = (uint8_t)(!touched[r] ? 4 : (hasgen[r] ? (r == target ? 3 : 2) : 1));
  /* ext2int: kept rows in row order */
  int* e2i = (int*)malloc(nrows * sizeof(int));
  int* i2e = (int*)malloc(nrows * sizeof(int));
  int n = 0;
  for (int r = 0; r < nrows; ++r) { if (touched[r]) { e2i[r] = n; i2e[n++] = r; } else e2i[r] = -1; }
  int rc = 2, ref = -1;
  int *typ = NULL, *perm = NULL;
  double *vm = NULL, *va = NULL, *psp = NULL, *qsp = NULL;
  YMat Y = {0, NULL, NULL, NULL};
  if (n == 0) goto done;
  /* bustypes: ref / pv / pq  (typ: 0 ref, 1 pv, 2 pq) */
  typ = (int*)malloc(n * sizeof(int));
  for (int i = 0; i < n; ++i) {
    const int r = i2e[i];
    const int on = genon[r];
    if (e->btype[r] == 3 && on) { typ[i] = 0; ref = i; }
    else if (e->btype[r] == 2 && on) typ[i] = 1;
    else typ[i] = 2;
  }
  if (ref < 0) {
    for (int i = 0; i < n; ++i) if (typ[i] == 1) { ref = i; typ[i] = 0; break; }
    if (ref < 0) goto done;                 /* IndexError in bustypes -> "not connexe" */
  }
  vm = (double*)malloc(n * sizeof(double)); va = (double*)malloc(n * sizeof(double));
  psp = (double*)calloc(n, sizeof(double)); qsp = (double*)calloc(n, sizeof(double));
  for (int i = 0; i < n; ++i) { vm[i] = e->vm[i2e[i]]; va[i] = e->va[i2e[i]] * (M_PI / 180.0); }
  int n_gen_on = 0;
  for (int g = 0; g < nP; ++g) {
    const int i = e2i[c->gen_sub[g] + e->pn[g] * nS];
    if (i >= 0 && e->vg[g] > 0.0) { psp[i] += e->pg[g] / c->baseMVA; ++n_gen_on; if (typ[i] != 2) vm[i] = e->vg[g]; }
  }
  for (int q = 0; q < nL; ++q) {
    const int i = e2i[c->load_sub[q] + e->ln[q] * nS];
    if (i >= 0) { psp[i] -= e->pd[q] / c->baseMVA; qsp[i] -= e->qd[q] / c->baseMVA; }
  }
  /* elimination order of the kept buses: static substation order, twins adjacent */
  perm = (int*)malloc(n * sizeof(int));
  {
    int k = 0;
    for (int p = 0; p < nS; ++p) {
      const int s = c->sub_pos[p];
      if (touched[s]) perm[k++] = e2i[s];
      if (touched[s + nS]) perm[k++] = e2i[s + nS];
    }
  }
  /* Ybus */
  Y.n = n; Y.deg = (int*)calloc(n, sizeof(int)); Y.col = (int*)malloc((size_t)n * MAXDEG * sizeof(int));
  Y.val = (cplx*)malloc((size_t)n * MAXDEG * sizeof(cplx));
  for (int i = 0; i < n; ++i) y_add(&Y, i, i, (c->gs[i2e[i]] + I * c->bs[i2e[i]]) / c->baseMVA);
  for (int l = 0; l < nl; ++l) if (e->st[l]) {
    const int f = e2i[c->or_sub[l] + e->on[l] * nS], t = e2i[c->ex_sub[l] + e->en[l] * nS];
    cplx yff, yft, ytf, ytt;
    branch_y(c->r[l], c->x[l], c->b[l], c->tap[l], c->shift[l], &yff, &yft, &ytf, &ytt);
    if (y_add(&Y, f, f, yff) | y_add(&Y, f, t, yft) | y_add(&Y, t, f, ytf) | y_add(&Y, t, t, ytt)) { rc = 4; goto done; }
  }
  if (!connected(n, &Y)) goto done;          /* singular B' / J / B: the reference ends in "not connexe"/outage */

  int success = 0;
  double slack_dp = 0.0;
  if (R->mode == PPN_MODE_DC) {
    /* makeBdc + dcpf */
    RowMat B; rm_init(&B, n);
    double* pbus = (double*)malloc(n * sizeof(double));
    for (int i = 0; i < n; ++i) pbus[i] = psp[i] - c->gs[i2e[i]] / c->baseMVA;
    for (int l = 0; l < nl; ++l) if (e->st[l]) {
      const int f = e2i[c->or_sub[l] + e->on[l] * nS], t = e2i[c->ex_sub[l] + e->en[l] * nS];
      const double b = 1.0 / c->x[l] / ((c->tap[l] != 0.0) ? c->tap[l] : 1.0);
      const double pfinj = -b * c->shift[l] * M_PI / 180.0;
      rm_add(&B, f, f, b); rm_add(&B, f, t, -b); rm_add(&B, t, f, -b); rm_add(&B, t, t, b);
      pbus[f] -= pfinj; pbus[t] += pfinj;
    }
    /* reduced system over non-ref buses */
    int* idx = (int*)malloc(n * sizeof(int)); int m = 0;
    for (int i = 0; i < n; ++i) idx[i] = (i == ref) ? -1 : m++;
    RowMat A; rm_init(&A, m);
    double* rhs = (double*)calloc(m, sizeof(double)); double* sol = (double*)calloc(m, sizeof(double));
    for (int i = 0; i < n; ++i) if (i != ref) {
      rhs[idx[i]] = pbus[i];
      for (int k = 0; k < B.cnt[i]; ++k) {
        const int j = B.col[(size_t)i * MAXDEG + k]; const double v = B.val[(size_t)i * MAXDEG + k];
        if (j == ref) rhs[idx[i]] -= v * va[ref]; else rm_add(&A, idx[i], idx[j], v);
      }
    }
    int* pm = (int*)malloc(m * sizeof(int)); int k2 = 0;
    for (int p = 0; p < n; ++p) if (perm[p] != ref) pm[k2++] = idx[perm[p]];
    SpLU F; lu_init(&F, m);
    if (lu_factor(&F, &A, pm)) rc = 4; else {
      lu_solve(&F, rhs, sol);
      for (int i = 0; i < n; ++i) { if (i != ref) va[i] = sol[idx[i]]; vm[i] = 1.0; }
      double acc = 0.0;
      for (int k = 0; k < B.cnt[ref]; ++k) acc += B.val[(size_t)ref * MAXDEG + k] * va[B.col[(size_t)ref * MAXDEG + k]];
      slack_dp = (acc - pbus[ref]) * c->baseMVA;
      success = 1; *iters = 1; rc = 0;
    }
    lu_free(&F); free(pm); rm_free(&A); rm_free(&B); free(pbus); free(idx); free(rhs); free(sol);
    if (rc) goto done;
  } else if (R->solver == PPN_SOLVER_NEWTON) {
    /* newtonpf: unknowns theta (non-ref) and Vm (pq), numbered bus-interleaved in elimination order */
    int* ith = (int*)malloc(n * sizeof(int)); int* ivm = (int*)malloc(n * sizeof(int)); int m = 0;
    int* pm = (int*)malloc(2 * n * sizeof(int));
    for (int p = 0; p < n; ++p) {
      const int i = perm[p];
      ith[i] = (typ[i] != 0) ? m++ : -1;
      ivm[i] = (typ[i] == 2) ? m++ : -1;
    }
    for (int q = 0; q < m; ++q) pm[q] = q;
    cplx* V = (cplx*)malloc(n * sizeof(cplx)); cplx* Ib = (cplx*)malloc(n * sizeof(cplx));
    double* F = (double*)malloc((m > 0 ? m : 1) * sizeof(double)); double* dx = (double*)malloc((m > 0 ? m : 1) * sizeof(double));
    if (g_cyc_on < 0) g_cyc_on = getenv("ORC_CYCLE_STATS") != NULL;
    double* hist = g_cyc_on ? (double*)malloc((size_t)(R->max_it + 2) * 2 * n * sizeof(double)) : NULL;
    int cyc_zero = 0, cyc_rep = 0, cyc_nan = 0, cyc_at = 0;
    for (int it = 0;; ++it) {
      if (hist && it <= R->max_it) {
        memcpy(hist + (size_t)it * 2 * n, vm, n * sizeof(double)); memcpy(hist + (size_t)it * 2 * n + n, va, n * sizeof(double));
        for (int j = 0; j < it && !cyc_rep; ++j) if (!memcmp(hist + (size_t)j * 2 * n, hist + (size_t)it * 2 * n, 2 * n * sizeof(double))) { cyc_rep = 1; if (!cyc_at) cyc_at = it; }
      }
      for (int i = 0; i < n; ++i) V[i] = vm[i] * cexp(I * va[i]);
      double normF = 0.0;
      for (int i = 0; i < n; ++i) {
        cplx acc = 0;
        for (int k = 0; k < Y.deg[i]; ++k) acc += Y.val[(size_t)i * MAXDEG + k] * V[Y.col[(size_t)i * MAXDEG + k]];
        Ib[i] = acc;
        const cplx mis = V[i] * conj(acc) - (psp[i] + I * qsp[i]);
        if (ith[i] >= 0) { F[ith[i]] = creal(mis); const double a = fabs(creal(mis)); if (a > normF || a != a) if (normF == normF) normF = a; }
        if (ivm[i] >= 0) { F[ivm[i]] = cimag(mis); const double a = fabs(cimag(mis)); if (a > normF || a != a) if (normF == normF) normF = a; }
      }
      if (normF < R->tol) { success = 1; break; }
      if (hist && !(normF == normF && normF < 1e300) && !cyc_nan) { cyc_nan = 1; if (!cyc_at) cyc_at = it; }
      if (it >= R->max_it) break;
      ++*iters;
      /* dSbus_dV */
      RowMat J; rm_init(&J, m);
      int bad = 0;
      for (int i = 0; i < n; ++i) {
        for (int k = 0; k < Y.deg[i]; ++k) {
          const int j = Y.col[(size_t)i * MAXDEG + k];
          const cplx yij = Y.val[(size_t)i * MAXDEG + k];
          cplx dva, dvm;
          if (i == j) {
            dva = I * V[i] * conj(Ib[i] - yij * V[i]);
            dvm = V[i] * conj(yij * V[i] / cabs(V[i])) + conj(Ib[i]) * V[i] / cabs(V[i]);
          } else {
            dva = I * V[i] * conj(-yij * V[j]);
            dvm = V[i] * conj(yij * V[j] / cabs(V[j]));
          }
          if (ith[i] >= 0 && ith[j] >= 0) bad |= rm_add(&J, ith[i], ith[j], creal(dva));
          if (ith[i] >= 0 && ivm[j] >= 0) bad |= rm_add(&J, ith[i], ivm[j], creal(dvm));
          if (ivm[i] >= 0 && ith[j] >= 0) bad |= rm_add(&J, ivm[i], ith[j], cimag(dva));
          if (ivm[i] >= 0 && ivm[j] >= 0) bad |= rm_add(&J, ivm[i], ivm[j], cimag(dvm));
        }
      }
      SpLU LU; lu_init(&LU, m);
      if (bad || lu_factor(&LU, &J, pm)) { lu_free(&LU); rm_free(&J); rc = 4; break; }
      lu_solve(&LU, F, dx);
      lu_free(&LU); rm_free(&J);
      if (hist && !cyc_zero) { int z = 1; for (int q = 0; q < m; ++q) if (dx[q] != 0.0) { z = 0; break; } if (z) { cyc_zero = 1; if (!cyc_at) cyc_at = it; } }
      for (int i = 0; i < n; ++i) {
        if (ith[i] >= 0) va[i] -= dx[ith[i]];
        if (ivm[i] >= 0) vm[i] -= dx[ivm[i]];
        const cplx v = vm[i] * cexp(I * va[i]);
        vm[i] = cabs(v); va[i] = carg(v);
        if (vm[i] < e->min_vm) e->min_vm = vm[i];   /* test diagnostic (an iterate at V = 0 exactly turns the next one into NaN) */
      }
    }
    if (hist) {
#pragma omp critical
      { g_cyc[0]++; if (!success && rc != 4) { g_cyc[1]++; g_cyc[2] += cyc_zero; g_cyc[3] += cyc_rep; g_cyc[4] += cyc_nan; g_cyc[5] += cyc_at; } }
    }
    free(ith); free(ivm); free(pm); free(V); free(Ib); free(F); free(dx);
    if (rc == 4) goto done;
    rc = 0;
  } else {
    /* makeB (XB) + fdpf */
    int* ip = (int*)malloc(n * sizeof(int)); int* iq = (int*)malloc(n * sizeof(int)); int mp = 0, mq = 0;
    for (int i = 0; i < n; ++i) { ip[i] = (typ[i] != 0) ? mp++ : -1; iq[i] = (typ[i] == 2) ? mq++ : -1; }
    if (mq == 0) { free(ip); free(iq); goto done; }       /* ValueError: norm of an empty array */
    RowMat Bp, Bq; rm_init(&Bp, mp); rm_init(&Bq, mq);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (iq[i] >= 0) bad |= rm_add(&Bq, iq[i], iq[i], -c->bs[i2e[i]] / c->baseMVA);
    for (int l = 0; l < nl; ++l) if (e->st[l]) {
      const int f = e2i[c->or_sub[l] + e->on[l] * nS], t = e2i[c->ex_sub[l] + e->en[l] * nS];
      cplx yff, yft, ytf, ytt;
      branch_y(0.0, c->x[l], 0.0, 1.0, c->shift[l], &yff, &yft, &ytf, &ytt);     /* B': r=0, b=0, tap=1 */
      if (ip[f] >= 0) bad |= rm_add(&Bp, ip[f], ip[f], -cimag(yff));
      if (ip[t] >= 0) bad |= rm_add(&Bp, ip[t], ip[t], -cimag(ytt));
      if (ip[f] >= 0 && ip[t] >= 0) { bad |= rm_add(&Bp, ip[f], ip[t], -cimag(yft)); bad |= rm_add(&Bp, ip[t], ip[f], -cimag(ytf)); }
      branch_y(c->r[l], c->x[l], c->b[l], c->tap[l], 0.0, &yff, &yft, &ytf, &ytt); /* B'': shift = 0 */
      if (iq[f] >= 0) bad |= rm_add(&Bq, iq[f], iq[f], -cimag(yff));
      if (iq[t] >= 0) bad |= rm_add(&Bq, iq[t], iq[t], -cimag(ytt));
      if (iq[f] >= 0 && iq[t] >= 0) { bad |= rm_add(&Bq, iq[f], iq[t], -cimag(yft)); bad |= rm_add(&Bq, iq[t], iq[f], -cimag(ytf)); }
    }
    int* pmp = (int*)malloc((mp > 0 ? mp : 1) * sizeof(int)); int* pmq = (int*)malloc(mq * sizeof(int)); int a1 = 0, a2 = 0;
    for (int p = 0; p < n; ++p) { const int i = perm[p]; if (ip[i] >= 0) pmp[a1++] = ip[i]; if (iq[i] >= 0) pmq[a2++] = iq[i]; }
    SpLU Fp, Fq; lu_init(&Fp, mp); lu_init(&Fq, mq);
    if (bad || lu_factor(&Fp, &Bp, pmp) || lu_factor(&Fq, &Bq, pmq)) rc = 4;
    else {
      double* P = (double*)malloc((mp > 0 ? mp : 1) * sizeof(double)); double* Q = (double*)malloc(mq * sizeof(double));
      double* dp = (double*)malloc((mp > 0 ? mp : 1) * sizeof(double)); double* dq = (double*)malloc(mq * sizeof(double));
      int i_fd = 0, half = 0;
      for (;;) {
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) {
          cplx acc = 0;
          for (int k = 0; k < Y.deg[i]; ++k) { const int j = Y.col[(size_t)i * MAXDEG + k]; acc += Y.val[(size_t)i * MAXDEG + k] * (vm[j] * cexp(I * va[j])); }
          const cplx vi = vm[i] * cexp(I * va[i]);
          const cplx mis = (vi * conj(acc) - (psp[i] + I * qsp[i])) / vm[i];
          if (ip[i] >= 0) { P[ip[i]] = creal(mis); const double a = fabs(creal(mis)); if (a > nrm || a != a) if (nrm == nrm) nrm = a; }
          if (iq[i] >= 0) { Q[iq[i]] = cimag(mis); const double a = fabs(cimag(mis)); if (a > nrm || a != a) if (nrm == nrm) nrm = a; }
        }
        if (nrm < R->tol) { success = 1; break; }
        if (half == 0) { if (i_fd >= R->max_it) break; ++i_fd; }
        ++*iters;
        if (half == 0) { lu_solve(&Fp, P, dp); for (int i = 0; i < n; ++i) if (ip[i] >= 0) va[i] -= dp[ip[i]]; }
        else { lu_solve(&Fq, Q, dq); for (int i = 0; i < n; ++i) if (iq[i] >= 0) vm[i] -= dq[iq[i]]; }
        half ^= 1;
      }
      for (int i = 0; i < n; ++i) { const cplx v = vm[i] * cexp(I * va[i]); vm[i] = cabs(v); va[i] = carg(v); }
      free(P); free(Q); free(dp); free(dq);
      rc = 0;
    }
    lu_free(&Fp); lu_free(&Fq); free(pmp); free(pmq); rm_free(&Bp); rm_free(&Bq); free(ip); free(iq);
    if (rc) goto done;
  }

  /* pfsoln + int2ext */
  {
    int bad = 0;
    cplx* V = (cplx*)malloc(n * sizeof(cplx));
    for (int i = 0; i < n; ++i) {
      V[i] = vm[i] * cexp(I * va[i]);
      e->vm[i2e[i]] = vm[i]; e->va[i2e[i]] = va[i] * 180.0 / M_PI;
      if (vm[i] != vm[i] || va[i] != va[i] || vm[i] > 1e10 || e->va[i2e[i]] > 1e10) bad = 1;
    }
    for (int g = 0; g < nP; ++g) {
      const int row = c->gen_sub[g] + e->pn[g] * nS, i = e2i[row];
      if (i < 0 || !(e->vg[g] > 0.0)) { e->pg[g] = 0.0; e->qg[g] = 0.0; continue; }
      const int q = c->sub_load[row % nS];
      const int ld = (q >= 0 && e->ln[q] == row / nS);
      if (R->mode == PPN_MODE_DC) { if (i == ref) e->pg[g] += slack_dp; continue; }
      cplx acc = 0;
      for (int k = 0; k < Y.deg[i]; ++k) acc += Y.val[(size_t)i * MAXDEG + k] * V[Y.col[(size_t)i * MAXDEG + k]];
      const cplx S = V[i] * conj(acc);
      double qg = cimag(S) * c->baseMVA + (ld ? e->qd[q] : 0.0);
      if (n_gen_on > 1 && c->qmin[g] != c->qmax[g])
        qg = c->qmin[g] + ((qg - c->qmin[g]) / (c->qmax[g] - c->qmin[g] + 2.220446049250313e-16)) * (c->qmax[g] - c->qmin[g]);
      e->qg[g] = qg;
      if (i == ref) e->pg[g] = creal(S) * c->baseMVA + (ld ? e->pd[q] : 0.0);
    }
    for (int l = 0; l < nl; ++l) {
      double pf = 0, qf = 0, pt = 0, qt = 0, amp = 0;
      if (e->st[l]) {
        const int rf = c->or_sub[l] + e->on[l] * nS;
        const int f = e2i[rf], t = e2i[c->ex_sub[l] + e->en[l] * nS];
        if (R->mode == PPN_MODE_DC) {
          const double b = 1.0 / c->x[l] / ((c->tap[l] != 0.0) ? c->tap[l] : 1.0);
          pf = (b * (va[f] - va[t]) - b * c->shift[l] * M_PI / 180.0) * c->baseMVA; pt = -pf;
        } else {
          cplx yff, yft, ytf, ytt;
          branch_y(c->r[l], c->x[l], c->b[l], c->tap[l], c->shift[l], &yff, &yft, &ytf, &ytt);
          const cplx Sf = V[f] * conj(yff * V[f] + yft * V[t]) * c->baseMVA;
          const cplx St = V[t] * conj(ytf * V[f] + ytt * V[t]) * c->baseMVA;
          pf = creal(Sf); qf = cimag(Sf); pt = creal(St); qt = cimag(St);
        }
        amp = 1000.0 * sqrt(pf * pf + qf * qf) / (pow(3.0, 0.5) * (vm[f] * c->kv[rf]));
      }
      e->pf[l] = pf; e->qf[l] = qf; e->pt[l] = pt; e->qt[l] = qt; e->amps[l] = amp;
      if (pf != pf || qf != qf || pt != pt || qt != qt || pf > 1e10 || qf > 1e10 || pt > 1e10 || qt > 1e10) bad = 1;
    }
    free(V);
    rc = (success && !bad) ? 0 : 1;
    for (int i = 0; i < n; ++i) if (vm[i] < e->min_vm) e->min_vm = vm[i];   /* NaN compares false */
  }
done:
  free(touched); free(hasgen); free(genon); free(e2i); free(i2e); free(typ); free(perm);
  free(vm); free(va); free(psp); free(qsp);
  if (Y.deg) { free(Y.deg); free(Y.col); free(Y.val); }
  return rc;
}

#undef malloc
#undef calloc
#undef free

/* ------------------------------------------------------------------------------------------------------ */
static int flag_of(int rc) { return rc == 0 ? 0 : (rc == 4 ? 4 : 1); }

/* counter-based generator of the chronic draws: the function include/ppn.h specifies for PPN_LOOP_RANDOM */
static unsigned mix32(unsigned seed, unsigned env, unsigned draw) {
  unsigned h = seed * 0x9E3779B1u ^ (env + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (draw + 1u) * 0xC2B2AE35u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}
static int next_slot(const OCase* c, OEnv* e) {      /* chronic.py:283-291 */
  if (c->R.chronic_looping == PPN_LOOP_FIXED) return e->slot;
  if (c->R.chronic_looping == PPN_LOOP_RANDOM) return (int)(mix32((unsigned)c->R.rng_seed, (unsigned)e->id, e->draws++) % (unsigned)c->n_slots);
  return (e->slot + 1) % c->n_slots;
}

static void orc_advance(const OCase* c, OEnv* e, int sim, int rec_ev) {
  const int nP = c->nP, nL = c->nL, nl = c->nl, nS = c->nS;
  int slot = e->slot, row = e->row;
  const int cur_slot = slot, cur_row = row < 0 ? 0 : row;
  int T = c->cT[slot];
  if (row == T - 1 && !sim) {
    /* roll-over with quirk q2 (game.py:481-493) */
    const int next = next_slot(c, e);
    int i0 = 0; for (int k = 0; k < T; ++k) if (c->ids[slot][k] == 0) { i0 = k; break; }
    const int nid = c->ids[slot][(i0 + 1 < T) ? i0 + 1 : T - 1];
    int r = -1; for (int k = 0; k < c->cT[next]; ++k) if (c->ids[next][k] == nid) { r = k; break; }
    slot = next; T = c->cT[slot]; row = r < 0 ? ((1 < T) ? 1 : T - 1) : r;
  } else if (row < 0) row = 0;
  else row = (row + 1 < T) ? row + 1 : T - 1;
  const float* pp = sim ? c->ppp[cur_slot] + (size_t)cur_row * nP : c->pp[slot] + (size_t)row * nP;
  const float* pv = sim ? c->pvp[cur_slot] + (size_t)cur_row * nP : c->pv[slot] + (size_t)row * nP;
  const float* lp = sim ? c->lpp[cur_slot] + (size_t)cur_row * nL : c->lp[slot] + (size_t)row * nL;
  const float* lq = sim ? c->lqp[cur_slot] + (size_t)cur_row * nL : c->lq[slot] + (size_t)row * nL;
  /* normalize_prods_voltages (grid.py:266-271): divisors = baseKV of the production-hosting rows in ROW order */
  double* kvlist = (double*)malloc(nP * sizeof(double)); int k = 0;
  for (int node = 0; node < 2; ++node) for (int g = 0; g < nP; ++g) if (e->pn[g] == node) kvlist[k++] = c->kv[c->gen_sub[g] + node * nS];
  for (int g = 0; g < nP; ++g) { e->pg[g] = (double)pp[g]; e->vg[g] = (double)(pv[g] <= 0.0f ? 0.0f : pv[g]) / kvlist[g]; }
  free(kvlist);
  for (int q = 0; q < nL; ++q) { e->pd[q] = (double)lp[q]; e->qd[q] = (double)lq[q]; }
  if (!sim) {
    for (int l = 0; l < nl; ++l) { if (e->rec[l] > 0) e->rec[l]--; if (e->lcd[l] > 0) e->lcd[l]--; }
    for (int s = 0; s < nS; ++s) if (e->ncd[s] > 0) e->ncd[s]--;
  }
  const float* mt = c->mt[slot] + (size_t)row * nl;
  const float* hz = c->hz[slot] + (size_t)row * nl;
  for (int l = 0; l < nl; ++l) {
    if (mt[l] > 0.0f) { e->st[l] = 0; if ((int)mt[l] > e->rec[l]) e->rec[l] = (int)mt[l]; if (rec_ev) e->lev[l] |= PPN_EV_MAINTENANCE; }
    if (!sim && hz[l] > 0.0f) { e->st[l] = 0; if ((int)hz[l] > e->rec[l]) e->rec[l] = (int)hz[l]; if (rec_ev) e->lev[l] |= PPN_EV_HAZARD; }
  }
  e->slot = slot; e->row = row;
}

static int orc_cascade(const OCase* c, OEnv* e, int rec_ev) {
  const int nl = c->nl;
  char* over = (char*)calloc(nl, 1);
  int solves = 0, rc = 0;
  for (;;) {
    int its = 0;
    rc = orc_solve(c, e, &its);
    ++solves; e->niter += its;
    if (rc) break;
    int nover = 0, cut = 0;
    for (int l = 0; l < nl; ++l) { over[l] = e->amps[l] > c->limits[l]; nover += over[l]; }
    if (!nover) break;
    for (int l = 0; l < nl; ++l) if (over[l] && e->amps[l] > c->R.hard_overflow_coefficient * c->limits[l]) {
      e->st[l] = 0; e->rec[l] = c->R.n_timesteps_hard_overflow_is_broken; over[l] = 0; cut = 1;
      if (rec_ev) e->lev[l] |= PPN_EV_HARD_OVERFLOW;
    }
    for (int l = 0; l < nl; ++l) if (over[l] && (double)e->soft[l] >= c->R.n_timesteps_consecutive_soft_overflow_breaks) {
      e->st[l] = 0; e->rec[l] = c->R.n_timesteps_soft_overflow_is_broken; over[l] = 0; cut = 1;
      if (rec_ev) e->lev[l] |= PPN_EV_SOFT_OVERFLOW;
    }
    if (!cut) break;
  }
  if (!rc) for (int l = 0; l < nl; ++l) e->soft[l] = over[l] ? e->soft[l] + 1 : 0;
  e->nsolve += solves; e->succ = (rc == 0);
  if (rec_ev) { e->depth = solves - 1; e->src = rc; }
  free(over);
  return rc;
}

static int orc_cut_flags(const OCase* c, OEnv* e) {
  const int nS = c->nS;
  char* touched = (char*)calloc(c->nrows, 1);
  for (int l = 0; l < c->nl; ++l) if (e->st[l]) { touched[c->or_sub[l] + e->on[l] * nS] = 1; touched[c->ex_sub[l] + e->en[l] * nS] = 1; }
  int nlc = 0, npc = 0;
  for (int q = 0; q < c->nL; ++q) nlc += !touched[c->load_sub[q] + e->ln[q] * nS];
  for (int g = 0; g < c->nP; ++g) npc += !touched[c->gen_sub[g] + e->pn[g] * nS];
  free(touched);
  e->nlc = nlc; e->npc = npc;
  if (nlc > c->R.max_number_loads_game_over) return 2;
  if (npc > c->R.max_number_prods_game_over) return 3;
  return 0;
}

static int orc_apply_action(const OCase* c, OEnv* e, const uint8_t* action, int apply) {
  const int nP = c->nP, nL = c->nL, nl = c->nl, nS = c->nS, ntopo = c->ntopo;
  uint8_t* a = (uint8_t*)malloc(c->alen);
  char* chg = (char*)calloc(nS, 1);
  for (int k = 0; k < c->alen; ++k) a[k] = action[k] ? 1 : 0;
  for (int k = 0; k < ntopo; ++k) if (a[k]) chg[c->elem_sub[k]] = 1;
  int ns = 0, nln = 0, bits = 0;
  for (int s = 0; s < nS; ++s) ns += chg[s];
  for (int l = 0; l < nl; ++l) nln += a[ntopo + l];
  int nb = 0, nc = 0, nn = 0, swn = 0, swl = nln;       /* counts behind the flag; switches of the action as seen after the step */
  for (int k = 0; k < ntopo; ++k) swn += a[k];
  if (ns > c->R.max_number_actionned_substations || nln > c->R.max_number_actionned_lines || ns + nln > c->R.max_number_actionned_total) {
    bits = PPN_ILL_TOO_MANY;
    memset(a, 0, c->alen); memset(chg, 0, nS);
    swn = 0; swl = 0;                 /* Action.set_as_do_nothing edits the caller's object in place (game.py:813) */
  } else {
    for (int l = 0; l < nl; ++l) if (a[ntopo + l]) { if (e->rec[l] > 0) { bits |= PPN_ILL_BROKEN_LINE; ++nb; } if (e->lcd[l] > 0) { bits |= PPN_ILL_LINE_COOLDOWN; ++nc; } }
    for (int s = 0; s < nS; ++s) if (chg[s] && e->ncd[s] > 0) { bits |= PPN_ILL_NODE_COOLDOWN; ++nn; }
    if (bits && apply) {
      for (int l = 0; l < nl; ++l) if (e->rec[l] > 0 || e->lcd[l] > 0) a[ntopo + l] = 0;
      for (int k = 0; k < ntopo; ++k) { const int s = c->elem_sub[k]; if (chg[s] && e->ncd[s] > 0) a[k] = 0; }
      for (int s = 0; s < nS; ++s) if (e->ncd[s] > 0) chg[s] = 0;
      swn = 0; swl = 0;
      for (int k = 0; k < ntopo; ++k) swn += a[k];
      for (int l = 0; l < nl; ++l) swl += a[ntopo + l];
    }
  }
  if (apply) { e->illn[0] = nb; e->illn[1] = nc; e->illn[2] = nn; e->actsw[0] = swn; e->actsw[1] = swl; }
  if (apply) for (int l = 0; l < nl; ++l) e->lev[l] = a[ntopo + l] ? PPN_EV_SWITCHED : 0;     /* events of this step start here */
  if (apply) {
    for (int g = 0; g < nP; ++g) if (a[g]) e->pn[g] ^= 1;
    for (int q = 0; q < nL; ++q) if (a[nP + q]) e->ln[q] ^= 1;
    for (int l = 0; l < nl; ++l) {
      if (a[nP + nL + l]) e->on[l] ^= 1;
      if (a[nP + nL + nl + l]) e->en[l] ^= 1;
      if (a[ntopo + l]) { e->st[l] ^= 1; e->lcd[l] = c->R.n_timesteps_actionned_line_reactionable; }
    }
    for (int s = 0; s < nS; ++s) if (chg[s]) e->ncd[s] = c->R.n_timesteps_actionned_node_reactionable;
  }
  free(a); free(chg);
  return bits;
}

static void orc_reset_grid(const OCase* c, OEnv* e) {
  for (int l = 0; l < c->nl; ++l) { e->rec[l] = 0; e->lcd[l] = 0; e->on[l] = 0; e->en[l] = 0; e->st[l] = c->status0[l]; e->pf[l] = e->qf[l] = e->pt[l] = e->qt[l] = 0.0; }
  for (int s = 0; s < c->nS; ++s) e->ncd[s] = 0;
  memset(e->pn, 0, c->nP); memset(e->ln, 0, c->nL);
  for (int r = 0; r < c->nrows; ++r) { e->vm[r] = c->vm0[r]; e->va[r] = c->va0[r]; }
}

static void orc_step_env(const OCase* c, OEnv* e, const uint8_t* action, int sim) {
  e->min_vm = 1e300;
  if (e->dead) return;
  const int ill = orc_apply_action(c, e, action, 1);
  orc_advance(c, e, sim, 1);
  const int rc = orc_cascade(c, e, 1);
  int flag = flag_of(rc);
  if (!flag) flag = orc_cut_flags(c, e);
  e->flag = flag; e->ill = ill; e->done = flag != 0; e->dead = flag != 0;
  if (!sim) e->nstep++;
}

static void orc_game_over_env(const OCase* c, OEnv* e, int force) {
  if (!e->dead && !force) return;
  int rc = 0;
  for (int attempt = 0; attempt < 64; ++attempt) {
    orc_reset_grid(c, e);
    if (c->R.game_over_mode_hard) {
      const int slot = next_slot(c, e);
      int j0 = 0; for (int k = 0; k < c->cT[slot]; ++k) if (c->ids[slot][k] == 0) { j0 = k; break; }
      e->slot = slot; e->row = ((j0 + 1 < c->cT[slot]) ? j0 + 1 : c->cT[slot] - 1) - 1;
    }
    e->epoch++;
    orc_advance(c, e, 0, 0);
    rc = orc_cascade(c, e, 0);
    if (rc == 0 || rc == 4) break;
  }
  orc_cut_flags(c, e);
  /* dead = 3: 64 restarts in a row diverged as well (the engine's PPN_RESTART_ATTEMPTS, include/ppn.h -- the reference recurses
   * without a bound, game.py:776-780): the environment stays over and the next pass goes on trying */
  e->dead = rc == 0 ? 0 : (rc == 4 ? 1 : 3);
}

/* ------------------------------------------------------------------------------------------------------ */
static void env_alloc(const OCase* c, OEnv* e) {
  memset(e, 0, sizeof *e);
#define D(n) (double*)calloc((n) > 0 ? (n) : 1, sizeof(double))
#define U(n) (uint8_t*)calloc((n) > 0 ? (n) : 1, 1)
#define N(n) (int*)calloc((n) > 0 ? (n) : 1, sizeof(int))
  e->vm = D(c->nrows); e->va = D(c->nrows); e->pg = D(c->nP); e->qg = D(c->nP); e->vg = D(c->nP); e->pd = D(c->nL); e->qd = D(c->nL);
  e->pf = D(c->nl); e->qf = D(c->nl); e->pt = D(c->nl); e->qt = D(c->nl); e->amps = D(c->nl);
  e->pn = U(c->nP); e->ln = U(c->nL); e->on = U(c->nl); e->en = U(c->nl); e->st = U(c->nl); e->btype = U(c->nrows); e->lev = U(c->nl);
  e->rec = N(c->nl); e->lcd = N(c->nl); e->ncd = N(c->nS); e->soft = N(c->nl);
#undef D
#undef U
#undef N
}
static void env_copy(const OCase* c, OEnv* d, const OEnv* s) {
#define CP(f, n, t) memcpy(d->f, s->f, (size_t)(n) * sizeof(t));
  CP(vm, c->nrows, double) CP(va, c->nrows, double) CP(pg, c->nP, double) CP(qg, c->nP, double) CP(vg, c->nP, double)
  CP(pd, c->nL, double) CP(qd, c->nL, double) CP(pf, c->nl, double) CP(qf, c->nl, double) CP(pt, c->nl, double)
  CP(qt, c->nl, double) CP(amps, c->nl, double) CP(pn, c->nP, uint8_t) CP(ln, c->nL, uint8_t) CP(on, c->nl, uint8_t)
  CP(en, c->nl, uint8_t) CP(st, c->nl, uint8_t) CP(btype, c->nrows, uint8_t) CP(rec, c->nl, int) CP(lcd, c->nl, int)
  CP(ncd, c->nS, int) CP(soft, c->nl, int) CP(lev, c->nl, uint8_t)
#undef CP
  d->id = s->id; d->src = s->src; d->draws = s->draws;
  d->done = s->done; d->dead = s->dead; d->succ = s->succ; d->flag = s->flag; d->ill = s->ill; d->depth = s->depth;
  d->nsolve = s->nsolve; d->niter = s->niter; d->slot = s->slot; d->row = s->row; d->nlc = s->nlc; d->npc = s->npc; d->epoch = s->epoch;
}

static void min_degree(int nS, int nl, const int* f, const int* t, int* order) {
  char* adj = (char*)calloc((size_t)nS * nS, 1); char* gone = (char*)calloc(nS, 1);
  for (int l = 0; l < nl; ++l) if (f[l] != t[l]) { adj[f[l] * nS + t[l]] = 1; adj[t[l] * nS + f[l]] = 1; }
  for (int step = 0; step < nS; ++step) {
    int best = -1, bd = 1 << 30;
    for (int i = 0; i < nS; ++i) if (!gone[i]) { int d = 0; for (int j = 0; j < nS; ++j) if (!gone[j] && adj[i * nS + j]) ++d; if (d < bd) { bd = d; best = i; } }
    gone[best] = 1; order[step] = best;
    for (int a = 0; a < nS; ++a) if (!gone[a] && adj[best * nS + a]) for (int b = 0; b < nS; ++b) if (b != a && !gone[b] && adj[best * nS + b]) adj[a * nS + b] = 1;
  }
  free(adj); free(gone);
}

static char g_err[256];

int orc_create(const ppn_case* pc, const ppn_rules* r, int32_t batch, int32_t device, orc_engine** out) {
  (void)device;
  if (!pc || !r || !out || batch <= 0) { snprintf(g_err, sizeof g_err, "orc_create: bad arguments"); return PPN_E_INVALID; }
  orc_engine* E = (orc_engine*)calloc(1, sizeof *E);
  OCase* c = &E->c;
  const int nrows = pc->n_bus_rows, nS = nrows / 2, nP = pc->n_gen, nl = pc->n_branch;
  c->nS = nS; c->nP = nP; c->nl = nl; c->nrows = nrows; c->baseMVA = pc->base_mva; c->R = *r;
  if (c->R.max_it <= 0) c->R.max_it = (c->R.solver == PPN_SOLVER_NEWTON) ? 10 : 25;
  c->gs = (double*)malloc(nrows * 8); c->bs = (double*)malloc(nrows * 8); c->kv = (double*)malloc(nrows * 8);
  c->vm0 = (double*)malloc(nrows * 8); c->va0 = (double*)malloc(nrows * 8);
  int* loads = (int*)malloc(nrows * sizeof(int)); int nL = 0; c->slack_row = -1;
  for (int i = 0; i < nrows; ++i) {
    const double* b = pc->bus + (size_t)i * pc->bus_cols;
    c->gs[i] = b[4]; c->bs[i] = b[5]; c->kv[i] = b[9]; c->vm0[i] = b[7]; c->va0[i] = b[8];
    if (b[2] != 0.0 || b[3] != 0.0) loads[nL++] = i;
    if (c->slack_row < 0 && (int)b[1] == 3) c->slack_row = i;
  }
  c->nL = nL; c->ntopo = nP + nL + 2 * nl; c->alen = c->ntopo + nl;
#define ROW_OF(id, dst) { dst = -1; for (int i_ = 0; i_ < nrows; ++i_) if ((long long)pc->bus[(size_t)i_ * pc->bus_cols] == (long long)(id)) { dst = i_; break; } }
  c->gen_sub = (int*)malloc((nP + 1) * sizeof(int)); c->load_sub = (int*)malloc((nL + 1) * sizeof(int));
  c->or_sub = (int*)malloc(nl * sizeof(int)); c->ex_sub = (int*)malloc(nl * sizeof(int));
  c->sub_load = (int*)malloc(nS * sizeof(int)); for (int s = 0; s < nS; ++s) c->sub_load[s] = -1;
  c->qmax = (double*)malloc((nP + 1) * 8); c->qmin = (double*)malloc((nP + 1) * 8); c->qg0 = (double*)malloc((nP + 1) * 8);
  for (int g = 0; g < nP; ++g) {
    const double* gr = pc->gen + (size_t)g * pc->gen_cols; int rr; ROW_OF(gr[0], rr);
    c->gen_sub[g] = rr; c->qmax[g] = gr[3]; c->qmin[g] = gr[4]; c->qg0[g] = gr[2];
  }
  for (int q = 0; q < nL; ++q) { c->load_sub[q] = loads[q]; c->sub_load[loads[q]] = q; }
  free(loads);
  c->r = (double*)malloc(nl * 8); c->x = (double*)malloc(nl * 8); c->b = (double*)malloc(nl * 8);
  c->tap = (double*)malloc(nl * 8); c->shift = (double*)malloc(nl * 8); c->status0 = (uint8_t*)malloc(nl);
  for (int l = 0; l < nl; ++l) {
    const double* b = pc->branch + (size_t)l * pc->branch_cols; int f, t; ROW_OF(b[0], f); ROW_OF(b[1], t);
    c->or_sub[l] = f; c->ex_sub[l] = t; c->r[l] = b[2]; c->x[l] = b[3]; c->b[l] = b[4]; c->tap[l] = b[8]; c->shift[l] = b[9];
    c->status0[l] = b[10] != 0.0;
  }
  c->elem_sub = (int*)malloc(c->ntopo * sizeof(int));
  { int k = 0; for (int g = 0; g < nP; ++g) c->elem_sub[k++] = c->gen_sub[g]; for (int q = 0; q < nL; ++q) c->elem_sub[k++] = c->load_sub[q];
    for (int l = 0; l < nl; ++l) c->elem_sub[k++] = c->or_sub[l]; for (int l = 0; l < nl; ++l) c->elem_sub[k++] = c->ex_sub[l]; }
  c->sub_pos = (int*)malloc(nS * sizeof(int));
  min_degree(nS, nl, c->or_sub, c->ex_sub, c->sub_pos);
  c->limits = (double*)malloc(nl * 8); for (int l = 0; l < nl; ++l) c->limits[l] = 1e30;
  E->batch = batch;
  E->env = (OEnv*)calloc(batch, sizeof(OEnv)); E->sim = (OEnv*)calloc(batch, sizeof(OEnv));
  for (int b = 0; b < batch; ++b) { env_alloc(c, &E->env[b]); env_alloc(c, &E->sim[b]); E->env[b].id = b; E->sim[b].id = b; }
  *out = E;
  return PPN_OK;
}

int orc_destroy(orc_engine* E) { (void)E; return PPN_OK; }   /* test helper: memory is reclaimed at exit */
const char* orc_last_error(const orc_engine* E) { return E ? E->err : g_err; }
const char* orc_version(void) { return "pypownet C oracle"; }

int orc_set_thermal_limits(orc_engine* E, const double* lim) { memcpy(E->c.limits, lim, E->c.nl * 8); return PPN_OK; }

int orc_load_chronic(orc_engine* E, int32_t slot, const ppn_chronic* ch) {
  OCase* c = &E->c;
  if (slot != c->n_slots) { snprintf(E->err, sizeof E->err, "chronic slots must be loaded in order"); return PPN_E_INVALID; }
  const int ns = c->n_slots + 1;
#define GROW(f, type) c->f = (type**)realloc(c->f, ns * sizeof(type*));
  GROW(pp, float) GROW(pv, float) GROW(lp, float) GROW(lq, float) GROW(ppp, float) GROW(pvp, float) GROW(lpp, float)
  GROW(lqp, float) GROW(mt, float) GROW(hz, float) GROW(ids, int)
#undef GROW
  c->cT = (int*)realloc(c->cT, ns * sizeof(int));
  c->cT[slot] = ch->T;
#define DUP(f, src, n) c->f[slot] = (float*)malloc((size_t)ch->T * (n) * 4 + 4); memcpy(c->f[slot], ch->src, (size_t)ch->T * (n) * 4);
  DUP(pp, prods_p, c->nP) DUP(pv, prods_v, c->nP) DUP(lp, loads_p, c->nL) DUP(lq, loads_q, c->nL)
  DUP(ppp, prods_p_planned, c->nP) DUP(pvp, prods_v_planned, c->nP) DUP(lpp, loads_p_planned, c->nL) DUP(lqp, loads_q_planned, c->nL)
  DUP(mt, maintenance, c->nl) DUP(hz, hazards, c->nl)
#undef DUP
  c->ids[slot] = (int*)malloc(ch->T * sizeof(int)); memcpy(c->ids[slot], ch->ids, ch->T * sizeof(int));
  c->n_slots = ns;
  return PPN_OK;
}

int orc_reset(orc_engine* E, const int32_t* env_ids, int32_t n, const int32_t* slots, const int32_t* t0) {
  const OCase* c = &E->c;
  if (!env_ids) n = E->batch;
#pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < n; ++k) {
    OEnv* e = &E->env[env_ids ? env_ids[k] : k];
    orc_reset_grid(c, e);
    for (int l = 0; l < c->nl; ++l) { e->soft[l] = 0; e->amps[l] = 0; }
    for (int g = 0; g < c->nP; ++g) { e->pg[g] = 0; e->qg[g] = c->qg0[g]; e->vg[g] = 0; }
    e->slot = slots ? slots[k] : 0; e->row = (t0 ? t0[k] : 0) - 1; e->epoch = 1; e->nsolve = 0; e->niter = 0; e->nstep = 0;
    memset(e->lev, 0, c->nl);
    orc_advance(c, e, 0, 1);
    const int rc = orc_cascade(c, e, 1);
    orc_cut_flags(c, e);
    e->flag = flag_of(rc); e->ill = 0; e->done = e->flag != 0; e->dead = e->done;
  }
  return PPN_OK;
}

int orc_step(orc_engine* E, const uint8_t* actions, int32_t on_device, int32_t simulate, int32_t auto_reset) {
  (void)on_device;
  const OCase* c = &E->c;
#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < E->batch; ++b) {
    OEnv* e = &E->env[b];
    if (simulate) { env_copy(c, &E->sim[b], e); e = &E->sim[b]; }
    /* a restart that ran out of attempts is taken up again BEFORE the step of the next auto-reset launch (include/ppn.h,
     * ppn_process_game_over): once it succeeds the environment steps in the same call -- what the reference's unbounded
     * recursion amounts to */
    int retried = 0;
    if (e->dead == 3 && auto_reset && !simulate) { orc_game_over_env(c, e, 0); retried = e->dead != 0; }
    orc_step_env(c, e, actions + (size_t)b * c->alen, simulate ? 1 : 0);
    if (auto_reset && !simulate && !retried) orc_game_over_env(c, e, 0);
  }
  return PPN_OK;
}

int orc_process_game_over(orc_engine* E, const uint8_t* mask) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < E->batch; ++b) orc_game_over_env(&E->c, &E->env[b], mask ? mask[b] : 0);
  return PPN_OK;
}

int orc_is_action_valid(orc_engine* E, const uint8_t* actions, uint8_t* valid) {
  for (int b = 0; b < E->batch; ++b) valid[b] = orc_apply_action(&E->c, &E->env[b], actions + (size_t)b * E->c.alen, 0) == 0;
  return PPN_OK;
}

int orc_runpf_batch(orc_engine* E) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int b = 0; b < E->batch; ++b) {
    OEnv* e = &E->env[b]; int its = 0;
    const int rc = orc_solve(&E->c, e, &its);
    e->succ = rc == 0; e->nsolve++; e->niter += its; e->flag = flag_of(rc);
  }
  return PPN_OK;
}

static int field_ptr(const OCase* c, OEnv* e, ppn_field f, void** p, size_t* bytes) {
#define A(ptr, n, t) { *p = (void*)(ptr); *bytes = (size_t)(n) * sizeof(t); return 0; }
#define S(v) { *p = (void*)&(v); *bytes = sizeof(int); return 0; }
  switch (f) {
    case PPN_F_VM: A(e->vm, c->nrows, double) case PPN_F_VA: A(e->va, c->nrows, double)
    case PPN_F_PG: A(e->pg, c->nP, double) case PPN_F_QG: A(e->qg, c->nP, double) case PPN_F_VG: A(e->vg, c->nP, double)
    case PPN_F_PD: A(e->pd, c->nL, double) case PPN_F_QD: A(e->qd, c->nL, double)
    case PPN_F_PF: A(e->pf, c->nl, double) case PPN_F_QF: A(e->qf, c->nl, double) case PPN_F_PT: A(e->pt, c->nl, double)
    case PPN_F_QT: A(e->qt, c->nl, double) case PPN_F_AMPS: A(e->amps, c->nl, double)
    case PPN_F_PRODS_NODES: A(e->pn, c->nP, uint8_t) case PPN_F_LOADS_NODES: A(e->ln, c->nL, uint8_t)
    case PPN_F_LINES_OR_NODES: A(e->on, c->nl, uint8_t) case PPN_F_LINES_EX_NODES: A(e->en, c->nl, uint8_t)
    case PPN_F_LINES_STATUS: A(e->st, c->nl, uint8_t) case PPN_F_BUS_TYPE: A(e->btype, c->nrows, uint8_t)
    case PPN_F_RECONNECTABLE: A(e->rec, c->nl, int) case PPN_F_LINE_COOLDOWN: A(e->lcd, c->nl, int)
    case PPN_F_NODE_COOLDOWN: A(e->ncd, c->nS, int) case PPN_F_SOFT_COUNT: A(e->soft, c->nl, int)
    case PPN_F_FLAG: S(e->flag) case PPN_F_ILLEGAL: S(e->ill) case PPN_F_CASCADE_DEPTH: S(e->depth)
    case PPN_F_ILLEGAL_COUNTS: A(e->illn, 3, int) case PPN_F_ACTION_SWITCHES: A(e->actsw, 2, int)
    case PPN_F_LINE_EVENTS: A(e->lev, c->nl, uint8_t) case PPN_F_SOLVE_OUTCOME: S(e->src)
    case PPN_F_N_SOLVES: S(e->nsolve) case PPN_F_N_ITERS: S(e->niter) case PPN_F_CHRONIC_SLOT: S(e->slot)
    case PPN_F_CHRONIC_ROW: S(e->row) case PPN_F_N_LOADS_CUT: S(e->nlc) case PPN_F_N_PRODS_CUT: S(e->npc)
    case PPN_F_EPOCH: S(e->epoch) case PPN_F_N_STEPS: S(e->nstep)
    default: return -1;
  }
#undef A
#undef S
}

size_t orc_field_bytes(const orc_engine* E, ppn_field f) {
  void* p; size_t n;
  if (f == PPN_F_DONE || f == PPN_F_SUCCESS || f == PPN_F_DEAD) return 1;
  if (field_ptr(&E->c, &((orc_engine*)E)->env[0], f, &p, &n)) return 0;
  return n;
}

int orc_read(orc_engine* E, ppn_field f, void* dst, size_t bytes, int32_t to_host, int32_t from_sim) {
  (void)to_host; (void)bytes;
  for (int b = 0; b < E->batch; ++b) {
    OEnv* e = from_sim ? &E->sim[b] : &E->env[b];
    if (f == PPN_F_DONE) { ((uint8_t*)dst)[b] = (uint8_t)e->done; continue; }
    if (f == PPN_F_SUCCESS) { ((uint8_t*)dst)[b] = (uint8_t)e->succ; continue; }
    if (f == PPN_F_DEAD) { ((uint8_t*)dst)[b] = (uint8_t)e->dead; continue; }
    void* p; size_t n;
    if (field_ptr(&E->c, e, f, &p, &n)) { snprintf(E->err, sizeof E->err, "orc_read: unsupported field %d", (int)f); return PPN_E_INVALID; }
    memcpy((char*)dst + (size_t)b * n, p, n);
  }
  return PPN_OK;
}

int orc_write(orc_engine* E, ppn_field f, const void* src, size_t bytes) {
  (void)bytes;
  for (int b = 0; b < E->batch; ++b) {
    void* p; size_t n;
    if (field_ptr(&E->c, &E->env[b], f, &p, &n)) return PPN_E_INVALID;
    memcpy(p, (const char*)src + (size_t)b * n, n);
  }
  return PPN_OK;
}

int orc_sync(orc_engine* E) { (void)E; return PPN_OK; }
/* oracle-only diagnostic (tests): smallest |V| of an active bus over the successful solves of each environment's last step
 * (before any game-over restart); lets the lock-step tests set aside voltage-collapse cases, which are rounding luck. */
int orc_debug_cycle_stats(long long* out6) { for (int k = 0; k < 6; ++k) out6[k] = g_cyc[k]; return PPN_OK; }
int orc_debug_min_vm(orc_engine* E, double* out) { for (int k = 0; k < E->batch; ++k) out[k] = E->env[k].min_vm; return PPN_OK; }
void* orc_stream(orc_engine* E) { (void)E; return NULL; }
int orc_kernel_time(orc_engine* E, int32_t reset, double* ms, int64_t* n) { (void)E; (void)reset; if (ms) *ms = 0; if (n) *n = 0; return PPN_OK; }
int32_t orc_dim(const orc_engine* E, int32_t which) {
  const OCase* c = &E->c;
  switch (which) { case 0: return c->nS; case 1: return c->nP; case 2: return c->nL; case 3: return c->nl; case 4: return c->alen;
    case 5: return 9 * c->nL + 9 * c->nP + 18 * c->nl + 2 * c->nS + 6; case 6: return E->batch; case 10: return c->n_slots;
    case 12:
#ifdef _OPENMP
      return omp_get_max_threads();
#else
      return 1;
#endif
    default: return 0; }
}
