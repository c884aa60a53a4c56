// ppn_device.h -- device-side data model of the load-flow step engine.
//
// One WAVEFRONT (64 lanes, one 64-thread workgroup) owns one environment.  All per-solve working data
// (admittance matrix, Jacobian/LU, voltages, adjacency bitsets) live in that workgroup's LDS; HBM holds
// only the compulsory per-step state (SURVEY.md 8d "B_io") laid out env-major, one contiguous row per
// environment and field.
//
// The kernels are written in "phase style": LANE_LOOP { ... } regions (straight-line per-lane code on the
// GPU) separated by WSYNC(), with wave-uniform control flow in between.  The same source also compiles
// lane-serially with a host compiler when PPN_EMU is defined -- that build exists ONLY so that the kernel
// logic can be unit-tested in a container without a GPU (tests/test_emu_*.py); the Python package never
// loads it and there is no CPU fallback in the product path.
#pragma once
#include <stdint.h>
#include <math.h>

typedef unsigned long long u64;
typedef unsigned short u16;
typedef unsigned char u8;
struct __attribute__((aligned(16))) ppn_d2 { double x, y; };   // one 16-byte LDS access

#ifdef PPN_EMU
#define PPN_DEV static inline
// Lane order of the lane-serial build.  On the GPU the 64 lanes of a LANE_LOOP region run in lockstep, instruction by instruction;
// the emulation runs one lane's whole region after the other, so a region in which one lane reads (or overwrites) an LDS / global
// location that ANOTHER lane writes in the same region has a lane-order-dependent result here and an instruction-order-dependent
// one on the GPU -- an ordering assumption nothing guarantees.  A correct region gives the same result in EVERY lane order (up to
// the rounding of floating-point atomic sums, whose order the GPU does not fix either).  ppn_emu_set_lane_order (exported by the
// emulation build only; tests/test_emu_lane_order.py): 0 ascending (default), 1 descending, >= 2 a fresh pseudo-random permutation
// for every region, seeded with the value (VERDICT r04 #2a: the race-hunting mode of the emulation).
static int ppn_emu_order_mode_ = 0;
static unsigned ppn_emu_order_state_ = 1u;
static unsigned char ppn_emu_perm_[64];
static inline int ppn_emu_lane_of(int pos) {
  if (ppn_emu_order_mode_ == 0) return pos;
  if (ppn_emu_order_mode_ == 1) return 63 - pos;
  if (pos == 0) {
    for (int i = 0; i < 64; ++i) ppn_emu_perm_[i] = (unsigned char)i;
    for (int i = 63; i > 0; --i) {
      unsigned x = ppn_emu_order_state_; x ^= x << 13; x ^= x >> 17; x ^= x << 5; ppn_emu_order_state_ = x;
      const int j = (int)(x % (unsigned)(i + 1));
      const unsigned char t = ppn_emu_perm_[i]; ppn_emu_perm_[i] = ppn_emu_perm_[j]; ppn_emu_perm_[j] = t;
    }
  }
  return ppn_emu_perm_[pos & 63];
}
#define LANE_LOOP for (int li_ = 0, lane = ppn_emu_lane_of(0); li_ < 64; lane = ppn_emu_lane_of(++li_ < 64 ? li_ : 63))
#define WSYNC() ((void)0)
#define WSYNC_G() ((void)0)
#define PPN_SCHED_FENCE() ((void)0)
#define PPN_WAVE_FULL(prof_, id) ((void)0)
// per-lane variables that live across LANE_LOOP regions (registers on the GPU)
#define PPN_OPAQUE_S(x) (x)
#define PPN_OPAQUE_V(x) (x)
#define LANE_VAR(type, name) type name[64]
#define LANE_ARR(type, name, n) type name[64][n]
#define LV(name) name[lane]
#define LANE_READ(v, l) ((v)[(l)])          // value held by lane l (l wave-uniform)
#define LANE_READ_D(arr, j, l) ((arr)[(l)][(j)])   // element j of a per-lane double array, as held by lane l
#define LANE_READ_DV(v, l) ((v)[(l)])
// references to per-lane state in helper signatures; broadcasts inside a row of 16 lanes (lane T of the row), T a compile-time constant
#define LANE_ARR_REF(type, name, n) type (&name)[64][n]
#define LANE_VAR_REF(type, name) type (&name)[64]
#define LANE_BCAST16_A(arr, j, T) ((arr)[(lane & ~15) + (T)][(j)])
#define LANE_FMAC16_A(arr, j, T, mul) ((arr)[lane][(j)] = __builtin_fma((arr)[(lane & ~15) + (T)][(j)], (mul), (arr)[lane][(j)]))
#define LANE_FMAC16_S(acc, T, mul) ((acc)[lane] = __builtin_fma((acc)[(lane & ~15) + (T)], (mul), (acc)[lane]))
#define LANE_FMAC16_V(acc, src, T, mul) ((acc)[lane] = __builtin_fma((src)[(lane & ~15) + (T)], (mul), (acc)[lane]))
#define PPN_UNI(x) (x)
static inline int ppn_popc(u64 x) { return __builtin_popcountll(x); }
static inline int ppn_ctz(u64 x) { return __builtin_ctzll(x); }
#else
#define PPN_DEV __device__ __forceinline__
// (every phase takes an OPAQUE copy of the lane index: address arithmetic and trip counts that depend only on the lane are
//  loop invariants of the episode / cascade / Newton loops, and hoisted out of them they would sit in VGPRs for the whole
//  kernel -- hundreds of them; recomputing them per phase costs a few VALU instructions)
__device__ __forceinline__ int ppn_opaque_lane(int x) { __asm__ volatile("" : "+v"(x)); return x; }
// (the same for the environment index, in a scalar register: a function that re-derives its environment's row pointers from an
//  opaque copy keeps them alive only for its own duration -- derived once at kernel entry they are ~70 scalar registers that
//  live, i.e. are spilled and reloaded, across every phase of the kernel)
__device__ __forceinline__ int ppn_opaque_uniform(int x) { __asm__ volatile("" : "+s"(x)); return x; }
#define PPN_OPAQUE_S(x) ppn_opaque_uniform(x)
// (and for a per-lane value whose derived quantities must be recomputed where they are used instead of being hoisted out of
//  the Newton loop into registers -- or spilled scalar pairs -- of their own)
#define PPN_OPAQUE_V(x) ((unsigned)ppn_opaque_lane((int)(x)))
#define LANE_LOOP for (int lane = ppn_opaque_lane(lane0), once_ = 1; once_; once_ = 0)
// One wavefront per workgroup: the LDS unit executes the DS instructions of a wave in issue order (a ds_read issued
// after a ds_write / ds_add of any lane of the same wave observes it), so ordering LDS phases only needs the COMPILER
// not to move memory accesses across the point; the waits for loaded registers are the compiler's own.  Nothing is
// drained: stores and atomics of a phase complete underneath the loads of the next one.  (-DPPN_WSYNC_WAIT restores
// an s_waitcnt lgkmcnt(0) at every phase boundary.)  Global loads stay in flight across WSYNC (prefetched schedule
// records); WSYNC_G also orders global memory.
#ifdef PPN_WSYNC_WAIT
#define WSYNC() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define WSYNC() __asm__ volatile("" ::: "memory")
#endif
#define WSYNC_G() __syncthreads()
#define PPN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)      // the instruction scheduler moves nothing across this point
// PPN_WAVE_FULL(): a RECONVERGENCE POINT at the head of every wave-uniform loop (round 5; understood in round 6).
// One wavefront plays one environment and every decision between phases is the same in all 64 lanes.  At the head of such a loop --
// never inside a LANE_LOOP or under a per-lane condition -- the statement is a side-effecting, convergent instruction that all 64
// lanes execute together.  What it prevents (tools/ubench/README_gpu_only_failures.md, "Round 6: the root cause of (iii)";
// tools/ubench/convergent_threading_repro.hip is the same thing in twenty lines):
//   a work loop that ENDS with `if (lane == 0) publish(...)` and BEGINS with `if (lane == 0) item = atomicAdd(...);
//   item = readfirstlane(item);` has, per thread, two branches on the same condition in a row across the back edge.  LLVM's SimplifyCFG
//   threads them: lanes 1..63 jump from the tail straight to the block that holds readfirstlane (their `item` is the constant 0), lane
//   0 goes round through the atomic.  readfirstlane is a CONVERGENT operation and is now the header of a cycle only lanes 1..63 run:
//   the structuriser builds two nested loops, lanes 1..63 spin in the inner one with lane 0 masked off, readfirstlane hands them
//   their own 0, and they replay item 0 for ever with lane 0 missing -- whose stale registers every LANE_READ(x, 0) then takes for
//   wave-uniform values (the chronic slot and row: the address fault rocgdb caught, EXEC = 0xfffffffffffffffe).
// With a side-effecting statement in the loop header there is no empty block to thread through, the loop has ONE back edge, and the
// wave reconverges there by construction.  Any such statement does it (an empty `asm volatile`, `__builtin_amdgcn_wave_barrier()`,
// this one: all three verified on the engine kernel and on the reproducer); this form also re-arms EXEC and fences memory for the
// compiler, one scalar instruction where the phase boundary is a compiler fence anyway.
// Rule for this code base: a loop whose iterations begin or end with lane-0-only code carries PPN_WAVE_FULL at its head (work loops
// of the persistent / rollout / server kernels, episode loop, cascade loop, around every solve, Newton and fast-decoupled loops).
// (The two older GPU-only incidents -- round 2: line-end tables inside the matrix region; round 3: B'/B'' assembly -- behaved
//  differently (no lane missing at the loop heads, -O3 only) and are NOT explained by this; see the README.)
// -DPPN_EXEC_CHECK additionally records (bit `id` of the environment's prof[31], count in prof[30]) lanes found missing there.
#if defined(PPN_WAVE_FULL_OFF)      // (the kernels as they were written until round 5: for the reproducer only)
#define PPN_WAVE_FULL(prof_, id) ((void)0)
#elif defined(PPN_WAVE_FULL_BARRIER_ONLY)      // (reproducer: the compiler fence without the EXEC write)
#define PPN_WAVE_FULL(prof_, id) __asm__ volatile("" ::: "memory")
#elif defined(PPN_WAVE_FULL_EXEC_ONLY)         // (reproducer: the EXEC write without the compiler fence)
#define PPN_WAVE_FULL(prof_, id) __asm__ volatile("s_mov_b64 exec, -1")
#elif defined(PPN_WAVE_FULL_WAVE_BARRIER)      // (reproducer: the compiler's own convergent no-op)
#define PPN_WAVE_FULL(prof_, id) __builtin_amdgcn_wave_barrier()
#elif defined(PPN_EXEC_CHECK)      // (= 2: inside the step body -- ids 4 and up -- lanes found parked are recorded but NOT switched on: where does it start?)
#define PPN_WAVE_FULL(prof_, id) do { const u64 ex_ = __builtin_amdgcn_read_exec(); \
    if (PPN_EXEC_CHECK != 2 || (id) < 4) __asm__ volatile("s_mov_b64 exec, -1" ::: "memory"); \
    if (ex_ != ~0ull && (prof_) != nullptr && (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ctzll(ex_)) { (prof_)[31] |= 1ll << (id); (prof_)[30] += 1; } } while (0)
#else
#define PPN_WAVE_FULL(prof_, id) __asm__ volatile("s_mov_b64 exec, -1" ::: "memory")
#endif
#define LANE_VAR(type, name) type name
#define LANE_ARR(type, name, n) type name[n]
#define LV(name) name
#define LANE_READ(v, l) ((unsigned)__builtin_amdgcn_readlane((int)(v), (l)))
#define LANE_READ_D(arr, j, l) ppn_readlane_d((arr)[(j)], (l))
#define LANE_READ_DV(v, l) ppn_readlane_d((v), (l))
#define LANE_ARR_REF(type, name, n) type (&name)[n]
#define LANE_VAR_REF(type, name) type& name
// DPP row_newbcast: lane T of every row of 16 lanes, as a VGPR operand -- no trip through scalar registers.  The f64 multiply-add
// takes the broadcast as its DPP source: acc += bcast_T(src) * mul is ONE instruction (v_readlane x 2 + v_fma before).  (A DPP
// read needs two wait states after a VALU write of its source, and the hazard recogniser does not look into inline assembly:
// the form for a freshly written source carries its own s_nop.)
template <int T> __device__ __forceinline__ double ppn_bcast16_d(double v) {
  double o;      // (inline assembly with its own s_nop, like the multiply-adds below: `v` usually is the result of one of THEIR statements,
                 //  which the hazard recogniser does not see as a VALU write)
  __asm__ volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v), "n"(T));
  return o;
}
template <int T, bool FRESH> __device__ __forceinline__ void ppn_fmac_bcast16(double& acc, double src, double mul) {
  // FRESH: `src` may have been written by one of the two instructions before this one
  if (FRESH) __asm__ volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
  else __asm__ volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(T));
}
// acc += bcast_T(acc) * mul, accumulator and DPP source ONE operand.  The s_nop is NOT optional even though the source was
// computed a whole elimination step ago: the register allocator may place a copy of the accumulator -- a VALU write -- right in
// front of the statement (seen with a 16-row tail: wrong results in the 5th digit, caught by tools/ubench/dense_tail_test.hip).
template <int T> __device__ __forceinline__ void ppn_fmac_bcast16_self(double& acc, double mul) {
  __asm__ volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mul), "n"(T));
}
#define LANE_BCAST16_A(arr, j, T) ppn_bcast16_d<(T)>((arr)[(j)])
#define LANE_FMAC16_A(arr, j, T, mul) ppn_fmac_bcast16_self<(T)>((arr)[(j)], (mul))
#define LANE_FMAC16_S(acc, T, mul) ppn_fmac_bcast16_self<(T)>((acc), (mul))
#define LANE_FMAC16_V(acc, src, T, mul) ppn_fmac_bcast16<(T), true>((acc), (src), (mul))
__device__ __forceinline__ double ppn_readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define PPN_UNI(x) __builtin_amdgcn_readfirstlane(x)
#ifdef PPN_MARKS      // developer aid (tools/dev): named comment lines in the ISA listing
#define PPN_MARK(s) __asm__ volatile("; MARK " s ::: "memory")
#endif
__device__ __forceinline__ int ppn_popc(u64 x) { return __popcll(x); }
__device__ __forceinline__ int ppn_ctz(u64 x) { return __builtin_ctzll(x); }
#endif
#ifndef PPN_MARK
#define PPN_MARK(s) ((void)0)
#endif
// Bounded per-lane global loads without a branch: the index is clamped into the array (cnt >= 1) and the load is
// unconditional -- scalar base + 32-bit lane offset.  A guarded load (`i < n ? p[i] : 0`) is compiled as an exec-mask save /
// branch / restore around 64-bit address arithmetic, per fetch; the step prologue and the setup of a solve hold dozens.
// ppn_ldc: lanes past the end get the last element (for consumers that are guarded themselves); ppn_ldz: they get 0.
template <class T> PPN_DEV T ppn_ldc(const T* p, int idx, int cnt) {
  return *(const T*)((const char*)p + (size_t)((unsigned)(idx < cnt ? idx : cnt - 1) * (unsigned)sizeof(T)));
}
template <class T> PPN_DEV T ppn_ldz(const T* p, int idx, int cnt) {
  const T v = ppn_ldc(p, idx, cnt);
  return idx < cnt ? v : (T)0;
}
// x if bit `bit` of m is set, else +0.0 -- as an AND with a sign-extended bit field (v_bfe_i32 + 2 v_and_b32): no condition
// register, nothing for the compiler to keep in (spilled) scalar pairs across the iterations
PPN_DEV double ppn_keep_if_bit(double x, unsigned m, int bit) {
  const u64 k = (u64)(long long)(-(int)((m >> bit) & 1u));
  u64 b;
  __builtin_memcpy(&b, &x, 8);
  b &= k;
  __builtin_memcpy(&x, &b, 8);
  return x;
}

// phase profiling (tools/profile_phases.py builds a separate libppn_prof.so with -DPPN_PROF)
#if defined(PPN_PROF) && !defined(PPN_EMU)
#define PROF_BEGIN() long long t_prof_ = clock64()
#define PROF_MARK(E_, id) do { const long long t2_ = clock64(); if (lane0 == 0) (E_).prof[id] += t2_ - t_prof_; t_prof_ = clock64(); } while (0)
#define PROF_BODY_BEGIN() const long long pb_c_ = clock64(), pb_w_ = wall_clock64()
#define PROF_BODY_END(E_) do { if (lane0 == 0) { (E_).prof[14] += clock64() - pb_c_; (E_).prof[15] += wall_clock64() - pb_w_; (E_).prof[13] = pb_w_; } } while (0)      /* [13]: when the body began (100 MHz wall ticks): launch-order studies */
#else
#define PROF_BEGIN() ((void)0)
#ifdef PPN_MARKS
#define PROF_MARK(E_, id) __asm__ volatile("; MARK prof_" #id ::: "memory")
#else
#define PROF_MARK(E_, id) ((void)0)
#endif
#define PROF_BODY_BEGIN() ((void)0)
#define PROF_BODY_END(E_) ((void)0)
#endif

// LDS atomics (one wave per workgroup: conflicting lanes are serialised by the LDS unit in a fixed order, so
// results are reproducible run to run)
#ifdef PPN_EMU
#define LDS_ADD(p, v) (*(p) += (v))
#define LDS_OR(p, v) (*(p) |= (v))
#else
#define LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define LDS_OR(p, v) ((void)__hip_atomic_fetch_or((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif

// Ybus entries (and, halved, lines) a lane holds in registers in the W-word kernels: 64 * PPN_YPL(W) >= buses + 2 lines
#define PPN_YPL(W) ((W) == 1 ? 4 : ((W) == 2 ? 8 : 12))
#define PPN_FILL_REGS 4   // fill-in entries of the Jacobian pattern kept per lane (4 x 64 = 256; beyond that the whole matrix is zeroed)
#define PPN_NONE 0xFFu     // 'no internal index' in the u8 row -> bus table (max_active_buses <= 254)
#define PPN_PI 3.14159265358979323846

// solve outcomes (internal)
enum { SOLVE_OK = 0, SOLVE_DIVERGED = 1, SOLVE_NOT_CONNECTED = 2, SOLVE_CAPACITY = 4 };

struct DevRules {
  int mode, solver, max_it;
  double tol;
  double hard_coef, n_soft_consecutive;
  int n_hard_broken, n_soft_broken, horizon;
  int max_prods_cut, max_loads_cut;
  int n_line_cooldown, n_node_cooldown;
  int max_subs, max_lines, max_total;
  int hard_mode;
  int loop_mode; unsigned seed;   // chronic looping (PPN_LOOP_*), seed of the per-environment chronic draws (PPN_LOOP_RANDOM)
  double rw[13];     // ppn_reward_params, in declaration order
};

// RESTART MEMO (round 6, include/ppn.h: ppn_restart_memo; off unless asked for).  process_game_over of an episode that ended at
// chronic position (slot, row) -- reset_grid, the next timestep, the cascade from the flat start, repeated while the restarted grid
// diverges as well (game.py:762-797) -- is a function of that position alone, with one exception: reset_grid does not clear the
// soft-overflow counters (quirk q3), which decide cuts for lines whose counter has reached n_soft_consecutive and are incremented for
// the lines still overflowed at the end.  So for an environment that ends with every counter BELOW the threshold the restarted
// state is f(slot, row), up to counter[l] = overflowed_after[l] ? counter[l] + 1 : 0 -- and the engine may keep it: `index[key]`
// names the snapshot (a packed row of `stride` bytes in `blob`: memo_xfer, ppn_game.inc) of key = c_off[slot] + slot + row + 1, or
// -1.  The first restart from a position computes and saves, the later ones copy 18 KB instead of running ~6 Newton iterations;
// solve / iteration / epoch counters move by what the computed restart added (meta).  Not used with random chronic looping.
struct DevMemo {
  int* index;        // [n_keys] snapshot number, -1 none, -2 being written
  int* count;        // [1] snapshots handed out
  int cap, n_keys;
  unsigned char* blob;
  size_t stride;
  int* meta;         // [cap x 4] solves, iterations, epochs the computed restart added; spare
  int* tmp;          // [batch x 8] key (-1: not eligible) and counters of an environment at the moment its restart began
  int* stats;        // [4] restarts served from a snapshot (apply kernel), snapshots saved, restarts not eligible, spare
};

// Static (shared by all environments) + chronic tensors.  All pointers are device pointers.
struct DevCase {
  int nS, nP, nL, nl, nrows, ntopo, alen, obslen;
  int NB, YCAP, LUCAP, ECAP;   // capacities: active buses, Ybus entries, LU doubles (Newton: 2 (ECAP + QCAP)), filled pattern entries
  int QCAP;                    // capacity of the Q plane of the Newton storage: pattern entries in rows of PQ buses (<= ECAP)
  double baseMVA;
  const double *bus_gs, *bus_bs, *bus_kv, *vm0, *va0;   // [nrows]  (va0 degrees)
  const double* gen_kv;                                 // [2 nP] baseKV of the bus row a production sits on: busbar 0, busbar 1
  const double* line_kv;                                // [2 nl] baseKV of the bus row at a line's origin: busbar 0, busbar 1
  const int *gen_sub, *load_sub, *or_sub, *ex_sub;       // substation index of each element
  const int *sub_load;                                   // [nS] load index at substation or -1
  const double *gen_qmax, *gen_qmin, *gen_qg0;           // [nP]  (qg0: case value of gen[:,QG])
  const double *ly;      // [8][nl] component-major: yff.re, yff.im, yft.re, yft.im, ytf.re, ytf.im, ytt.re, ytt.im  (status on)
  const double *lb;      // [nl*10] B' (ff,ft,tf,tt), B'' (ff,ft,tf,tt), bdc, pfinj
  const int *pos_row;    // [nrows] bus row at elimination position p (static min-degree order, twins adjacent)
  // static level schedule of the elimination forest of the base graph (valid for every topology obtained by line
  // cuts and node splits): level of busbar (s,node) = 2*level(s) + node
  const int *lvl_row;    // [nrows] bus rows sorted by (level, elimination position)
  const int *lvl_start;  // [nlev+1] start of each level inside lvl_row
  int nlev;
  int MCAP, TCAP;        // schedule capacities: pivot-neighbour pairs, update triples
  // per-environment schedule cache (global memory): byte offsets of its tables inside one environment's blob
  int co_sig, co_r2s, co_i2r, co_ediag, co_ydiag, co_rowptr, co_le4, co_ly4, co_ymeta, co_lvl, co_tail, co_fill, co_trik, cache_stride;
  // the schedule of the reference topology (every element on busbar 0), shared by all environments: one copy of the
  // tables and records that stays in the L2s instead of `batch` private ones streaming from HBM.  Null until the first
  // ppn_reset has produced it.
  const u8 *b_cache; const u64 *b_tri, *b_pair; const unsigned *b_piv;
  const int *sub_le_ptr; // [nS+1]  CSR of line ends per substation
  const int *sub_le;     //         (line << 1) | end   (end 0 = origin, 1 = extremity)
  const int *elem_sub;   // [ntopo] substation of each element of [prods | loads | lines_or | lines_ex]
  const u8 *status0;     // [nl] initial line status
  const double *limits;  // [nl] thermal limits (A)
  const int *sub_ids;    // [nS] external substation ids (observation)
  int slack_row;
  // chronics: slot s occupies rows [c_off[s], c_off[s]+c_T[s]) of every tensor
  int n_slots;
  const float *c_pp, *c_pv, *c_lp, *c_lq, *c_ppp, *c_pvp, *c_lpp, *c_lqp, *c_mt, *c_hz;
  const int* c_mnext;              // [rows x nl] first row >= this one of the same chronic with a maintenance on the line (chronic-relative index; INT_MAX: none) -- the observation's planned-maintenance field in one load
  const int *c_off, *c_T, *c_next, *c_roll, *c_restart;
  const int* c_roll2;    // [n_slots x n_slots] row loaded first when the chronic `old` rolls over into `new` (quirk q2), any pair
  const int *c_dates;    // [rows x 6]
  DevRules R;
};

// Mutable per-environment state, env-major: field[env * n + k].
struct DevState {
  double *vm, *va;                 // [nrows]  (va degrees)
  double *pg, *qg, *vg;            // [nP]
  double *pd, *qd;                 // [nL]
  double *pf, *qf, *pt, *qt, *amps;  // [nl]
  u8 *pn, *ln, *on, *en, *st;      // node bits / line status
  int *rec, *lcd, *ncd, *soft;     // counters
  u8 *done, *dead, *succ, *btype;   // done: reported by the last step; dead: must be reset before stepping
  int *flag, *ill, *depth, *nsolve, *niter, *slot, *row, *nlc, *npc, *epoch;
  u8* big;                         // two-capacity stepping (ppn_engine.hip): 1 = the topology this environment's next step solves on needs the large
                                   // matrix storage (written by the schedule pre-pass before every step)
  int* nbuild;                     // schedules this environment's own wavefront had to build inside a solve since ppn_reset (internal field 102: with the
                                   // schedule pre-pass on, that is what the pre-pass did not foresee)
  int* nstep;                      // Game.step calls this environment has executed since ppn_reset (PPN_F_N_STEPS)
  u8* lev;                         // [nl] PPN_EV_* bits of the last step
  int* src;                        // outcome (SOLVE_*) of the last solve of the last step's cascade
  unsigned* draws;                 // chronics drawn so far by this environment (PPN_LOOP_RANDOM)
  int* prow;                       // absolute chronic row whose planned_* series the observation shows: the row just loaded, or -- in a
                                   // simulation -- the row the simulation STARTED from (game.py:410-413: simulate does not advance the entries)
  long long* prof;                 // [32] cycle counters per phase (only written by -DPPN_PROF builds)
  double* reward;                  // [5] reward components of the last step
  double* ret;                     // sum of the reward components over the steps executed since ppn_reset (PPN_F_RETURN)
  double* report;                  // [3] (done, flag, reward sum) of the last step, one row per environment (PPN_F_STEP_REPORT)
  int *illn, *actsw;               // [3] illegal-action counts, [2] node / line switches of the action after the step
  float* prio;                     // expected cost of the NEXT step (largest ampere flow / limit after this one): launch order
  // per-environment solve workspace (L2-resident, streamed sequentially by the numeric phases)
  u64 *ws_tri, *ws_pair;           // [TCAP], [MCAP] update triples / (pivot, neighbour) pairs with entry indices
  unsigned *ws_piv;                // [NB] pivots (diagonal entry | k << 16)
  u8 *ws_cache;                    // [cache_stride] schedule cache: header, node-assignment signature, index tables
};

// LDS carve-up (pointers into the workgroup's dynamic shared memory).
//   * arrays that live across a whole step (topology working copies, cascade flags, ampere flows),
//   * arrays that live across one solve (bus maps, types, the vectors the Newton loop accesses at random),
//   * ONE region R that is, in turn: the scratch of schedule_build; the setup scratch of a solve (adjacency bitsets,
//     temporary Ybus, production flags) next to the solve's bus vectors (vm, va, injections, mismatch); the matrix storage.
//     The Newton kernels keep the bus vectors in registers while their matrix storage -- which overlays them -- is alive,
//     the fast-decoupled / DC kernels have them next to their (smaller) factors.
// Newton matrix storage = two planes of 16-byte half blocks on the filled pattern: the P plane holds, for EVERY pattern entry
// (i,j), the row of the P_i equation (dP_i/dVa_j, dP_i/dVm_j); the Q plane holds the row of the Q_i equation only for buses
// that have one (PQ buses: nv == 2).  A PV bus or the reference bus has no Q row, so its entries take no Q storage: qrel[i] is
// the byte offset that turns entry index e of a row-i entry into its Q half ((char*)lu + qrel[i] + 16 e), 0xFFFF = no Q row.
#ifndef PPN_TAIL_BUSES
#define PPN_TAIL_BUSES 8          // multiple of 2, at most 8 (2 rows per bus, 16 rows in a DPP row of 16 lanes per column block); until round 3: 6 (one row of 16 lanes held whole rows)
#endif
struct Smem {
  // across the step
  u8 *st, *on, *en, *pn, *ln, *touched;
  u8 *over, *subchg;                  // between solves only: they share the bytes of lf / lt (which live inside a solve only)
  double* amps;                       // ampere flows of the last solve: at the head of region R (dead from the solve's outputs to the next solve's setup)
  // across a solve
  u8 *r2s, *nv, *lf, *lt;             // r2s: bus row -> schedule index (every busbar of the schedule; the live ones are those with touched[row])
  u16 *qrel;
  double* tinv;                       // fast-decoupled kernels: inverses of the dense tails of B' and B'' (2 x T x T, row-major)
  double *vc, *rhs, *zero;            // vc: V = vc[2i] + j vc[2i+1]; zero: {0, 0, 0, 1} = the halves a missing Q row reads as
  // region R
  double* lu;
  double *vm, *va, *psp, *qsp, *mr, *mi;      // bus vectors (view of R)
  u64 *adj0;                                   // setup scratch of a solve (view of R)
  double *yre, *yim;
  u8 *hasgen, *genon;
  u64* adjF;                                   // schedule_build scratch (view of R, with adj0)
  u16 *yptr, *scn, *moffq, *toffq, *rowptr, *lvlp, *lvlm, *lvlt, *int2row, *ediag;
  u8 *pvl, *kq, *mem, *mown;
  unsigned *pfx0, *pfxF;              // schedule_build: per row, the number of set bits below word 1 | 2 | 3 of its adj0 / adjF bitset (one byte each)
  // compact carve only (is_action_valid), schedule pre-pass
  u8* act;
  int* scr;                           // schedule pre-pass only: exchange words of the workgroup-wide collectives (sb_scan_u16, sb_sum_i)
};
#define PPN_QNONE 0xFFFFu

#ifdef PPN_EMU
#define PPN_HD static inline
#else
#define PPN_HD __host__ __device__ __forceinline__
#endif

// counter-based generator of the chronic draws (include/ppn.h, PPN_LOOP_RANDOM)
PPN_HD unsigned ppn_mix32(unsigned seed, unsigned env, unsigned draw) {
  unsigned h = seed * 0x9E3779B1u ^ (env + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (draw + 1u) * 0xC2B2AE35u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}

// Single definition of the LDS layout: carves `base` into S and returns the total size in bytes
// (call with base == nullptr on the host to size the launch).  NT: solver flavour of the kernel (1: Newton).
// compact = true: only what the kernels without a solve touch (observation gather, is_action_valid) -- a few KB instead of
// the solver's working set, so that all their workgroups are resident at once.
PPN_HD size_t ppn_carve(const DevCase& d, int W, int NT, unsigned char* base, Smem* Sp, bool compact = false) {
  Smem& S = *Sp;
  size_t o = 0;
  const size_t NB = d.NB, nl = d.nl, nrows = d.nrows;
  S.scr = nullptr;
#define PPN_TAKE(field, type, bytes) S.field = (type*)(base + o); o += (((size_t)(bytes)) + 15) & ~(size_t)15;
  if (compact) {
    PPN_TAKE(rhs, double, 2 * NB * 8) PPN_TAKE(touched, u8, nrows) PPN_TAKE(subchg, u8, (size_t)d.nS) PPN_TAKE(act, u8, (size_t)d.alen)
    PPN_TAKE(st, u8, nl) PPN_TAKE(on, u8, nl) PPN_TAKE(en, u8, nl) PPN_TAKE(pn, u8, (size_t)d.nP) PPN_TAKE(ln, u8, (size_t)d.nL)
    return o;
  }
  // (LDS is handed out in granules of 1280 bytes -- 128 per CU: seven environments per CU need 18 granules or fewer each, 23040
  //  bytes -- so the byte arrays are packed back to back, only the group is aligned)
#define PPN_TAKE8(field, bytes) S.field = (u8*)(base + o); o += (size_t)(bytes);
  PPN_TAKE8(st, nl) PPN_TAKE8(on, nl) PPN_TAKE8(en, nl) PPN_TAKE8(pn, (size_t)d.nP) PPN_TAKE8(ln, (size_t)d.nL)
  PPN_TAKE8(touched, nrows) PPN_TAKE8(r2s, nrows) PPN_TAKE8(nv, NB)
  // line-end tables (schedule index of a line's two busbars).  (Moving the Newton kernels' copies into region R -- they only
  // need them up to the connectivity sweep -- broke the 4-word kernels on the GPU and stays out.)
  const size_t lb = (nl > (size_t)d.nS ? nl : (size_t)d.nS);
  { PPN_TAKE8(lf, lb) PPN_TAKE8(lt, lb) }
  o = (o + 15) & ~(size_t)15;
  PPN_TAKE(qrel, u16, NB * 2)
  PPN_TAKE(vc, double, 2 * NB * 8) PPN_TAKE(rhs, double, 2 * NB * 8) PPN_TAKE(zero, double, 32)
  S.tinv = nullptr;      // (until round 3: the inverses of the fast-decoupled tails; they live in registers now)
  const size_t r0 = o;
  S.lu = (double*)(base + r0);
  S.amps = (double*)(base + r0);
  // view: schedule_build scratch
  PPN_TAKE(adj0, u64, NB * W * 8)
#undef PPN_TAKE8
  S.over = S.lf; S.subchg = S.lt;      // between solves only (cascade flags, per-substation action flags): they share the bytes of lf / lt
  const size_t after_adj0 = o;
  PPN_TAKE(adjF, u64, NB * W * 8)
  PPN_TAKE(yptr, u16, (NB + 1) * 2) PPN_TAKE(scn, u16, (nrows + 1) * 2) PPN_TAKE(moffq, u16, (NB + 1) * 2) PPN_TAKE(toffq, u16, (NB + 1) * 2)
  PPN_TAKE(rowptr, u16, (NB + 1) * 2) PPN_TAKE(lvlp, u16, ((size_t)d.nlev + 1) * 2) PPN_TAKE(lvlm, u16, ((size_t)d.nlev + 1) * 2)
  PPN_TAKE(lvlt, u16, ((size_t)d.nlev + 1) * 2) PPN_TAKE(int2row, u16, NB * 2) PPN_TAKE(ediag, u16, NB * 2)
  PPN_TAKE(pvl, u8, NB) PPN_TAKE(kq, u8, NB) PPN_TAKE(mem, u8, (size_t)d.MCAP) PPN_TAKE(mown, u8, (size_t)d.MCAP)
  PPN_TAKE(pfx0, unsigned, NB * 4) PPN_TAKE(pfxF, unsigned, NB * 4)
  const size_t build_end = o;
  // view: setup scratch of a solve (adj0 shared with the view above), then the bus vectors
  o = after_adj0;
  PPN_TAKE(yre, double, (size_t)d.YCAP * 8) PPN_TAKE(yim, double, (size_t)d.YCAP * 8)
  PPN_TAKE(hasgen, u8, NB) PPN_TAKE(genon, u8, NB)
  const size_t fd_lu_end = r0 + (size_t)d.ECAP * 16;      // B' and B'' as scalar matrices: 2 x ECAP doubles
  if (!NT && o < fd_lu_end) o = fd_lu_end;               // (fast-decoupled / DC: the vectors sit beside the factors)
  PPN_TAKE(vm, double, NB * 8) PPN_TAKE(va, double, NB * 8) PPN_TAKE(psp, double, NB * 8) PPN_TAKE(qsp, double, NB * 8)
  PPN_TAKE(mr, double, NB * 8) PPN_TAKE(mi, double, NB * 8)
  const size_t vec_end = o;
  const size_t lu_end = NT ? r0 + ((size_t)d.ECAP + (size_t)d.QCAP) * 16 : fd_lu_end;
  o = build_end;
  if (vec_end > o) o = vec_end;
  if (lu_end > o) o = lu_end;
  if (r0 + nl * 8 > o) o = r0 + nl * 8;
#undef PPN_TAKE
  return (o + 15) & ~(size_t)15;
}

// LDS of the schedule pre-pass (K_SCHED): the node bits it builds for, the action, and schedule_build's scratch view -- nothing of
// a solve.  (IEEE-118 with every busbar active: ~23 KB, seven four-wave workgroups per CU.)
PPN_HD size_t ppn_carve_sched(const DevCase& d, int W, unsigned char* base, Smem* Sp) {
  Smem& S = *Sp;
  size_t o = 0;
  const size_t NB = d.NB, nl = d.nl, nrows = d.nrows;
#define PPN_TAKE(field, type, bytes) S.field = (type*)(base + o); o += (((size_t)(bytes)) + 15) & ~(size_t)15;
  PPN_TAKE(scr, int, 64)
  PPN_TAKE(act, u8, (size_t)d.alen) PPN_TAKE(subchg, u8, (size_t)d.nS)
  PPN_TAKE(on, u8, nl) PPN_TAKE(en, u8, nl) PPN_TAKE(pn, u8, (size_t)d.nP) PPN_TAKE(ln, u8, (size_t)d.nL)
  PPN_TAKE(touched, u8, nrows) PPN_TAKE(r2s, u8, nrows)
  PPN_TAKE(lf, u8, nl) PPN_TAKE(lt, u8, nl)
  PPN_TAKE(adj0, u64, NB * W * 8) PPN_TAKE(adjF, u64, NB * W * 8)
  PPN_TAKE(yptr, u16, (NB + 1) * 2) PPN_TAKE(scn, u16, (nrows + 1) * 2) PPN_TAKE(moffq, u16, (NB + 1) * 2) PPN_TAKE(toffq, u16, (NB + 1) * 2)
  PPN_TAKE(rowptr, u16, (NB + 1) * 2) PPN_TAKE(lvlp, u16, ((size_t)d.nlev + 1) * 2) PPN_TAKE(lvlm, u16, ((size_t)d.nlev + 1) * 2)
  PPN_TAKE(lvlt, u16, ((size_t)d.nlev + 1) * 2) PPN_TAKE(int2row, u16, NB * 2) PPN_TAKE(ediag, u16, NB * 2)
  PPN_TAKE(pvl, u8, NB) PPN_TAKE(kq, u8, NB) PPN_TAKE(mem, u8, (size_t)d.MCAP) PPN_TAKE(mown, u8, (size_t)d.MCAP)
  PPN_TAKE(pfx0, unsigned, NB * 4) PPN_TAKE(pfxF, unsigned, NB * 4)
#undef PPN_TAKE
  S.st = nullptr; S.over = nullptr; S.amps = nullptr; S.nv = nullptr; S.qrel = nullptr; S.vc = nullptr; S.rhs = nullptr; S.zero = nullptr;
  S.lu = nullptr; S.tinv = nullptr;
  return o;
}
