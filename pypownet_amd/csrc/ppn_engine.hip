// ppn_engine.hip -- host side of libppn.so: the C ABI of include/ppn.h, device memory management, kernel launches.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ppn_engine.hip -o libppn.so
// (tests only: g++ -DPPN_EMU -x c++ ... -> lane-serial host build of the same kernels, see ppn_device.h)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>

#ifndef PPN_EMU
#include <hip/hip_runtime.h>
#endif

#include "../../include/ppn.h"
#include "ppn_device.h"
#include "ppn_solve.inc"
#include "ppn_game.inc"
#include "ppn_obs.inc"

// ---------------------------------------------------------------------------------------------------------------------
// device memory abstraction (HIP, or plain host memory in the test-only emulation build)
#ifdef PPN_EMU
typedef int hipStream_t;
static int dev_malloc(void** p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xA5, n ? n : 1); return *p ? 0 : -1; }   // (device memory is not zeroed either)
static void dev_free(void* p) { free(p); }
static int dev_h2d(void* d, const void* h, size_t n, hipStream_t) { memcpy(d, h, n); return 0; }
static int dev_d2h(void* h, const void* d, size_t n, hipStream_t) { memcpy(h, d, n); return 0; }
static int dev_d2d(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
static int dev_zero(void* d, size_t n, hipStream_t) { memset(d, 0, n); return 0; }
static const char* dev_err() { return "emulation"; }
#else
#ifdef PPN_DEV_MEM_KNOB      // (experiments, DESIGN 12.9: what does the memory type of the state cost / buy?  PPN_DEV_MEM=fine | uncached)
static int dev_malloc(void** p, size_t n) {
  const char* v = getenv("PPN_DEV_MEM");
  if (v && v[0] == 'f') return hipExtMallocWithFlags(p, n ? n : 1, hipDeviceMallocFinegrained) == hipSuccess ? 0 : -1;
  if (v && v[0] == 'u') return hipExtMallocWithFlags(p, n ? n : 1, hipDeviceMallocUncached) == hipSuccess ? 0 : -1;
  return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : -1;
}
#else
static int dev_malloc(void** p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : -1; }
#endif
static void dev_free(void* p) { if (p) (void)hipFree(p); }
static int dev_h2d(void* d, const void* h, size_t n, hipStream_t s) {
  if (hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;   // host buffers are pageable and caller-owned
}
static int dev_d2h(void* h, const void* d, size_t n, hipStream_t s) {
  if (hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
static int dev_d2d(void* d, const void* s_, size_t n, hipStream_t s) {
  return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1;
}
static int dev_zero(void* d, size_t n, hipStream_t s) { return hipMemsetAsync(d, 0, n, s) == hipSuccess ? 0 : -1; }
static const char* dev_err() { return hipGetErrorString(hipGetLastError()); }
#endif

// ---------------------------------------------------------------------------------------------------------------------
struct HostChronic {
  int T;
  std::vector<float> pp, pv, lp, lq, ppp, pvp, lpp, lqp, mt, hz;
  std::vector<int> ids, dates;
};

struct FieldDesc { void** base; size_t elem; int n; };

struct ppn_engine {
  int batch = 0, device = 0, W = 1;
  DevCase dc;
  DevState st, sim;
  DevState cand;              // forked states of ppn_simulate_candidates (capacity cand_cap rows)
  int cand_cap = 0, n_cand = 0;
  u8* d_cand_actions = nullptr; double* d_cand_obs = nullptr; int* d_cand_ids = nullptr; int* d_cand_perm = nullptr;
  std::vector<void*> allocs;
  std::vector<HostChronic> chronics;
  bool chronics_dirty = true;
  bool mem_failed = false;    // sticky: a device allocation or an upload failed (checked by the entry point that caused it)
  bool newton = false;        // rules: AC mode with the Newton-Raphson solver -> the NT = 1 kernels
  bool maybe_dead = true;     // some environment may be over at the next ppn_step (see ppn_step)
  float restart_prio = 1.3f;      // (PPN_RESTART_PRIO: where in the launch order an environment that owes its restart starts)
  bool pending_restart = false;   // auto_reset = 2: environments that ended the last step still owe their restart (settle_restarts)
  // shared schedule of the reference topology (DevCase::b_*): allocated by ppn_create, filled from the first environment
  // of the first ppn_reset (whose solve built it), then handed to every later launch
  u8* base_cache = nullptr; u64 *base_tri = nullptr, *base_pair = nullptr; unsigned* base_piv = nullptr;
  bool base_ready = false;
  std::vector<void*> chronic_allocs;
  std::vector<void*> cand_allocs;     // candidate slots of ppn_simulate_candidates (released when they are outgrown)
  hipStream_t stream = 0;
  size_t lds_bytes = 0;
  size_t lds_small = 0;       // compact carve of the kernels without a solve (K_VALID, K_OBS)
  size_t lds_sched = 0;       // carve of the schedule pre-pass (ppn_carve_sched)
  // schedule pre-pass (body_sched): PPN_SCHED_PREPASS = 0 never (schedules are built inside the step kernel, as until round 4),
  // 64 / 256 always, with that many threads per environment; default (1): one-wave workgroups when the batch is at least
  // PPN_SCHED_PREPASS_ROUNDS (2) times what the step kernel holds at once -- the throughput regime, where a build inside the step
  // kernel occupies a solver slot (40 KB of LDS, 1 wave per SIMD) for 67-90 k cycles and a build in the pre-pass a 23 KB one; below
  // that a launch lasts as long as its longest environment either way and the pre-pass would only add its own launch to it
  int sched_prepass = 1, sched_threads = 64, sched_rounds = 2;
  // Two-capacity stepping (round 5) of the engines whose busbars may split, when the caller left the matrix capacity to the
  // engine (rules.lu_capacity = 0): the capacity that is safe for every topology (pattern 2.15 x the base one, full Q plane: 55 KB of
  // LDS, two environments per CU) is needed by hardly any (the largest pattern over 10^6 random topologies is 1.39 x).  ppn_step
  // therefore launches the step kernel twice: carved for a SMALL storage -- the largest P = Q capacity that still lets four
  // environments share a CU (1.48 x at IEEE-118) -- for the environments whose schedule fits it, then carved for the large one for
  // the rest (usually none: the workgroups of that launch return at once).  Which class an environment is in is decided BEFORE its
  // step touches anything, by the schedule pre-pass, from the pattern size of the schedule the step will solve on (the small Q
  // plane is as large as the small P plane, so nothing else can overflow).  PPN_TWO_CAP=0 turns it off.
  bool cand_cache = true;       // PPN_CAND_CACHE=0: candidate slots do not keep their schedules (every fork copies the environment's, as until round 5)
  bool two_cap = false, two_cap_allowed = true;
  int two_cap_forced = 0;         // (tests, PPN_TWO_CAP_ECAP: a small storage so small that some environments need the large one)
  int ecap_small = 0;             // pattern capacity (P plane = Q plane) of the small storage
  size_t lds_small_cap = 0;       // LDS per environment of the small-storage launch
  size_t lds_override = 0;        // (launch_w: dynamic LDS of the launch in flight when it is not lds_bytes)
  // Q plane of the Newton storage (Smem): sized from the chronics unless rules.lu_capacity fixes the storage
  bool auto_qcap = true;
  std::vector<int> rowlen_sub, sub_gen_;   // filled-pattern row length of every substation's busbar, production of a substation (-1: none)
  int pattern_pairs = 0;
  int base_fill = 0;
  // host copies of the case tables the MATPOWER-array boundary decodes against (ppn_mpc.inc)
  std::vector<long long> h_ids;
  std::vector<int> h_gen_sub, h_load_sub, h_or_sub, h_ex_sub, h_sub_load;
  std::vector<double> h_br;         // [nl x 5] r, x, b, tap, shift of the case
  std::string err;
  u8* d_actions = nullptr;
  u8* d_rollout = nullptr; size_t d_rollout_cap = 0;     // action sequence of a ppn_rollout handed over in host memory
  double* d_obs = nullptr;
  u8* d_valid = nullptr;
  int* d_perm = nullptr;            // launch order of the step kernel
  int* d_work = nullptr;            // position counter of the persistent step kernel (ppn_kernels.inc)
  int* d_xwork = nullptr; int n_xcd = 0;      // per-XCD work counters of the closed-loop rollout kernel (32 ints apart); XCDs of this GPU (0: not probed yet)
  int* d_progress = nullptr;        // ppn_rollout_policy: steps of the launch every environment has completed
  int resident_slots = 0;           // workgroups of the step kernel the GPU holds at once (CUs x environments per CU)
  size_t resident_for = 0;          // ... computed for this LDS size
  bool persistent = true;           // PPN_PERSISTENT=0: one workgroup per environment whatever the batch, as until round 3
  int persistent_rounds = 4;        // ... persistent from this many environments per resident slot on (PPN_PERSISTENT_ROUNDS)
  bool order_launches = true;       // PPN_LAUNCH_ORDER=0 disables the loading-ordered launch (A/B measurements)
  int* d_ids = nullptr;     // scratch for ppn_reset: env ids, slots, t0 (3 * batch)
  ppn_rules rules;
  // kernel timing
#ifndef PPN_EMU
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
#endif
  double time_ms = 0.0;
  long long launches = 0;
  int timing_every = 1; long long timed_calls = 0;
  // ---- asynchronous session (ppn_async.inc: ppn_async_start / ppn_send / ppn_recv / ppn_async_stop) ----------------------------
  struct Async {
    bool active = false;            // a session is configured
    bool server_running = false;    // the K_SERVE launch is believed to be resident (it may have left on its idle timeout)
    int layout = 0, f32 = 0, sections = 0, stride = 0;
    void* obs = nullptr; double* report_out = nullptr;
    int n_wg = 0, n_wg_req = 0, idle_ms = 100;
    unsigned mask = 0;              // ring size - 1 (a power of two >= 2 x batch)
    unsigned long long* d_items = nullptr; unsigned* d_ctl = nullptr; int* d_rids = nullptr;      // d_rids: [batch] ids of the last receive (rows of its observation gather)
    unsigned long long* h_done = nullptr;                      // pinned: completion ring
    int *h_ids = nullptr, *h_out = nullptr; u8* h_acts = nullptr;      // pinned: ids / action rows of sends, ids of receives
    unsigned long long published = 0;   // items pushed into the ring since the session began (re-publications included)
    unsigned long long received = 0;    // completions handed to the caller (= index of the next ring entry to look at)
    long long n_inflight = 0;           // environments sent and not yet received
    long long restarts = 0, republished = 0;
    std::vector<u8> inflight;
    std::vector<u8> mark;               // scratch of the recovery pass
    hipStream_t s_server = 0, s_async = 0;
  } as;
  hipStream_t launch_stream = 0;  // (launch_w: the stream of the launch in flight when it is not `stream`)
  bool memo_on = false; size_t memo_max_bytes = (size_t)1 << 30; std::vector<void*> memo_allocs; DevMemo memo_h; const DevMemo* d_memo = nullptr; int memo_learn_left = 0, memo_learn_every = 16; long memo_tick = 0;      // (memo_h: host copy of *d_memo)
       // restart memo (DevMemo; ppn_restart_memo)
  int last_step_form = 0;     // ppn_dim(19): kernel form of the last step launch (0 K_STEP, 1 K_STEP_PERSIST, 2 K_STEP_OBS, 3 K_ROLLOUT; + 4: two-capacity stepping)
};

static std::string g_create_error;
static int memo_setup(ppn_engine* e);

static int fail(ppn_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

template <typename T>
static T* upload(ppn_engine* e, const std::vector<T>& v, std::vector<void*>& pool) {
  void* p = nullptr;
  if (dev_malloc(&p, v.size() * sizeof(T) + 16)) { e->mem_failed = true; return nullptr; }
  pool.push_back(p);
  if (!v.empty() && dev_h2d(p, v.data(), v.size() * sizeof(T), e->stream)) e->mem_failed = true;
  return (T*)p;
}

template <typename T>
static T* dalloc(ppn_engine* e, size_t n) {
  void* p = nullptr;
  if (dev_malloc(&p, n * sizeof(T) + 16)) { e->mem_failed = true; return nullptr; }
  e->allocs.push_back(p);
  if (dev_zero(p, n * sizeof(T), e->stream)) e->mem_failed = true;
  return (T*)p;
}

#include "ppn_kernels.inc"

#if defined(PPN_SPLIT_BUILD) && !defined(PPN_EMU)
// Split build (ppn_kernel_tu.hip, __graft_entry__.build_hip): the kernel instances are compiled in translation units of their own, six
// shares in parallel; here they are only named.
#define PPN_EXT(W_, K_, N_) extern template __global__ void ppn_kernel<W_, K_, N_>(const KArgs);
#define PPN_EXT_W(W_) \
  PPN_EXT(W_, K_STEP, 0) PPN_EXT(W_, K_STEP, 1) PPN_EXT(W_, K_GAMEOVER, 0) PPN_EXT(W_, K_GAMEOVER, 1) PPN_EXT(W_, K_RESET, 0) PPN_EXT(W_, K_RESET, 1) \
  PPN_EXT(W_, K_RUNPF, 0) PPN_EXT(W_, K_RUNPF, 1) PPN_EXT(W_, K_ROLLOUT, 0) PPN_EXT(W_, K_ROLLOUT, 1) PPN_EXT(W_, K_STEP_PERSIST, 0) PPN_EXT(W_, K_STEP_PERSIST, 1) \
  PPN_EXT(W_, K_POLICY_ROLLOUT, 0) PPN_EXT(W_, K_POLICY_ROLLOUT, 1) PPN_EXT(W_, K_STEP_OBS, 0) PPN_EXT(W_, K_STEP_OBS, 1) PPN_EXT(W_, K_SERVE, 0) PPN_EXT(W_, K_SERVE, 1) \
  PPN_EXT(W_, K_VALID, 0) PPN_EXT(W_, K_OBS, 0) PPN_EXT(W_, K_POLICY, 0)
PPN_EXT_W(1) PPN_EXT_W(2) PPN_EXT_W(4)
extern template __global__ void ppn_sched_kernel<4, 0, 64>(const KArgs);
extern template __global__ void ppn_sched_kernel<4, 0, 256>(const KArgs);
extern template __global__ void ppn_sched_kernel<4, 1, 64>(const KArgs);
extern template __global__ void ppn_sched_kernel<4, 1, 256>(const KArgs);
#undef PPN_EXT_W
#undef PPN_EXT
#endif

// dispatch on the bitset width of the engine's kernels (-DPPN_ONLY_W1, developer builds for compiler bisection: only the one-word
// kernels are instantiated)
#ifdef PPN_ONLY_W1
#define PPN_BY_W(w_, x1, x2, x4) (x1)
#elif defined(PPN_ONLY_W2)      // (developer builds for same-box A/B measurements of the headline kernels: two-word kernels only)
#define PPN_BY_W(w_, x1, x2, x4) (x2)
#else
#define PPN_BY_W(w_, x1, x2, x4) ((w_) == 1 ? (x1) : ((w_) == 2 ? (x2) : (x4)))
#endif

template <int W, int KIND, int NT>
static int launch_w(ppn_engine* e, const KArgs& a, int nblocks, bool timed) {
#ifdef PPN_EMU
  (void)timed;
  std::vector<unsigned char> lds(std::max(e->lds_bytes, e->lds_small) + 64);
  unsigned char* base = (unsigned char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  Smem S;
  ppn_carve(a.d, W, NT, base, &S, KIND == K_VALID || KIND == K_OBS || KIND == K_POLICY);
  if (KIND == K_SERVE) {
    // the step server of an asynchronous session, emulated: every item published so far is played NOW, in a pseudo-random order (on
    // the GPU they complete in whatever order their cascades end), each followed by its completion record
    unsigned head = a.q_ctl[QC_HEAD]; const unsigned tail = a.q_ctl[QC_TAIL];
    for (unsigned k = head; k != tail; ++k) if ((unsigned)(a.q_items[k & a.q_mask] >> 32) != k + 1u) return -1;      // (every published slot carries its stamp)
    std::vector<unsigned> order;
    for (unsigned k = head; k != tail; ++k) order.push_back(k);
    unsigned x = 2463534242u ^ (head * 2654435761u);
    for (size_t i = order.size(); i > 1; --i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; std::swap(order[i - 1], order[x % i]); }
    for (unsigned k : order) {
      const int e_ = (int)(unsigned)(a.q_items[k & a.q_mask] & 0xFFFFFFFFull);
      memset(base, getenv("PPN_EMU_LDS_FILL") ? atoi(getenv("PPN_EMU_LDS_FILL")) : 0xA5, std::max(e->lds_bytes, e->lds_small));
      body_step<W, NT, true>(a.d, a.st, S, a.actions, 0, 1, a.restart_prio, e_, 0, -1, a.memo);
      if (a.obs) { if (a.obs_f32) body_obs<W, float>(a.d, a.st, S, (float*)a.obs, a.obs_sections, a.obs_stride, e_, 0); else body_obs<W, double>(a.d, a.st, S, (double*)a.obs, a.obs_sections, a.obs_stride, e_, 0); }
      if (a.report_out) for (int j = 0; j < 3; ++j) a.report_out[3 * (size_t)e_ + j] = a.st.report[3 * (size_t)e_ + j];
      const unsigned c = a.q_ctl[QC_DONE]++;
      a.q_done[c & a.q_mask] = ((unsigned long long)(c + 1u) << 32) | (unsigned)e_;
    }
    a.q_ctl[QC_HEAD] = tail;
    return 0;
  }
  if (KIND == K_POLICY_ROLLOUT) {      // the items in their hand-out order: (step, environment)
    for (int item = 0; item < a.n_work; ++item) {
      const int s_ = item / a.n_envs, k_ = item - s_ * a.n_envs;
      const int e_ = a.perm ? a.perm[k_] : k_;
      memset(base, getenv("PPN_EMU_LDS_FILL") ? atoi(getenv("PPN_EMU_LDS_FILL")) : 0xA5, std::max(e->lds_bytes, e->lds_small));
      const bool replay_ = a.policy.id == PPN_POLICY_REPLAY;
      if (!replay_) policy_action<W>(a.d, a.st, a.policy, e_, a.policy_out + (size_t)e_ * a.d.alen, 0);
      body_step<W, NT, true>(a.d, a.st, S, replay_ ? a.actions + (size_t)s_ * a.action_step_stride : (const u8*)a.policy_out, 0, 1, a.restart_prio, e_, 0, -1, a.memo);
    }
    return 0;
  }
  for (int env = 0; env < nblocks; ++env) {
    memset(base, getenv("PPN_EMU_LDS_FILL") ? atoi(getenv("PPN_EMU_LDS_FILL")) : 0xA5, std::max(e->lds_bytes, e->lds_small));   // LDS is NOT zero-initialised on the GPU either
    if (KIND == K_STEP || KIND == K_ROLLOUT || KIND == K_STEP_PERSIST) { for (int s_ = 0; s_ < a.n_steps; ++s_) body_step<W, NT>(a.d, a.st, S, a.actions + (size_t)s_ * a.action_step_stride, a.sim, a.auto_reset, a.restart_prio, env, 0, a.cap_class); }
    else if (KIND == K_STEP_OBS) {
      if (a.cap_class >= 0 && (int)a.st.big[env] != a.cap_class) continue;
      body_step<W, NT, true, true>(a.d, a.st, S, a.actions, a.sim, a.auto_reset, a.restart_prio, env, 0, a.cap_class, a.memo);
      if (a.obs_f32) body_obs<W, float>(a.d, a.st, S, (float*)a.obs, a.obs_sections, a.obs_stride, env, 0); else body_obs<W, double>(a.d, a.st, S, (double*)a.obs, a.obs_sections, a.obs_stride, env, 0);
    }
    else if (KIND == K_GAMEOVER) body_game_over<W, NT>(a.d, a.st, S, a.valid, a.sim, env, 0, a.memo);
    else if (KIND == K_RESET) body_reset<W, NT>(a.d, a.st, S, a.ids, a.slots, a.t0, env, 0);
    else if (KIND == K_RUNPF) body_runpf<W, NT>(a.d, a.st, S, env, 0);
    else if (KIND == K_VALID) body_valid(a.d, a.st, S, a.actions, a.valid, env, 0);
    else if (KIND == K_POLICY) body_policy<W>(a.d, a.st, a.policy, a.policy_out, env, 0);
    else if (KIND == K_OBS) { if (a.obs_f32) body_obs<W, float>(a.d, a.st, S, (float*)a.obs, a.obs_sections, a.obs_stride, env, 0); else body_obs<W, double>(a.d, a.st, S, (double*)a.obs, a.obs_sections, a.obs_stride, env, 0); }
  }
  if (timed) e->launches++;
  return 0;
#else
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // the two event records around a step launch cost ~7 us of stream time (2 % of a 0.34 ms step): PPN_KERNEL_TIMING_EVERY = n
  // brackets every n-th timed launch only (bench.py: 4; default: every launch)
  if (timed && e->timing_every > 1 && (e->timed_calls++ % e->timing_every) != 0) timed = false;
  if (timed) {
    if (e->ev_used + 2 > e->ev.size()) {
      for (int k = 0; k < 512; ++k) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return -1; e->ev.push_back(ev); }
    }
    e0 = e->ev[e->ev_used++]; e1 = e->ev[e->ev_used++];
    (void)hipEventRecord(e0, e->stream);
  }
  hipLaunchKernelGGL((ppn_kernel<W, KIND, NT>), dim3(nblocks), dim3(64), (KIND == K_VALID || KIND == K_OBS || KIND == K_POLICY) ? e->lds_small : (e->lds_override ? e->lds_override : e->lds_bytes), e->launch_stream ? e->launch_stream : e->stream, a);
  if (timed) { (void)hipEventRecord(e1, e->stream); e->launches++; }
  return hipGetLastError() == hipSuccess ? 0 : -1;
#endif
}

// Explicit bracket of a GROUP of launches (two-capacity stepping: schedule pre-pass + small-storage launch + large-storage launch are
// one step -- ADVICE r05: timing the first launch alone overstated the step-kernel figures of the configs[4] bench lines).  Subject to
// PPN_KERNEL_TIMING_EVERY like a single timed launch; counts as ONE launch of ppn_kernel_time.
static bool time_group_begin(ppn_engine* e) {
#ifdef PPN_EMU
  (void)e; return true;
#else
  if (e->timing_every > 1 && (e->timed_calls++ % e->timing_every) != 0) return false;
  if (e->ev_used + 2 > e->ev.size()) {
    for (int k = 0; k < 512; ++k) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return false; e->ev.push_back(ev); }
  }
  (void)hipEventRecord(e->ev[e->ev_used], e->stream);
  return true;
#endif
}
static void time_group_end(ppn_engine* e, bool began) {
  if (!began) return;
#ifndef PPN_EMU
  (void)hipEventRecord(e->ev[e->ev_used + 1], e->stream);
  e->ev_used += 2;
#endif
  e->launches++;
}

// Schedule pre-pass of the engines whose busbars may split (body_sched, ppn_game.inc): warms every environment's schedule cache for
// the topology its step is about to solve.
template <int W>
static int launch_sched(ppn_engine* e, const KArgs& a, int nblocks) {
#ifdef PPN_EMU
  std::vector<unsigned char> lds(e->lds_sched + 64);
  unsigned char* base = (unsigned char*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  Smem S;
  ppn_carve_sched(a.d, W, base, &S);
  for (int env = 0; env < nblocks; ++env) {
    memset(base, getenv("PPN_EMU_LDS_FILL") ? atoi(getenv("PPN_EMU_LDS_FILL")) : 0xA5, e->lds_sched);
    if (e->newton) body_sched<W, PPN_TAIL_BUSES, 64>(a.d, a.st, S, a.actions, a.auto_reset, env, 0, a.ecap_small, a.ssrc);
    else body_sched<W, PPN_FD_TAIL_BUSES, 64>(a.d, a.st, S, a.actions, a.auto_reset, env, 0, a.ecap_small, a.ssrc);
  }
  return 0;
#else
  if (e->sched_threads == 256) {
    if (e->newton) hipLaunchKernelGGL((ppn_sched_kernel<W, 1, 256>), dim3(nblocks), dim3(256), e->lds_sched, e->stream, a);
    else hipLaunchKernelGGL((ppn_sched_kernel<W, 0, 256>), dim3(nblocks), dim3(256), e->lds_sched, e->stream, a);
  } else {
    if (e->newton) hipLaunchKernelGGL((ppn_sched_kernel<W, 1, 64>), dim3(nblocks), dim3(64), e->lds_sched, e->stream, a);
    else hipLaunchKernelGGL((ppn_sched_kernel<W, 0, 64>), dim3(nblocks), dim3(64), e->lds_sched, e->stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
#endif
}

// Kernels that contain a solve exist in two flavours (NT = 1: AC Newton-Raphson only; NT = 0: fast-decoupled XB / DC);
// the others only as NT = 0.
template <int W, int KIND>
static int launch_nt(ppn_engine* e, const KArgs& a, int nblocks, bool timed) {
  constexpr bool solves = (KIND == K_STEP || KIND == K_STEP_OBS || KIND == K_STEP_PERSIST || KIND == K_ROLLOUT || KIND == K_POLICY_ROLLOUT || KIND == K_SERVE || KIND == K_GAMEOVER || KIND == K_RESET || KIND == K_RUNPF);
  if (solves && e->newton) return launch_w<W, KIND, solves ? 1 : 0>(e, a, nblocks, timed);
  return launch_w<W, KIND, 0>(e, a, nblocks, timed);
}
template <int KIND>
static int launch(ppn_engine* e, const KArgs& a, int nblocks, bool timed = false) {
#ifdef PPN_ONLY_W1      // (developer builds for compiler bisection, tools/ubench/: the one-word kernels only)
  return e->W == 1 ? launch_nt<1, KIND>(e, a, nblocks, timed) : -1;
#elif defined(PPN_ONLY_W2)
  return e->W == 2 ? launch_nt<2, KIND>(e, a, nblocks, timed) : -1;
#else
  switch (e->W) {
    case 1: return launch_nt<1, KIND>(e, a, nblocks, timed);
    case 2: return launch_nt<2, KIND>(e, a, nblocks, timed);
    default: return launch_nt<4, KIND>(e, a, nblocks, timed);
  }
#endif
}

#ifndef PPN_EMU
template <int W>
static int set_lds_attr(size_t bytes) {
  int rc = 0;
#define PPN_ATTR(K, N) rc |= hipFuncSetAttribute((const void*)ppn_kernel<W, K, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess;
  PPN_ATTR(K_STEP, 0) PPN_ATTR(K_STEP, 1) PPN_ATTR(K_GAMEOVER, 0) PPN_ATTR(K_GAMEOVER, 1) PPN_ATTR(K_RESET, 0) PPN_ATTR(K_RESET, 1)
  PPN_ATTR(K_RUNPF, 0) PPN_ATTR(K_RUNPF, 1) PPN_ATTR(K_VALID, 0) PPN_ATTR(K_OBS, 0) PPN_ATTR(K_ROLLOUT, 0) PPN_ATTR(K_ROLLOUT, 1) PPN_ATTR(K_STEP_PERSIST, 0) PPN_ATTR(K_STEP_PERSIST, 1) PPN_ATTR(K_POLICY_ROLLOUT, 0) PPN_ATTR(K_POLICY_ROLLOUT, 1) PPN_ATTR(K_POLICY, 0) PPN_ATTR(K_STEP_OBS, 0) PPN_ATTR(K_STEP_OBS, 1) PPN_ATTR(K_SERVE, 0) PPN_ATTR(K_SERVE, 1)
#undef PPN_ATTR
  return rc;
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// host-side analysis: static elimination order (minimum degree on the substation graph) and its fill
// Minimum degree with MULTIPLE elimination: every round eliminates a maximal independent set of the vertices whose
// degree is within +1 of the current minimum.  Independent vertices share a level of the elimination forest, so the
// forest comes out shallower (IEEE-118: 13 levels instead of 16 for plain minimum degree) at practically the same
// fill (652 vs 648 block entries) -- and the level count is what bounds the GPU's sequential depth.
static void min_degree_order(int nS, const std::vector<int>& f, const std::vector<int>& t, std::vector<int>& order) {
  std::vector<std::vector<char>> adj(nS, std::vector<char>(nS, 0));
  for (size_t k = 0; k < f.size(); ++k) if (f[k] != t[k]) { adj[f[k]][t[k]] = 1; adj[t[k]][f[k]] = 1; }
  std::vector<char> gone(nS, 0);
  order.clear();
  while ((int)order.size() < nS) {
    std::vector<int> deg(nS, 1 << 30);
    int dmin = 1 << 30;
    for (int i = 0; i < nS; ++i) if (!gone[i]) {
      int dg = 0;
      for (int j = 0; j < nS; ++j) if (!gone[j] && adj[i][j]) ++dg;
      deg[i] = dg;
      if (dg < dmin) dmin = dg;
    }
    std::vector<int> cand;
    for (int i = 0; i < nS; ++i) if (!gone[i] && deg[i] <= dmin + 1) cand.push_back(i);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<char> blocked(nS, 0);
    std::vector<int> picked;
    for (int k : cand) {
      if (blocked[k]) continue;
      picked.push_back(k);
      blocked[k] = 1;
      for (int j = 0; j < nS; ++j) if (adj[k][j]) blocked[j] = 1;
    }
    for (int k : picked) {
      gone[k] = 1;
      order.push_back(k);
      std::vector<int> nb;
      for (int j = 0; j < nS; ++j) if (!gone[j] && adj[k][j]) nb.push_back(j);
      for (int a : nb) for (int b : nb) if (a != b) adj[a][b] = 1;
    }
  }
}

// number of (i,j) pairs of the filled pattern (incl. diagonal) when eliminating in `order`
static int filled_pairs(int nS, const std::vector<int>& f, const std::vector<int>& t, const std::vector<int>& order,
                        std::vector<std::vector<char>>* out = nullptr) {
  std::vector<int> pos(nS);
  for (int p = 0; p < nS; ++p) pos[order[p]] = p;
  std::vector<std::vector<char>> a(nS, std::vector<char>(nS, 0));
  for (int i = 0; i < nS; ++i) a[i][i] = 1;
  for (size_t k = 0; k < f.size(); ++k) { a[pos[f[k]]][pos[t[k]]] = 1; a[pos[t[k]]][pos[f[k]]] = 1; }
  for (int k = 0; k < nS; ++k) {
    std::vector<int> nb;
    for (int j = k + 1; j < nS; ++j) if (a[k][j]) nb.push_back(j);
    for (int x : nb) for (int y : nb) a[x][y] = 1;
  }
  int cnt = 0;
  for (int i = 0; i < nS; ++i) for (int j = 0; j < nS; ++j) cnt += a[i][j];
  if (out) *out = a;
  return cnt;
}

// ---------------------------------------------------------------------------------------------------------------------
// Every entry point makes the engine's device current first: engines of different GPUs may live in one process (the
// stream, the events and the allocations all belong to e->device).
static int async_quiesce(ppn_engine* e);
// every entry point but the session's own settles a running asynchronous session first (include/ppn.h, "other calls")
#define PPN_QUIESCE(e_) do { if ((e_) && (e_)->as.server_running) { const int rq_ = async_quiesce(e_); if (rq_) return rq_; } } while (0)
static inline void enter(const ppn_engine* e) {
#ifndef PPN_EMU
  if (e) (void)hipSetDevice(e->device);
#else
  (void)e;
#endif
}

#ifdef PPN_EMU
// (emulation build only: lane order of the LANE_LOOP regions, see ppn_device.h)
extern "C" void ppn_emu_set_lane_order(int mode) { ppn_emu_order_mode_ = mode; ppn_emu_order_state_ = (unsigned)mode * 2654435761u + 12345u; if (!ppn_emu_order_state_) ppn_emu_order_state_ = 1u; }
#endif
extern "C" const char* ppn_version(void) { return "pypownet_amd libppn 0.3 (gfx950)"; }

extern "C" const char* ppn_last_error(const ppn_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

static void async_free(ppn_engine* e);
static void free_all(ppn_engine* e) {
  async_free(e);
  for (void* p : e->allocs) dev_free(p);
  for (void* p : e->chronic_allocs) dev_free(p);
  for (void* p : e->cand_allocs) dev_free(p);
  for (void* p : e->memo_allocs) dev_free(p);
  e->memo_allocs.clear();
  e->allocs.clear(); e->chronic_allocs.clear(); e->cand_allocs.clear();
#ifndef PPN_EMU
  for (auto ev : e->ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
#endif
}

extern "C" int ppn_destroy(ppn_engine* e) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  free_all(e);
  delete e;
  return PPN_OK;
}

static int alloc_state(ppn_engine* e, DevState* s, size_t B) {
  const DevCase& d = e->dc;
  s->vm = dalloc<double>(e, B * d.nrows); s->va = dalloc<double>(e, B * d.nrows);
  s->pg = dalloc<double>(e, B * d.nP); s->qg = dalloc<double>(e, B * d.nP); s->vg = dalloc<double>(e, B * d.nP);
  s->pd = dalloc<double>(e, B * d.nL); s->qd = dalloc<double>(e, B * d.nL);
  s->pf = dalloc<double>(e, B * d.nl); s->qf = dalloc<double>(e, B * d.nl); s->pt = dalloc<double>(e, B * d.nl);
  s->qt = dalloc<double>(e, B * d.nl); s->amps = dalloc<double>(e, B * d.nl);
  s->pn = dalloc<u8>(e, B * d.nP); s->ln = dalloc<u8>(e, B * d.nL); s->on = dalloc<u8>(e, B * d.nl);
  s->en = dalloc<u8>(e, B * d.nl); s->st = dalloc<u8>(e, B * d.nl);
  s->rec = dalloc<int>(e, B * d.nl); s->lcd = dalloc<int>(e, B * d.nl); s->ncd = dalloc<int>(e, B * d.nS);
  s->soft = dalloc<int>(e, B * d.nl);
  s->done = dalloc<u8>(e, B); s->dead = dalloc<u8>(e, B); s->succ = dalloc<u8>(e, B);
  s->btype = dalloc<u8>(e, B * d.nrows);
  s->flag = dalloc<int>(e, B); s->ill = dalloc<int>(e, B); s->depth = dalloc<int>(e, B);
  s->nsolve = dalloc<int>(e, B); s->niter = dalloc<int>(e, B); s->slot = dalloc<int>(e, B);
  s->row = dalloc<int>(e, B); s->nlc = dalloc<int>(e, B); s->npc = dalloc<int>(e, B); s->epoch = dalloc<int>(e, B);
  s->nstep = dalloc<int>(e, B); s->nbuild = dalloc<int>(e, B); s->big = dalloc<u8>(e, B);
  s->prow = dalloc<int>(e, B); s->lev = dalloc<u8>(e, B * d.nl); s->src = dalloc<int>(e, B); s->draws = dalloc<unsigned>(e, B);
  s->prof = dalloc<long long>(e, B * 32);
  s->prio = dalloc<float>(e, B);
  s->reward = dalloc<double>(e, B * 5); s->ret = dalloc<double>(e, B); s->report = dalloc<double>(e, B * 3); s->illn = dalloc<int>(e, B * 3); s->actsw = dalloc<int>(e, B * 2);
  s->ws_tri = dalloc<u64>(e, B * d.TCAP); s->ws_pair = dalloc<u64>(e, B * d.MCAP);
  s->ws_piv = dalloc<unsigned>(e, B * d.NB);
  s->ws_cache = dalloc<u8>(e, B * (size_t)d.cache_stride);   // zero-filled: header.valid == 0
  return e->mem_failed ? -1 : 0;     // (every dalloc above records its failure)
}

struct FieldInfo { size_t elem; int n; size_t off; };   // off: byte offset of the pointer inside DevState

static bool field_info(const ppn_engine* e, ppn_field f, FieldInfo* fi, bool* writable) {
  const DevCase& d = e->dc;
  if ((int)f == 100) {   /* phase cycle counters of -DPPN_PROF builds (tools/profile_phases.py) */
    fi->elem = sizeof(long long); fi->n = 32; fi->off = offsetof(DevState, prof); *writable = true; return true;
  }
  if ((int)f == 103) {   /* two-capacity stepping: capacity class the schedule pre-pass gave every environment for its last step */
    fi->elem = 1; fi->n = 1; fi->off = offsetof(DevState, big); *writable = false; return true;
  }
  if ((int)f == 102) {   /* schedules built INSIDE a solve since the engine was created (what the schedule pre-pass did not foresee) */
    fi->elem = sizeof(int); fi->n = 1; fi->off = offsetof(DevState, nbuild); *writable = false; return true;
  }
  if ((int)f == 101) {   /* raw schedule cache blobs (tools/fill_survey.py reads the headers: fill, records) */
    fi->elem = 1; fi->n = d.cache_stride; fi->off = offsetof(DevState, ws_cache); *writable = false; return true;
  }
#define FI(member, type, count, w) { fi->elem = sizeof(type); fi->n = (count); fi->off = offsetof(DevState, member); *writable = (w); return true; }
  switch (f) {
    case PPN_F_VM: FI(vm, double, d.nrows, true)
    case PPN_F_VA: FI(va, double, d.nrows, true)
    case PPN_F_PG: FI(pg, double, d.nP, true)
    case PPN_F_QG: FI(qg, double, d.nP, true)
    case PPN_F_VG: FI(vg, double, d.nP, true)
    case PPN_F_PD: FI(pd, double, d.nL, true)
    case PPN_F_QD: FI(qd, double, d.nL, true)
    case PPN_F_PF: FI(pf, double, d.nl, false)
    case PPN_F_QF: FI(qf, double, d.nl, false)
    case PPN_F_PT: FI(pt, double, d.nl, false)
    case PPN_F_QT: FI(qt, double, d.nl, false)
    case PPN_F_AMPS: FI(amps, double, d.nl, false)
    case PPN_F_PRODS_NODES: FI(pn, u8, d.nP, true)
    case PPN_F_LOADS_NODES: FI(ln, u8, d.nL, true)
    case PPN_F_LINES_OR_NODES: FI(on, u8, d.nl, true)
    case PPN_F_LINES_EX_NODES: FI(en, u8, d.nl, true)
    case PPN_F_LINES_STATUS: FI(st, u8, d.nl, true)
    case PPN_F_RECONNECTABLE: FI(rec, int, d.nl, true)
    case PPN_F_LINE_COOLDOWN: FI(lcd, int, d.nl, true)
    case PPN_F_NODE_COOLDOWN: FI(ncd, int, d.nS, true)
    case PPN_F_SOFT_COUNT: FI(soft, int, d.nl, true)
    case PPN_F_DONE: FI(done, u8, 1, false)
    case PPN_F_FLAG: FI(flag, int, 1, false)
    case PPN_F_ILLEGAL: FI(ill, int, 1, false)
    case PPN_F_CASCADE_DEPTH: FI(depth, int, 1, false)
    case PPN_F_N_SOLVES: FI(nsolve, int, 1, false)
    case PPN_F_N_ITERS: FI(niter, int, 1, false)
    case PPN_F_CHRONIC_SLOT: FI(slot, int, 1, false)
    case PPN_F_CHRONIC_ROW: FI(row, int, 1, false)
    case PPN_F_N_LOADS_CUT: FI(nlc, int, 1, false)
    case PPN_F_N_PRODS_CUT: FI(npc, int, 1, false)
    case PPN_F_SUCCESS: FI(succ, u8, 1, false)
    case PPN_F_BUS_TYPE: FI(btype, u8, d.nrows, false)
    case PPN_F_REWARD: FI(reward, double, 5, false)
    case PPN_F_ILLEGAL_COUNTS: FI(illn, int, 3, false)
    case PPN_F_ACTION_SWITCHES: FI(actsw, int, 2, false)
    case PPN_F_LINE_EVENTS: FI(lev, u8, d.nl, false)
    case PPN_F_SOLVE_OUTCOME: FI(src, int, 1, false)
    case PPN_F_N_STEPS: FI(nstep, int, 1, false)
    case PPN_F_RETURN: FI(ret, double, 1, false)
    case PPN_F_DEAD: FI(dead, u8, 1, false)
    case PPN_F_EPOCH: FI(epoch, int, 1, false)
    case PPN_F_STEP_REPORT: FI(report, double, 3, false)
    default: return false;
  }
#undef FI
}

extern "C" size_t ppn_field_bytes(const ppn_engine* e, ppn_field f) {
  if (!e) return 0;
  if (f == PPN_F_OBSERVATION) return (size_t)e->dc.obslen * sizeof(double);
  FieldInfo fi; bool w;
  if (!field_info(e, f, &fi, &w)) return 0;
  return fi.elem * fi.n;
}

#ifndef PPN_EMU
template <int W>
static int step_kernel_occupancy(const ppn_engine* e, bool persistent_form = false, size_t lds = 0) {
  int n = 0;
  hipError_t rc;
  if (!lds) lds = e->lds_bytes;
  // (the persistent form is a kernel symbol of its own -- a loop around the body, its own register allocation: the grid of that
  //  launch is sized from ITS occupancy)
  if (persistent_form)
    rc = e->newton ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ppn_kernel<W, K_STEP_PERSIST, 1>, 64, lds)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ppn_kernel<W, K_STEP_PERSIST, 0>, 64, lds);
  else
    rc = e->newton ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ppn_kernel<W, K_STEP, 1>, 64, lds)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ppn_kernel<W, K_STEP, 0>, 64, lds);
  return rc == hipSuccess ? n : -1;
}
#endif

#ifndef PPN_EMU
// workgroups of the step kernel resident at once: CUs x min(what the runtime computes, what the 1280-byte LDS granules allow --
// the runtime's figure was one too high for a 23 248-byte build)
static int resident_slots_of(ppn_engine* e, size_t lds = 0) {
  if (!lds) lds = e->lds_bytes;
  if (e->resident_for == lds && e->resident_slots > 0) return e->resident_slots;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0) return 0;
  const int occ = PPN_BY_W(e->W, step_kernel_occupancy<1>(e, true, lds), step_kernel_occupancy<2>(e, true, lds), step_kernel_occupancy<4>(e, true, lds));
  const int granules = (int)((lds + 1279) / 1280);
  int lds_per_cu = 0;      // bytes of LDS a CU hands out, in granules of 1280 bytes (gfx950: 160 KiB = 128 granules)
  if (hipDeviceGetAttribute(&lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, e->device) != hipSuccess || lds_per_cu <= 0) lds_per_cu = 160 * 1024;
  const int by_lds = granules > 0 ? (lds_per_cu / 1280) / granules : occ;
  const int per_cu = std::max(1, std::min(occ > 0 ? occ : by_lds, by_lds));
  e->resident_slots = cus * per_cu;
  e->resident_for = lds;
  return e->resident_slots;
}
#endif

extern "C" int32_t ppn_dim(const ppn_engine* e, int32_t which) {
  if (!e) return -1;
  const DevCase& d = e->dc;
  if (which == 16) {      // environments (workgroups of the step kernel) resident per CU, as the runtime computes it
#ifdef PPN_EMU
    return 1;
#else
    enter(e);
    return PPN_BY_W(e->W, step_kernel_occupancy<1>(e), step_kernel_occupancy<2>(e), step_kernel_occupancy<4>(e));
#endif
  }
  switch (which) {
    case 0: return d.nS; case 1: return d.nP; case 2: return d.nL; case 3: return d.nl;
    case 4: return d.alen; case 5: return d.obslen; case 6: return e->batch; case 7: return (int32_t)e->lds_bytes;
    case 8: return d.NB; case 9: return d.LUCAP; case 10: return (int32_t)e->chronics.size(); case 11: return e->base_fill;
    case 12: return d.ECAP; case 13: return d.MCAP; case 14: return d.TCAP; case 15: return d.QCAP;
    case 17: return e->two_cap ? e->ecap_small : 0; case 18: return e->two_cap ? (int32_t)e->lds_small_cap : 0;
    case 19: return e->last_step_form;
    default: return -1;
  }
}

static void default_reward(double* rw, double c);
static int rebalance_base_triples(ppn_engine* e);
extern "C" int ppn_create(const ppn_case* c, const ppn_rules* r, int32_t batch, int32_t device, ppn_engine** out) {
  if (!c || !r || !out || batch <= 0) return fail(nullptr, PPN_E_INVALID, "ppn_create: null argument or batch <= 0");
  if (c->n_bus_rows <= 0 || (c->n_bus_rows & 1) || c->bus_cols < 10 || c->gen_cols < 8 || c->branch_cols < 11)
    return fail(nullptr, PPN_E_INVALID, "ppn_create: case arrays have unexpected shapes");
  if (c->n_gen <= 0 || c->n_branch <= 0)
    return fail(nullptr, PPN_E_INVALID, "ppn_create: a case needs at least one production and one line");
  ppn_engine* e = new ppn_engine();
  e->batch = batch; e->device = device; e->rules = *r;
#ifndef PPN_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete e; return fail(nullptr, PPN_E_NODEVICE, "no HIP device visible"); }
  if (hipSetDevice(device) != hipSuccess) { delete e; return fail(nullptr, PPN_E_NODEVICE, "hipSetDevice(%d) failed", device); }
  if (hipStreamCreate(&e->stream) != hipSuccess) { delete e; return fail(nullptr, PPN_E_HIP, "hipStreamCreate failed"); }
#endif
  DevCase& d = e->dc;
  memset(&d, 0, sizeof d);
  const int nrows = c->n_bus_rows, nS = nrows / 2, nP = c->n_gen, nl = c->n_branch;
  const double* bus = c->bus; const double* gen = c->gen; const double* br = c->branch;
  const int bc = c->bus_cols, gc = c->gen_cols, rc = c->branch_cols;
  auto bad = [&](const char* m) { free_all(e); delete e; return fail(nullptr, PPN_E_INVALID, "ppn_create: %s", m); };

  // bus id -> row
  std::vector<long long> ids(nrows);
  for (int i = 0; i < nrows; ++i) ids[i] = (long long)bus[(size_t)i * bc + 0];
  auto row_of = [&](double id) { long long v = (long long)id; for (int i = 0; i < nrows; ++i) if (ids[i] == v) return i; return -1; };
  for (int i = 0; i < nS; ++i) {
    char tw[64];
    snprintf(tw, sizeof tw, "666%lld", ids[i]);
    if (ids[i + nS] != atoll(tw)) return bad("bus row i+nS must be the '666'-twin of row i");
  }
  std::vector<double> gs(nrows), bs(nrows), kv(nrows), vm0(nrows), va0(nrows);
  std::vector<int> loads;   // rows with Pd != 0 or Qd != 0 (grid.py:77)
  int slack_row = -1;
  for (int i = 0; i < nrows; ++i) {
    const double* b = bus + (size_t)i * bc;
    gs[i] = b[4]; bs[i] = b[5]; kv[i] = b[9]; vm0[i] = b[7]; va0[i] = b[8];
    if (b[2] != 0.0 || b[3] != 0.0) loads.push_back(i);
    if (slack_row < 0 && (int)b[1] == 3) slack_row = i;
  }
  if (slack_row < 0) return bad("case has no slack bus (type 3)");
  const int nL = (int)loads.size();
  if (nL <= 0) return bad("a case needs at least one load");
  std::vector<int> gen_sub(nP), load_sub(nL), or_sub(nl), ex_sub(nl), sub_load(nS, -1), sub_gen(nS, -1);
  std::vector<double> qmax(nP), qmin(nP), qg0(nP);
  for (int g = 0; g < nP; ++g) {
    const int r_ = row_of(gen[(size_t)g * gc + 0]);
    if (r_ < 0 || r_ >= nS) return bad("production bus must be a real (non-twin) bus of the case");
    gen_sub[g] = r_;
    if (sub_gen[r_] >= 0) return bad("more than one production per substation is not supported");
    if (g > 0 && gen_sub[g] <= gen_sub[g - 1]) return bad("productions must be sorted by substation");
    sub_gen[r_] = g;
    qmax[g] = gen[(size_t)g * gc + 3]; qmin[g] = gen[(size_t)g * gc + 4]; qg0[g] = gen[(size_t)g * gc + 2];
  }
  for (int q = 0; q < nL; ++q) {
    if (loads[q] >= nS) return bad("loads on twin rows in the reference grid are not supported");
    load_sub[q] = loads[q];
    sub_load[loads[q]] = q;
  }
  std::vector<double> ly((size_t)nl * 8), lb((size_t)nl * 10);
  std::vector<u8> status0(nl);
  for (int l = 0; l < nl; ++l) {
    const double* b = br + (size_t)l * rc;
    const int f = row_of(b[0]), t = row_of(b[1]);
    if (f < 0 || t < 0 || f >= nS || t >= nS) return bad("line endpoints must be real buses of the case");
    or_sub[l] = f; ex_sub[l] = t;
    status0[l] = (u8)(b[10] != 0.0);
    const double rr = b[2], x = b[3], bch = b[4];
    const double tapm = (b[8] != 0.0) ? b[8] : 1.0, sh = b[9] * M_PI / 180.0;
    // makeYbus (status on):  Ys = 1/(r+jx); Ytt = Ys + j b/2; Yff = Ytt/(tap conj(tap)); Yft = -Ys/conj(tap); Ytf = -Ys/tap
    auto branch_y = [&](double r_, double x_, double b_, double tm, double shift, double* o) {
      const double den = r_ * r_ + x_ * x_;
      const double ysr = r_ / den, ysi = -x_ / den;
      const double tr = tm * cos(shift), ti = tm * sin(shift);
      const double t2 = tr * tr + ti * ti;
      const double yttr = ysr, ytti = ysi + b_ / 2.0;
      o[0] = yttr / t2; o[1] = ytti / t2;                                 // Yff
      // -Ys / conj(tap) = -Ys * tap / |tap|^2
      o[2] = -(ysr * tr - ysi * ti) / t2; o[3] = -(ysr * ti + ysi * tr) / t2;   // Yft
      // -Ys / tap = -Ys * conj(tap) / |tap|^2
      o[4] = -(ysr * tr + ysi * ti) / t2; o[5] = -(-ysr * ti + ysi * tr) / t2;  // Ytf
      o[6] = yttr; o[7] = ytti;                                           // Ytt
    };
    branch_y(rr, x, bch, tapm, sh, &ly[(size_t)l * 8]);
    double yp[8], yq[8];
    branch_y(0.0, x, 0.0, 1.0, sh, yp);        // B' (XB): r = 0, b = 0, tap = 1 (shift kept)
    branch_y(rr, x, bch, tapm, 0.0, yq);       // B'': shift = 0
    double* o = &lb[(size_t)l * 10];
    o[0] = -yp[1]; o[1] = -yp[3]; o[2] = -yp[5]; o[3] = -yp[7];
    o[4] = -yq[1]; o[5] = -yq[3]; o[6] = -yq[5]; o[7] = -yq[7];
    o[8] = 1.0 / x / tapm;                     // makeBdc: b = status / x / tap
    o[9] = -o[8] * b[9] * M_PI / 180.0;        // Pfinj = b * (-shift)
  }
  // elimination order and fill
  std::vector<int> order;
  min_degree_order(nS, or_sub, ex_sub, order);
  std::vector<int> pos_row(nrows);
  for (int p = 0; p < nS; ++p) { pos_row[2 * p] = order[p]; pos_row[2 * p + 1] = order[p] + nS; }
  std::vector<std::vector<char>> filled;
  const int pairs = filled_pairs(nS, or_sub, ex_sub, order, &filled);
  // LU doubles of the Newton Jacobian with the case's own bus types (PV where a production sits, slack = ref),
  // and the static level schedule of the elimination forest
  std::vector<int> lvl_row(nrows), lvl_start;
  int base_pairs = 0, base_tri = 0, nlev = 0;
  {
    std::vector<int> pos(nS), nvs(nS);
    for (int p = 0; p < nS; ++p) pos[order[p]] = p;
    for (int s = 0; s < nS; ++s) nvs[pos[s]] = (s == slack_row % nS) ? 0 : (sub_gen[s] >= 0 ? 1 : 2);
    int tot = 0;
    for (int i = 0; i < nS; ++i) for (int j = 0; j < nS; ++j) if (filled[i][j]) tot += nvs[i] * nvs[j];
    e->base_fill = tot;
    std::vector<int> level(nS, 0);          // indexed by elimination position
    for (int k = 0; k < nS; ++k) {
      int c = 0;
      for (int j = k + 1; j < nS; ++j) if (filled[k][j]) { ++c; level[j] = std::max(level[j], level[k] + 1); }
      base_pairs += c; base_tri += c * c;
    }
    int maxl = 0;
    for (int k = 0; k < nS; ++k) maxl = std::max(maxl, level[k]);
    if (getenv("PPN_VERBOSE")) {   // schedule shape of the base topology: pivots / pairs / triples per level
      std::vector<int> np_(maxl + 1, 0), nm_(maxl + 1, 0), nt_(maxl + 1, 0);
      for (int k = 0; k < nS; ++k) {
        int c = 0;
        for (int j = k + 1; j < nS; ++j) if (filled[k][j]) ++c;
        np_[level[k]]++; nm_[level[k]] += c; nt_[level[k]] += c * c;
      }
      for (int lv = 0; lv <= maxl; ++lv) fprintf(stderr, "[ppn] level %2d: %3d pivots %4d pairs %5d triples\n", lv, np_[lv], nm_[lv], nt_[lv]);
      fprintf(stderr, "[ppn] filled block entries %d, pairs %d, triples %d\n", pairs, base_pairs, base_tri);
    }
    nlev = 2 * (maxl + 1);                   // busbar (s, node) sits in level 2*level(s) + node
    lvl_start.assign(nlev + 1, 0);
    int w = 0;
    for (int lv = 0; lv < nlev; ++lv) {
      lvl_start[lv] = w;
      for (int p = 0; p < nS; ++p) if (2 * level[p] + 0 == lv) lvl_row[w++] = order[p];
      for (int p = 0; p < nS; ++p) if (2 * level[p] + 1 == lv) lvl_row[w++] = order[p] + nS;
    }
    lvl_start[nlev] = w;
  }
  // sub -> line ends CSR
  std::vector<int> le_ptr(nS + 1, 0), le;
  for (int l = 0; l < nl; ++l) { le_ptr[or_sub[l] + 1]++; le_ptr[ex_sub[l] + 1]++; }
  for (int s = 0; s < nS; ++s) le_ptr[s + 1] += le_ptr[s];
  le.resize(2 * nl);
  {
    std::vector<int> fill(le_ptr.begin(), le_ptr.end() - 1);
    for (int l = 0; l < nl; ++l) { le[fill[or_sub[l]]++] = (l << 1); le[fill[ex_sub[l]]++] = (l << 1) | 1; }
  }
  std::vector<int> elem_sub;
  elem_sub.insert(elem_sub.end(), gen_sub.begin(), gen_sub.end());
  elem_sub.insert(elem_sub.end(), load_sub.begin(), load_sub.end());
  elem_sub.insert(elem_sub.end(), or_sub.begin(), or_sub.end());
  elem_sub.insert(elem_sub.end(), ex_sub.begin(), ex_sub.end());
  std::vector<int> sub_ids(nS);
  for (int s = 0; s < nS; ++s) sub_ids[s] = (int)ids[s];

  d.nS = nS; d.nP = nP; d.nL = nL; d.nl = nl; d.nrows = nrows; d.ntopo = nP + nL + 2 * nl; d.alen = d.ntopo + nl;
  d.obslen = 9 * nL + 9 * nP + 18 * nl + 2 * nS + 6;
  d.baseMVA = c->base_mva; d.slack_row = slack_row;
  int NB = r->max_active_buses > 0 ? std::min(r->max_active_buses, nrows) : nrows;
  if (NB < 2) NB = 2;
  if (NB > 254) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "more than 254 active busbars are not supported"); }
  if (NB < nS) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "max_active_buses (%d) is below the number of substations (%d)", NB, nS); }
  d.NB = NB;
  e->W = (NB <= 64) ? 1 : (NB <= 128 ? 2 : 4);      // bitset words of the kernel variant (no 3-word build)
  d.YCAP = NB + 2 * nl;
  {
    const int ypl = PPN_YPL(e->W);   // Ybus entries held per lane (registers)
    if (d.YCAP > 64 * ypl) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "Ybus of this case (%d entries) exceeds the register budget of the W=%d kernel", d.YCAP, e->W); }
  }
  d.nlev = nlev;
  if (nlev + 1 > 64) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "elimination forest deeper than 31 substation levels"); }
  {
    const double grow = 1.3 + 1.7 * ((NB > nS) ? (double)(NB - nS) / nS : 0.0);   // node splitting adds busbars and fill
    d.MCAP = ((int)(base_pairs * grow) + 64 + 15) & ~15;
    d.TCAP = ((int)(base_tri * grow) + 256 + 15) & ~15;
    if (d.MCAP > 65000 || d.TCAP > 21000) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "schedule capacity exceeds 16-bit offsets"); }
  }
  // filled-pattern capacity (entries).  Newton storage: a P plane of ECAP half blocks + a Q plane of QCAP half blocks (Smem);
  // QCAP starts at the safe maximum and is sized from the chronics by size_q_plane() once they are loaded
  int ecap = r->lu_capacity > 0 ? (r->lu_capacity + 3) / 4 : 0;
  if (ecap <= 0) {
    const double extra = (NB > nS) ? (double)(NB - nS) / nS : 0.0;   // share of busbars that may be split off
    // line cuts never add fill: without spare busbars the base fill is exact.  Splitting a substation spreads its lines
    // over two busbars, so the graph gets sparser as it grows: over 30 k random topologies of IEEE-118 with up to 210
    // active busbars the largest filled pattern was 1.34x the base one (tools/fill_survey.py); the default leaves
    // 1.15x + 1.0x per doubling (2.15x with every busbar active), rules.lu_capacity overrides it
    ecap = (NB > nS) ? (int)(pairs * (1.15 + 1.0 * extra)) + 16 : pairs;
  }
  d.ECAP = (NB > nS) ? ((ecap + 7) & ~7) : ecap;      // (exact without spare busbars: every 16 bytes count towards the LDS granule)
  d.QCAP = d.ECAP;
  d.LUCAP = 2 * (d.ECAP + d.QCAP);
  // rules.lu_capacity sizes the P plane.  The Q plane: without spare busbars the chronics give an exact bound (size_q_plane);
  // with spare busbars it stays full unless the caller opts into the chronic-derived share (rules.q_plane_auto)
  e->auto_qcap = (NB == nS) || r->q_plane_auto != 0;
  e->pattern_pairs = pairs;
  e->sub_gen_ = sub_gen;
  e->h_ids = ids; e->h_gen_sub = gen_sub; e->h_load_sub = load_sub; e->h_or_sub = or_sub; e->h_ex_sub = ex_sub;
  e->h_sub_load = sub_load;
  e->h_br.resize((size_t)nl * 5);
  for (int l = 0; l < nl; ++l) {
    const double* b = br + (size_t)l * rc;
    double* o = &e->h_br[(size_t)l * 5];
    o[0] = b[2]; o[1] = b[3]; o[2] = b[4]; o[3] = b[8]; o[4] = b[9];
  }
  {
    std::vector<int> pos(nS);
    for (int p = 0; p < nS; ++p) pos[order[p]] = p;
    e->rowlen_sub.assign(nS, 0);
    for (int s_ = 0; s_ < nS; ++s_) for (int j = 0; j < nS; ++j) e->rowlen_sub[s_] += filled[pos[s_]][j];
  }
  if (d.ECAP > 8191) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "LU capacity exceeds the 13-bit diagonal entry of a triple record (8191 pattern entries)"); }
  // Q relocations are 16-bit byte offsets into the storage (qrel = 16 * ECAP + 16 * (first Q entry of the row - row start),
  // 0xFFFF = no Q row): both planes have to end below that
  if (16L * ((long)d.ECAP + d.QCAP) >= 0xFFFFL) { free_all(e); delete e; return fail(nullptr, PPN_E_CAPACITY, "Newton storage (%d + %d half blocks) exceeds the 16-bit Q relocation offsets", d.ECAP, d.QCAP); }
  {   // schedule cache blob of one environment
    size_t o = 64;                      // header: 16 ints
    auto take = [&](int* off, size_t bytes) { *off = (int)o; o += (bytes + 15) & ~(size_t)15; };
    take(&d.co_sig, (size_t)d.ntopo); take(&d.co_r2s, (size_t)nrows); take(&d.co_i2r, (size_t)NB * 2);
    take(&d.co_ediag, (size_t)NB * 2); take(&d.co_ydiag, (size_t)NB * 2); take(&d.co_rowptr, (size_t)(NB + 1) * 2);
    take(&d.co_le4, (size_t)nl * 8); take(&d.co_ly4, (size_t)nl * 8);
    take(&d.co_ymeta, (size_t)d.YCAP * 4); take(&d.co_lvl, (size_t)(nlev + 1) * 8);
    take(&d.co_tail, 32 + 512);         // dense tail: up to 16 bus indices (+pad), up to 16 x 16 entry map (u16)
    take(&d.co_fill, (size_t)PPN_FILL_REGS * 64 * 4);   // fill-in entries of the pattern (the ones no Ybus entry covers)
    take(&d.co_trik, (size_t)d.TCAP * 2);               // second word of the triple records (e_kk | TK_* flags, lu_factor)
    d.cache_stride = (int)o;
  }
  d.bus_gs = upload(e, gs, e->allocs); d.bus_bs = upload(e, bs, e->allocs); d.bus_kv = upload(e, kv, e->allocs);
  {
    std::vector<double> gkv((size_t)2 * nP);
    for (int node = 0; node < 2; ++node) for (int g = 0; g < nP; ++g) gkv[(size_t)node * nP + g] = kv[gen_sub[g] + node * nS];
    d.gen_kv = upload(e, gkv, e->allocs);
    std::vector<double> lkv((size_t)2 * nl);
    for (int node = 0; node < 2; ++node) for (int l = 0; l < nl; ++l) lkv[(size_t)node * nl + l] = kv[or_sub[l] + node * nS];
    d.line_kv = upload(e, lkv, e->allocs);
  }
  d.vm0 = upload(e, vm0, e->allocs); d.va0 = upload(e, va0, e->allocs);
  d.gen_sub = upload(e, gen_sub, e->allocs); d.load_sub = upload(e, load_sub, e->allocs);
  d.or_sub = upload(e, or_sub, e->allocs); d.ex_sub = upload(e, ex_sub, e->allocs);
  d.sub_load = upload(e, sub_load, e->allocs);
  d.gen_qmax = upload(e, qmax, e->allocs); d.gen_qmin = upload(e, qmin, e->allocs); d.gen_qg0 = upload(e, qg0, e->allocs);
  {   // component-major copy of the line admittances: lanes of a wave read consecutive lines -> coalesced
    std::vector<double> lyt((size_t)nl * 8);
    for (int l = 0; l < nl; ++l) for (int c = 0; c < 8; ++c) lyt[(size_t)c * nl + l] = ly[(size_t)l * 8 + c];
    d.ly = upload(e, lyt, e->allocs);
  }
  d.lb = upload(e, lb, e->allocs);
  d.pos_row = upload(e, pos_row, e->allocs);
  d.lvl_row = upload(e, lvl_row, e->allocs); d.lvl_start = upload(e, lvl_start, e->allocs);
  d.sub_le_ptr = upload(e, le_ptr, e->allocs); d.sub_le = upload(e, le, e->allocs);
  d.elem_sub = upload(e, elem_sub, e->allocs);
  d.status0 = upload(e, status0, e->allocs);
  d.sub_ids = upload(e, sub_ids, e->allocs);
  std::vector<double> lim(nl, 1e30);
  d.limits = upload(e, lim, e->allocs);

  DevRules& R = d.R;
  R.mode = r->mode; R.solver = r->solver; R.max_it = r->max_it; R.tol = r->tol;
  e->newton = (r->mode != PPN_MODE_DC && r->solver == 1);
  R.hard_coef = r->hard_overflow_coefficient; R.n_soft_consecutive = r->n_timesteps_consecutive_soft_overflow_breaks;
  R.n_hard_broken = r->n_timesteps_hard_overflow_is_broken; R.n_soft_broken = r->n_timesteps_soft_overflow_is_broken;
  R.horizon = r->n_timesteps_horizon_maintenance;
  R.max_prods_cut = r->max_number_prods_game_over; R.max_loads_cut = r->max_number_loads_game_over;
  R.n_line_cooldown = r->n_timesteps_actionned_line_reactionable;
  R.n_node_cooldown = r->n_timesteps_actionned_node_reactionable;
  R.max_subs = r->max_number_actionned_substations; R.max_lines = r->max_number_actionned_lines;
  R.max_total = r->max_number_actionned_total; R.hard_mode = r->game_over_mode_hard;
  R.loop_mode = r->chronic_looping; R.seed = (unsigned)r->rng_seed;
  if (R.loop_mode != PPN_LOOP_NATURAL && R.loop_mode != PPN_LOOP_FIXED && R.loop_mode != PPN_LOOP_RANDOM) return bad("chronic_looping must be PPN_LOOP_NATURAL, _FIXED or _RANDOM");
  default_reward(R.rw, (double)nS);
  if (R.solver != PPN_SOLVER_NEWTON && R.solver != PPN_SOLVER_FDXB) return bad("solver must be NEWTON or FDXB");
  if (R.max_it <= 0) R.max_it = (R.solver == PPN_SOLVER_NEWTON) ? 10 : 25;
  if (!(R.tol > 0)) R.tol = 1e-6;
  { Smem tmp; e->lds_bytes = ppn_carve(d, e->W, e->newton ? 1 : 0, nullptr, &tmp); e->lds_small = ppn_carve(d, e->W, 0, nullptr, &tmp, true);
    e->lds_sched = ppn_carve_sched(d, e->W, nullptr, &tmp); }
  { const char* v = getenv("PPN_SCHED_PREPASS"); if (v) { const int k = atoi(v); e->sched_prepass = (k == 0) ? 0 : (k == 64 || k == 256 ? 2 : 1); if (k == 256) e->sched_threads = 256; } }
  { const char* v = getenv("PPN_TWO_CAP"); if (v && v[0] == '0') e->two_cap_allowed = false; }
  { const char* v = getenv("PPN_RESTART_MEMO"); if (v && v[0] != '0') { e->memo_on = true; if (atol(v) > 1) e->memo_max_bytes = (size_t)atol(v); } }      // (ppn_restart_memo; tests run whole suites with it on)
  { const char* v = getenv("PPN_TWO_CAP_ECAP"); if (v) e->two_cap_forced = atoi(v); }
  { const char* v = getenv("PPN_SCHED_PREPASS_ROUNDS"); if (v && atoi(v) > 0) e->sched_rounds = atoi(v); }
  if (e->lds_bytes > 160 * 1024) {
    free_all(e); delete e;
    return fail(nullptr, PPN_E_CAPACITY, "case needs %zu bytes of LDS per environment (limit 160 KiB)", e->lds_bytes);
  }

  if (alloc_state(e, &e->st, (size_t)batch) || alloc_state(e, &e->sim, (size_t)batch)) { free_all(e); delete e; return fail(nullptr, PPN_E_HIP, "device allocation failed: %s", dev_err()); }
  e->base_cache = dalloc<u8>(e, (size_t)d.cache_stride); e->base_tri = dalloc<u64>(e, (size_t)d.TCAP);
  e->base_pair = dalloc<u64>(e, (size_t)d.MCAP); e->base_piv = dalloc<unsigned>(e, (size_t)d.NB);
  if (e->mem_failed) { free_all(e); delete e; return fail(nullptr, PPN_E_HIP, "device allocation failed: %s", dev_err()); }
  e->d_actions = dalloc<u8>(e, (size_t)batch * d.alen);
  e->d_valid = dalloc<u8>(e, batch);
  e->d_perm = dalloc<int>(e, batch);
  e->d_work = dalloc<int>(e, 16);
  e->d_xwork = dalloc<int>(e, 16 * 32);
  e->d_progress = dalloc<int>(e, batch);
  { const char* v = getenv("PPN_PERSISTENT"); if (v && v[0] == '0') e->persistent = false; }
  { const char* v = getenv("PPN_PERSISTENT_ROUNDS"); if (v && atoi(v) > 0) e->persistent_rounds = atoi(v); }
  { const char* v = getenv("PPN_LAUNCH_ORDER"); if (v && v[0] == '0') e->order_launches = false; }
  { const char* v = getenv("PPN_KERNEL_TIMING_EVERY"); if (v && atoi(v) > 0) e->timing_every = atoi(v); }
  { const char* v = getenv("PPN_RESTART_PRIO"); if (v) e->restart_prio = (float)atof(v); }
  { const char* v = getenv("PPN_CAND_CACHE"); if (v && v[0] == '0') e->cand_cache = false; }
  e->d_ids = dalloc<int>(e, (size_t)3 * batch);
  e->d_obs = dalloc<double>(e, (size_t)batch * d.obslen);
  if (e->mem_failed) { free_all(e); delete e; return fail(nullptr, PPN_E_HIP, "device allocation or upload failed: %s", dev_err()); }
#ifndef PPN_EMU
  int rc_attr = 0;
  rc_attr = PPN_BY_W(e->W, set_lds_attr<1>(e->lds_bytes), set_lds_attr<2>(e->lds_bytes), set_lds_attr<4>(e->lds_bytes));
  if (rc_attr) { free_all(e); delete e; return fail(nullptr, PPN_E_HIP, "cannot reserve %zu bytes of LDS: %s", e->lds_bytes, dev_err()); }
  if (hipStreamSynchronize(e->stream) != hipSuccess) { free_all(e); delete e; return fail(nullptr, PPN_E_HIP, "upload failed: %s", dev_err()); }
#endif
  *out = e;
  return PPN_OK;
}

extern "C" int ppn_set_thermal_limits(ppn_engine* e, const double* limits) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !limits) return PPN_E_INVALID;
  if (dev_h2d((void*)e->dc.limits, limits, sizeof(double) * e->dc.nl, e->stream)) return fail(e, PPN_E_HIP, "limits upload failed");
  return e->chronics_dirty ? PPN_OK : memo_setup(e);      // (snapshots of restarts under the old limits are void)
}

extern "C" int ppn_restart_memo(ppn_engine* e, int32_t enable, int64_t max_bytes) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  e->memo_on = enable != 0;
  if (max_bytes > 0) e->memo_max_bytes = (size_t)max_bytes;
  return e->chronics_dirty ? PPN_OK : memo_setup(e);      // (sync_chronics builds it with the tables it keys on)
}
extern "C" int64_t ppn_restart_memo_stat(ppn_engine* e, int32_t which) {
  if (!e || !e->d_memo) return which == 3 ? 0 : -1;
  if (which == 3) return e->memo_h.cap;
  if (which == 4) return (int64_t)e->memo_h.stride;
  int v[4] = {0, 0, 0, 0};
#ifndef PPN_EMU
  if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
#endif
  if (which == 0) { if (dev_d2h(v, e->memo_h.count, sizeof(int), e->stream)) return -1; return std::min(v[0], e->memo_h.cap); }
  if (dev_d2h(v, e->memo_h.stats, sizeof v, e->stream)) return -1;
  return which == 1 ? v[0] : (which == 2 ? v[2] : -1);
}

extern "C" int ppn_load_chronic(ppn_engine* e, int32_t slot, const ppn_chronic* c) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !c || c->T <= 0) return PPN_E_INVALID;
  if (slot < 0 || slot > (int)e->chronics.size()) return fail(e, PPN_E_INVALID, "chronic slots must be loaded in order");
  const DevCase& d = e->dc;
  HostChronic h;
  h.T = c->T;
  auto cp = [&](std::vector<float>& v, const float* src, int n) { v.assign(src, src + (size_t)c->T * n); };
  cp(h.pp, c->prods_p, d.nP); cp(h.pv, c->prods_v, d.nP); cp(h.lp, c->loads_p, d.nL); cp(h.lq, c->loads_q, d.nL);
  cp(h.ppp, c->prods_p_planned, d.nP); cp(h.pvp, c->prods_v_planned, d.nP);
  cp(h.lpp, c->loads_p_planned, d.nL); cp(h.lqp, c->loads_q_planned, d.nL);
  cp(h.mt, c->maintenance, d.nl); cp(h.hz, c->hazards, d.nl);
  h.ids.assign(c->ids, c->ids + c->T);
  if (c->dates) h.dates.assign(c->dates, c->dates + (size_t)c->T * 6); else h.dates.assign((size_t)c->T * 6, 0);
  if (slot == (int)e->chronics.size()) e->chronics.push_back(std::move(h)); else e->chronics[slot] = std::move(h);
  e->chronics_dirty = true;
  return PPN_OK;
}

static int index_of(const std::vector<int>& v, int x) { for (size_t i = 0; i < v.size(); ++i) if (v[i] == x) return (int)i; return -1; }

// Q plane of the Newton storage (Smem): a bus has a Q row iff it is a PQ bus -- no production at its busbar, or one that is
// switched off (prods_v <= 0) in the row being played.  Without spare busbars (NB == nS: every element stays with its
// substation) the largest need over every row of every loaded chronic, realised and planned series, is an exact bound (buses
// that lose all their lines only lower it); with spare busbars the same fraction of the filled pattern plus a margin is
// reserved.  A solve that needs more (state written through ppn_write, an unusual split) reports PPN_FLAG_ENGINE_CAPACITY
// like any other capacity; rules.lu_capacity reserves the full planes.
static void setup_two_cap(ppn_engine* e);
static int size_q_plane(ppn_engine* e) {
  DevCase& d = e->dc;
  if (!e->newton || !e->auto_qcap) { setup_two_cap(e); return PPN_OK; }
  const int nS = d.nS, nP = d.nP;
  std::vector<int> gen_sub(nP);
  for (int s_ = 0; s_ < nS; ++s_) if (e->sub_gen_[s_] >= 0) gen_sub[e->sub_gen_[s_]] = s_;
  long base = 0;
  for (int s_ = 0; s_ < nS; ++s_) if (e->sub_gen_[s_] < 0) base += e->rowlen_sub[s_];
  long worst = base;
  for (const HostChronic& h : e->chronics) {
    for (const std::vector<float>* series : {&h.pv, &h.pvp}) {
      for (int t = 0; t < h.T; ++t) {
        long need = base;
        const float* v = series->data() + (size_t)t * nP;
        for (int g = 0; g < nP; ++g) if (v[g] <= 0.0f) need += e->rowlen_sub[gen_sub[g]];
        worst = std::max(worst, need);
      }
    }
  }
  int qcap;
  if (d.NB == nS) qcap = (int)worst;
  else qcap = (int)(((double)worst / e->pattern_pairs + 0.12) * d.ECAP) + 8;
  if (getenv("PPN_QCAP_FULL")) qcap = d.ECAP;
  qcap = std::min(d.ECAP, d.NB == nS ? qcap : ((qcap + 7) & ~7));
  d.QCAP = qcap;
  d.LUCAP = 2 * (d.ECAP + d.QCAP);
  Smem tmp;
  e->lds_bytes = ppn_carve(d, e->W, 1, nullptr, &tmp);
  { const char* v = getenv("PPN_LDS_PAD"); if (v) e->lds_bytes += (size_t)atoi(v); }      // (occupancy experiments: fewer environments per CU)
#ifndef PPN_EMU
  int rc_attr = 0;
  rc_attr = PPN_BY_W(e->W, set_lds_attr<1>(e->lds_bytes), set_lds_attr<2>(e->lds_bytes), set_lds_attr<4>(e->lds_bytes));
  if (rc_attr) return fail(e, PPN_E_HIP, "cannot reserve %zu bytes of LDS: %s", e->lds_bytes, dev_err());
#endif
  setup_two_cap(e);
  return PPN_OK;
}

// Two-capacity stepping (see ppn_engine::two_cap): the small storage = the largest P = Q capacity with which four environments
// share a CU (32 LDS granules of 1280 bytes each), if that still is a sensible margin over the base pattern.
static void setup_two_cap(ppn_engine* e) {
  e->two_cap = false;
  const DevCase& d = e->dc;
  if (!e->two_cap_allowed || e->W != 4 || d.NB <= d.nS || e->rules.lu_capacity > 0 || e->sched_prepass == 0 || e->dc.R.mode == 1) return;
  const size_t target = 32 * 1280;
  if (e->lds_bytes <= target) return;
  const int forced = e->two_cap_forced;
  for (int ecap = forced > 0 ? forced : (d.ECAP & ~7); ecap >= e->pattern_pairs + (forced > 0 ? 0 : 8); ecap -= 8) {
    DevCase t = d;
    t.ECAP = ecap; t.QCAP = ecap; t.LUCAP = 2 * (t.ECAP + t.QCAP);
    if (16L * ((long)t.ECAP + t.QCAP) >= 0xFFFFL) continue;
    Smem tmp;
    const size_t lds = ppn_carve(t, e->W, e->newton ? 1 : 0, nullptr, &tmp);
    if (lds <= target || forced > 0) {
      if (forced <= 0 && ecap < (int)(1.25 * e->pattern_pairs)) return;      // too tight to be worth a second launch
      e->ecap_small = ecap; e->lds_small_cap = lds; e->two_cap = true;
      return;
    }
  }
}

static int sync_chronics(ppn_engine* e) {
  if (!e->chronics_dirty) return PPN_OK;
  if (e->chronics.empty()) return fail(e, PPN_E_STATE, "no chronic loaded");
#ifndef PPN_EMU
  (void)hipStreamSynchronize(e->stream);
#endif
  for (void* p : e->chronic_allocs) dev_free(p);
  e->chronic_allocs.clear();
  const int ns = (int)e->chronics.size();
  std::vector<int> off(ns), T(ns), next(ns), roll(ns), restart(ns), roll2((size_t)ns * ns), dates;
  std::vector<float> pp, pv, lp, lq, ppp, pvp, lpp, lqp, mt, hz;
  std::vector<int> mnext;      // per (row, line): the first row >= this one of the SAME chronic with a maintenance on the line (chronic-relative; INT_MAX: none)
  int rows = 0;
  for (int s = 0; s < ns; ++s) {
    const HostChronic& h = e->chronics[s];
    off[s] = rows; T[s] = h.T; rows += h.T;
    next[s] = (e->rules.chronic_looping == PPN_LOOP_FIXED) ? s : (s + 1) % ns;
    auto app = [](std::vector<float>& dst, const std::vector<float>& src) { dst.insert(dst.end(), src.begin(), src.end()); };
    app(pp, h.pp); app(pv, h.pv); app(lp, h.lp); app(lq, h.lq); app(ppp, h.ppp); app(pvp, h.pvp); app(lpp, h.lpp);
    app(lqp, h.lqp); app(mt, h.mt); app(hz, h.hz);
    dates.insert(dates.end(), h.dates.begin(), h.dates.end());
    {
      const int nl_ = e->dc.nl;
      const size_t base = mnext.size();
      mnext.resize(base + (size_t)h.T * nl_);
      for (int l = 0; l < nl_; ++l) {
        int nx = 0x7fffffff;
        for (int r = h.T - 1; r >= 0; --r) {
          if (h.mt[(size_t)r * nl_ + l] != 0.0f) nx = r;
          mnext[base + (size_t)r * nl_ + l] = nx;
        }
      }
    }
  }
  for (int s = 0; s < ns; ++s) {
    // roll-over (game.py:481-493): get_next_chronic() sets the current id to 0, then the NEXT id is looked up in
    // the OLD chronic's id list and loaded from the NEW chronic (quirk q2)
    const HostChronic& old_ = e->chronics[s];
    const HostChronic& nw = e->chronics[next[s]];
    int i0 = index_of(old_.ids, 0);
    int nid = old_.ids[std::min(std::max(i0, 0) + 1, old_.T - 1)];
    int r_ = index_of(nw.ids, nid);
    roll[s] = r_ < 0 ? std::min(1, nw.T - 1) : r_;
    for (int s2 = 0; s2 < ns; ++s2) {       // the same rule for any successor (PPN_LOOP_RANDOM)
      const HostChronic& n2 = e->chronics[s2];
      const int r2 = index_of(n2.ids, nid);
      roll2[(size_t)s * ns + s2] = r2 < 0 ? std::min(1, n2.T - 1) : r2;
    }
    // hard game over (game.py:770-776): current id None -> get_next_chronic() (id 0) -> next id of the new chronic
    const HostChronic& me = e->chronics[s];
    int j0 = index_of(me.ids, 0);
    restart[s] = std::min(std::max(j0, 0) + 1, me.T - 1);
  }
  DevCase& d = e->dc;
  d.n_slots = ns;
  d.c_pp = upload(e, pp, e->chronic_allocs); d.c_pv = upload(e, pv, e->chronic_allocs);
  d.c_lp = upload(e, lp, e->chronic_allocs); d.c_lq = upload(e, lq, e->chronic_allocs);
  d.c_ppp = upload(e, ppp, e->chronic_allocs); d.c_pvp = upload(e, pvp, e->chronic_allocs);
  d.c_lpp = upload(e, lpp, e->chronic_allocs); d.c_lqp = upload(e, lqp, e->chronic_allocs);
  d.c_mt = upload(e, mt, e->chronic_allocs); d.c_hz = upload(e, hz, e->chronic_allocs);
  d.c_mnext = upload(e, mnext, e->chronic_allocs);
  d.c_off = upload(e, off, e->chronic_allocs); d.c_T = upload(e, T, e->chronic_allocs);
  d.c_next = upload(e, next, e->chronic_allocs); d.c_roll = upload(e, roll, e->chronic_allocs);
  d.c_restart = upload(e, restart, e->chronic_allocs); d.c_dates = upload(e, dates, e->chronic_allocs);
  d.c_roll2 = upload(e, roll2, e->chronic_allocs);
  if (e->mem_failed) { e->mem_failed = false; return fail(e, PPN_E_HIP, "chronic upload failed: %s", dev_err()); }
  e->chronics_dirty = false;
  { const int rcm = memo_setup(e); if (rcm) return rcm; }
  return size_q_plane(e);
}

// Restart memo: (re)built empty whenever the chronics change (the keys are chronic positions) or it is switched on.
static void memo_release(ppn_engine* e) {
#ifndef PPN_EMU
  if (!e->memo_allocs.empty()) (void)hipStreamSynchronize(e->stream);
#endif
  for (void* p : e->memo_allocs) dev_free(p);
  e->memo_allocs.clear();
  e->d_memo = nullptr; memset(&e->memo_h, 0, sizeof e->memo_h);
}
static int memo_setup(ppn_engine* e) {
  memo_release(e);
  if (!e->memo_on || e->dc.R.loop_mode == PPN_LOOP_RANDOM || e->chronics.empty()) return PPN_OK;
  DevMemo& M = e->memo_h;
  long keys = 0;
  for (const auto& c : e->chronics) keys += (long)c.T + 1;
  const size_t stride = ppn_memo_stride(e->dc);
  long cap = std::min<long>(keys, (long)(e->memo_max_bytes / stride));
  if (cap < 1) return PPN_OK;
  auto grab = [&](size_t bytes, int fill) -> void* {
    void* p = nullptr;
    if (dev_malloc(&p, bytes + 16)) return nullptr;
    e->memo_allocs.push_back(p);
    std::vector<unsigned char> h(bytes, (unsigned char)fill);
    if (dev_h2d(p, h.data(), bytes, e->stream)) return nullptr;
#ifndef PPN_EMU
    (void)hipStreamSynchronize(e->stream);      // (the staging vector dies here)
#endif
    return p;
  };
  void* index = grab(sizeof(int) * (size_t)keys, 0xFF);
  void* count = grab(64, 0);
  void* meta = grab(sizeof(int) * 4 * (size_t)cap, 0);
  void* tmp = grab(sizeof(int) * 8 * (size_t)e->batch, 0xFF);
  void* stats = grab(64, 0);
  void* blob = nullptr;
  if (index && count && meta && tmp && stats && !dev_malloc(&blob, stride * (size_t)cap + 16)) e->memo_allocs.push_back(blob);
  if (!blob) { memo_release(e); return fail(e, PPN_E_HIP, "restart memo: device allocation failed (%ld snapshots of %zu bytes)", cap, stride); }
  M.index = (int*)index; M.count = (int*)count; M.cap = (int)cap; M.n_keys = (int)keys; M.blob = (unsigned char*)blob; M.stride = stride;
  M.meta = (int*)meta; M.tmp = (int*)tmp; M.stats = (int*)stats;
  void* dm = nullptr;
  if (dev_malloc(&dm, sizeof(DevMemo) + 16)) { memo_release(e); return fail(e, PPN_E_HIP, "restart memo: device allocation failed"); }
  e->memo_allocs.push_back(dm);
  const DevMemo mh = M;      // (memo_release clears memo_h)
  if (dev_h2d(dm, &mh, sizeof(DevMemo), e->stream)) { memo_release(e); return fail(e, PPN_E_HIP, "restart memo: upload failed"); }
#ifndef PPN_EMU
  (void)hipStreamSynchronize(e->stream);
#endif
  e->d_memo = (const DevMemo*)dm;
  e->memo_learn_left = 32; e->memo_tick = 0;
  { const char* v = getenv("PPN_MEMO_LEARN_STEPS"); if (v) e->memo_learn_left = atoi(v); }
  { const char* v = getenv("PPN_MEMO_LEARN_EVERY"); if (v) e->memo_learn_every = atoi(v); }
  return PPN_OK;
}
static KArgs make_args(ppn_engine* e, bool sim_state) {
  KArgs a;
  memset(&a, 0, sizeof a);
  a.d = e->dc;
  a.st = sim_state ? e->sim : e->st;
  a.n_steps = 1;
  a.cap_class = -1;
  a.memo = sim_state ? nullptr : e->d_memo;
  return a;
}

// Deferred auto-reset (ppn_step with auto_reset = 2): the restart of an episode that ended is owed to the NEXT step launch.
// Whatever looks at the state in between -- ppn_sync, a read of anything but the step's report fields, a write, a simulation,
// a step in another mode -- settles the debt first with the plain process_game_over kernel.
static int settle_restarts(ppn_engine* e) {
  if (!e->pending_restart) return PPN_OK;
  e->pending_restart = false;
  KArgs a = make_args(e, false);
  a.sim = 1;      // only the owed restarts (PPN_F_DEAD = 2)
  if (launch<K_GAMEOVER>(e, a, e->batch)) return fail(e, PPN_E_HIP, "game-over kernel launch failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int ppn_reset(ppn_engine* e, const int32_t* env_ids, int32_t n, const int32_t* chronic_slot, const int32_t* t0) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  int rc = sync_chronics(e);
  if (rc) return rc;
  if (!env_ids) n = e->batch;
  if (n <= 0 || n > e->batch) return fail(e, PPN_E_INVALID, "ppn_reset: bad environment count");
  const int ns = (int)e->chronics.size();
  std::vector<int> buf((size_t)3 * e->batch, 0);
  for (int k = 0; k < n; ++k) {
    const int env = env_ids ? env_ids[k] : k;
    if (env < 0 || env >= e->batch) return fail(e, PPN_E_INVALID, "ppn_reset: environment id out of range");
    const int s = chronic_slot ? chronic_slot[k] : 0;
    if (s < 0 || s >= ns) return fail(e, PPN_E_INVALID, "ppn_reset: chronic slot %d not loaded", s);
    const int t = t0 ? t0[k] : 0;
    if (t < 0 || t >= e->chronics[s].T) return fail(e, PPN_E_INVALID, "ppn_reset: t0 out of range");
    buf[k] = env; buf[e->batch + k] = s; buf[2 * e->batch + k] = t;
  }
  if (dev_h2d(e->d_ids, buf.data(), buf.size() * sizeof(int), e->stream)) return fail(e, PPN_E_HIP, "ppn_reset upload failed");
  KArgs a = make_args(e, false);
  a.ids = e->d_ids; a.slots = e->d_ids + e->batch; a.t0 = e->d_ids + 2 * e->batch;
  if (launch<K_RESET>(e, a, n)) return fail(e, PPN_E_HIP, "reset kernel launch failed: %s", dev_err());
  e->maybe_dead = true;
  if (!e->base_ready) {
    // The environments just reset stand in the reference topology and have built its schedule: the first one's copy
    // becomes the shared one (stream-ordered copies; the one synchronisation checks that the build succeeded).
    const char* off = getenv("PPN_SHARED_SCHEDULE");
    if (!(off && off[0] == '0')) {
      const DevCase& d = e->dc;
      const size_t env = (size_t)(env_ids ? env_ids[0] : 0);
      int rcc = dev_d2d(e->base_cache, e->st.ws_cache + env * (size_t)d.cache_stride, (size_t)d.cache_stride, e->stream);
      rcc |= dev_d2d(e->base_tri, e->st.ws_tri + env * (size_t)d.TCAP, sizeof(u64) * (size_t)d.TCAP, e->stream);
      rcc |= dev_d2d(e->base_pair, e->st.ws_pair + env * (size_t)d.MCAP, sizeof(u64) * (size_t)d.MCAP, e->stream);
      rcc |= dev_d2d(e->base_piv, e->st.ws_piv + env * (size_t)d.NB, sizeof(unsigned) * (size_t)d.NB, e->stream);
      int hdr[16] = {0};
      rcc |= dev_d2h(hdr, e->base_cache, sizeof hdr, e->stream);
      if (rcc) return fail(e, PPN_E_HIP, "ppn_reset: shared schedule copy failed: %s", dev_err());
      if (hdr[0] == 1) {
        if (rebalance_base_triples(e)) return fail(e, PPN_E_HIP, "ppn_reset: shared schedule update failed: %s", dev_err());
        e->dc.b_cache = e->base_cache; e->dc.b_tri = e->base_tri; e->dc.b_pair = e->base_pair; e->dc.b_piv = e->base_piv;
        e->base_ready = true;
      }
    }
  }
  return PPN_OK;
}

// The SHARED schedule re-packed into fewer rounds of 64 lanes (host side, once per engine; PPN_NO_REBALANCE=1 keeps it as built).
// (1) Pivots.  The static level of a busbar is the earliest it can be eliminated at; it may be eliminated at any level up to the
//     one before its first later neighbour's.  Pivots with that slack leave a level whose pair list overflows a multiple of 64
//     (IEEE-118: level 0 holds 99 pairs -- a second round of the backward pass for 35 lanes) for later levels with lanes to
//     spare, the most flexible first; a pivot takes its pair and triple records along.
// (2) Schur updates.  An OFF-DIAGONAL triple of pivot k (level L) -- A(i,j) -= A(i,k) U'(k,j) -- may run in any level from L up
//     to the level before A(i,j) is first READ (as the pivot block or a triple operand of the earlier of i and j; by the dense
//     tail), so the records a level holds beyond a multiple of 64 move on the same way; behind its pivot's level a record finds
//     U'(k,j) in place (TK_UCOOK).  Diagonal triples carry the stores of their pair (U', forward push, inv(D), y') and stay.
// A level's list then is [off-diagonal ...][diagonal ...][moved in ...] and the TK_DCOOK flags of the diagonal lanes are set from
// their final positions (lu_factor).  Dependencies are untouched; only the order of the atomic adds into a block (and into a
// right-hand-side entry) changes.
static int rebalance_base_triples(ppn_engine* e) {
  const DevCase& d = e->dc;
  if (getenv("PPN_NO_REBALANCE")) return 0;
  std::vector<u8> cache((size_t)d.cache_stride);
  if (dev_d2h(cache.data(), e->base_cache, cache.size(), e->stream)) return -1;
  const int* hdr = (const int*)cache.data();
  const int n = hdr[1], nnzF = hdr[2], nla = hdr[4], n_pairs = hdr[5], n_tri = hdr[6];
  const int ltail = hdr[7];      // levels below the dense tail: records only move among those (the Newton and fast-decoupled factorisations stop there;
                                 // the DC one runs every level, for which the moves are as valid)
  if (n <= 0 || nla <= 1 || n_tri <= 0 || n_pairs <= 0 || ltail <= 1) return 0;
  if (n > 255 || n_tri > d.TCAP || n_pairs > d.MCAP || nnzF > 8191) return 0;      // (a malformed schedule stays as built)
  unsigned* lvl = (unsigned*)(cache.data() + d.co_lvl);
  u16* trik = (u16*)(cache.data() + d.co_trik);
  std::vector<unsigned> piv((size_t)n);
  std::vector<u64> pair((size_t)n_pairs), tri((size_t)n_tri);
  if (dev_d2h(piv.data(), e->base_piv, sizeof(unsigned) * (size_t)n, e->stream)) return -1;
  if (dev_d2h(pair.data(), e->base_pair, sizeof(u64) * (size_t)n_pairs, e->stream)) return -1;
  if (dev_d2h(tri.data(), e->base_tri, sizeof(u64) * (size_t)n_tri, e->stream)) return -1;
  std::vector<int> Lp((size_t)nla + 1), Lm((size_t)nla + 1), Lt((size_t)nla + 1);
  for (int lv = 0; lv <= nla; ++lv) { Lp[lv] = (int)(lvl[2 * lv] & 0xFFu); Lm[lv] = (int)(lvl[2 * lv] >> 8); Lt[lv] = (int)lvl[2 * lv + 1]; }
  if (Lp[nla] != n || Lm[nla] != n_pairs || Lt[nla] != n_tri) return 0;
  auto rounds = [](int c) { return (c + 63) / 64; };
  const int INF = 1 << 30;
  struct Tr { u64 r; unsigned kk; };

  // ---- (1) pivots ------------------------------------------------------------------------------------------------------
  std::vector<int> lev_of(256, INF), E(256, INF), latest(256, 0), deg(256, 0);
  std::vector<std::vector<u64>> pairs_of(256);
  std::vector<std::vector<Tr>> tris_of(256);
  std::vector<unsigned> piv_of(256, 0);
  std::vector<int> order_k;               // pivots in list order
  for (int lv = 0; lv < nla; ++lv) for (int q = Lp[lv]; q < Lp[lv + 1]; ++q) { const int k = (int)((piv[q] >> 16) & 0xFFu); lev_of[k] = lv; piv_of[k] = piv[q]; order_k.push_back(k); }
  for (int m = 0; m < n_pairs; ++m) pairs_of[(size_t)(pair[m] >> 56)].push_back(pair[m]);
  for (int t = 0; t < n_tri; ++t) tris_of[(size_t)(tri[t] >> 56)].push_back(Tr{tri[t], (unsigned)trik[t] & ~(unsigned)(0x4000u | 0x8000u)});
  for (int k : order_k) {
    int first_nb = INF;
    for (u64 r : pairs_of[k]) first_nb = std::min(first_nb, lev_of[(size_t)((r >> 48) & 0xFFu)]);
    deg[k] = (int)pairs_of[k].size();
    E[k] = lev_of[k];
    latest[k] = lev_of[k] >= ltail ? lev_of[k] : std::min(first_nb - 1, ltail - 1);
    if (latest[k] < lev_of[k]) latest[k] = lev_of[k];
  }
  long pr_before = 0, pr_after = 0;
  for (int lv = 0; lv < ltail; ++lv) pr_before += rounds(Lm[lv + 1] - Lm[lv]);
  {
    std::vector<std::vector<int>> pool((size_t)ltail);
    for (int k : order_k) if (lev_of[k] < ltail) pool[(size_t)lev_of[k]].push_back(k);
    for (int lv = 0; lv + 1 < ltail; ++lv) {
      int np = 0;
      for (int k : pool[lv]) np += deg[k];
      const int excess = np > 64 ? np % 64 : 0;       // pairs beyond the last full round (a level of one round gains nothing)
      if (excess == 0) continue;
      std::vector<int> cand;
      for (int k : pool[lv]) if (latest[k] > lv && deg[k] > 0) cand.push_back(k);
      std::stable_sort(cand.begin(), cand.end(), [&](int a_, int b_) { return latest[a_] > latest[b_]; });
      int moved = 0;
      std::vector<int> go;
      for (int k : cand) { if (moved >= excess) break; go.push_back(k); moved += deg[k]; }
      if (moved < excess) continue;
      for (int k : go) { E[k] = lv + 1; pool[lv].erase(std::find(pool[lv].begin(), pool[lv].end(), k)); pool[lv + 1].push_back(k); }
    }
    bool ok = true;
    for (int lv = 0; lv < ltail; ++lv) {
      int np = 0;
      for (int k : pool[lv]) np += deg[k];
      pr_after += rounds(np);
      if (pool[lv].empty()) ok = false;                                                             // (the level table keeps its shape)
    }
    if (!ok || pr_after >= pr_before) { for (int k : order_k) E[k] = lev_of[k]; pr_after = pr_before; }
  }
  // pivot and pair lists in the (new) level order: a level's own pivots in their old order, then the ones that moved in
  std::vector<unsigned> piv2; std::vector<u64> pair2;
  std::vector<int> Lp2((size_t)nla + 1), Lm2((size_t)nla + 1), Lt2((size_t)nla + 1);
  std::vector<std::vector<Tr>> offd((size_t)nla), diag((size_t)nla);
  for (int lv = 0; lv < nla; ++lv) {
    Lp2[lv] = (int)piv2.size(); Lm2[lv] = (int)pair2.size();
    for (int pass = 0; pass < 2; ++pass)
      for (int k : order_k) {
        if (E[k] != lv || (pass == 0) != (lev_of[k] == lv)) continue;
        piv2.push_back(piv_of[k]);
        pair2.insert(pair2.end(), pairs_of[k].begin(), pairs_of[k].end());
        for (const Tr& t : tris_of[k]) ((t.kk & 0x2000u) ? diag[lv] : offd[lv]).push_back(t);
      }
  }
  Lp2[nla] = (int)piv2.size(); Lm2[nla] = (int)pair2.size();
  if (Lp2[nla] != n || Lm2[nla] != n_pairs) return -1;

  // ---- (2) Schur updates -----------------------------------------------------------------------------------------------
  std::vector<int> first((size_t)nnzF + 1, INF);      // first level at which a matrix entry is read as an operand
  auto rd = [&](unsigned en, int lv) { if (en <= (unsigned)nnzF && lv < first[en]) first[en] = lv; };
  for (int lv = 0; lv < nla; ++lv) {
    for (int q = Lp2[lv]; q < Lp2[lv + 1]; ++q) rd(piv2[(size_t)q] & 0xFFFFu, lv);
    for (const std::vector<Tr>* list : {&offd[lv], &diag[lv]})
      for (const Tr& t : *list) { rd((unsigned)((t.r >> 16) & 0xFFFFu), lv); rd((unsigned)((t.r >> 32) & 0xFFFFu), lv); }
  }
  struct Rec { Tr t; int last; bool cooked; };      // last level that may hold the record; cooked: behind its pivot's level
  std::vector<std::vector<Rec>> stay((size_t)nla);
  std::vector<Rec> carry;
  long tr_before = 0, tr_after = 0;
  for (int lv = 0; lv < nla; ++lv) {
    const int own = (int)offd[lv].size() + (int)diag[lv].size();
    if (lv >= ltail) { for (const Tr& t : offd[lv]) stay[lv].push_back(Rec{t, lv, false}); continue; }
    tr_before += rounds(Lt[lv + 1] - Lt[lv]);
    std::vector<Rec> cand = carry;
    carry.clear();
    for (const Tr& t : offd[lv]) {
      if ((t.r & 0xFFFFu) > (u64)nnzF) return 0;      // (a record that points outside the pattern: keep the schedule as built)
      const int fr = first[(size_t)(t.r & 0xFFFFu)];
      int last = (fr < ltail ? fr : ltail) - 1;
      if (last < lv) last = lv;         // (cannot happen: the target of a level's update is read by a later level)
      cand.push_back(Rec{t, last, false});
    }
    const int T = (int)cand.size() + (int)diag[lv].size();
    const int x = (lv + 1 < ltail) ? T % 64 : 0;       // records beyond the last full round
    bool moved = false;
    if (x > 0 && T > 64) {
      std::vector<int> idx;
      for (int c = 0; c < (int)cand.size(); ++c) if (cand[c].last > lv) idx.push_back(c);
      if ((int)idx.size() >= x) {                      // the x most flexible records move on -- if there are that many that may
        std::stable_sort(idx.begin(), idx.end(), [&](int a_, int b_) { return cand[a_].last > cand[b_].last; });
        std::vector<char> go(cand.size(), 0);
        for (int c = 0; c < x; ++c) go[idx[c]] = 1;
        for (int c = 0; c < (int)cand.size(); ++c) { if (go[c]) { Rec r = cand[c]; r.cooked = true; carry.push_back(r); } else stay[lv].push_back(cand[c]); }
        moved = true;
      }
    }
    if (!moved) stay[lv] = cand;      // records that were carried here and cannot wait any longer stay; so does everything else
    tr_after += rounds((int)stay[lv].size() + (int)diag[lv].size());
    (void)own;
  }
  if (!carry.empty()) return 0;       // (cannot happen: `last` never exceeds ltail - 1, where nothing moves on)
  const bool pivots_moved = pr_after < pr_before;
  if (!pivots_moved && tr_after >= tr_before) return 0;      // nothing gained: the schedule stays as built
  // the triple lists: per level [off-diagonal (raw) ...][diagonal ...][moved in (cooked) ...], flags from the final positions
  std::vector<u64> tri2((size_t)n_tri);
  std::vector<u16> trik2((size_t)n_tri);
  int pos = 0;
  for (int lv = 0; lv < nla; ++lv) {
    Lt2[lv] = pos;
    for (const Rec& r : stay[lv]) if (!r.cooked) { tri2[(size_t)pos] = r.t.r; trik2[(size_t)pos] = (u16)r.t.kk; ++pos; }
    const int d0 = pos;
    int first_of[256];
    for (int k = 0; k < 256; ++k) first_of[k] = -1;
    for (const Tr& t : diag[lv]) {
      const int k = (int)(t.r >> 56);
      if (first_of[k] < 0) first_of[k] = pos;
      unsigned kk = t.kk;
      if (((pos - Lt2[lv]) >> 6) > ((first_of[k] - Lt2[lv]) >> 6)) kk |= 0x4000u;      // TK_DCOOK
      tri2[(size_t)pos] = t.r; trik2[(size_t)pos] = (u16)kk; ++pos;
    }
    (void)d0;
    for (const Rec& r : stay[lv]) if (r.cooked) { tri2[(size_t)pos] = r.t.r; trik2[(size_t)pos] = (u16)(r.t.kk | 0x4000u | 0x8000u); ++pos; }
  }
  Lt2[nla] = pos;
  if (pos != n_tri) return -1;
  // diagonal triples of ONE pivot must be contiguous in a level's list for the TK_DCOOK rule (they are: tris_of keeps the build order)
  if (getenv("PPN_VERBOSE"))
    fprintf(stderr, "[ppn] shared schedule: rounds of the backward pass %ld -> %ld, of the level passes %ld -> %ld\n", pr_before, pr_after, tr_before, tr_after);
  for (int lv = 0; lv <= nla; ++lv) { lvl[2 * lv] = (unsigned)Lp2[lv] | ((unsigned)Lm2[lv] << 8); lvl[2 * lv + 1] = (unsigned)Lt2[lv]; }
  memcpy(trik, trik2.data(), sizeof(u16) * (size_t)n_tri);
  if (dev_h2d(e->base_piv, piv2.data(), sizeof(unsigned) * (size_t)n, e->stream)) return -1;
  if (dev_h2d(e->base_pair, pair2.data(), sizeof(u64) * (size_t)n_pairs, e->stream)) return -1;
  if (dev_h2d(e->base_tri, tri2.data(), sizeof(u64) * (size_t)n_tri, e->stream)) return -1;
  if (dev_h2d(e->base_cache + d.co_lvl, lvl, sizeof(unsigned) * 2 * (size_t)(nla + 1), e->stream)) return -1;
  if (dev_h2d(e->base_cache + d.co_trik, trik, sizeof(u16) * (size_t)n_tri, e->stream)) return -1;
  return 0;
}

static int copy_state(ppn_engine* e, DevState* dst, const DevState* src) {
  const DevCase& d = e->dc;
  const size_t B = e->batch;
  int rc = 0;
#define CP(m, type, n) rc |= dev_d2d(dst->m, src->m, sizeof(type) * B * (n), e->stream);
  CP(vm, double, d.nrows) CP(va, double, d.nrows) CP(pg, double, d.nP) CP(qg, double, d.nP) CP(vg, double, d.nP)
  CP(pd, double, d.nL) CP(qd, double, d.nL) CP(pf, double, d.nl) CP(qf, double, d.nl) CP(pt, double, d.nl)
  CP(qt, double, d.nl) CP(amps, double, d.nl) CP(pn, u8, d.nP) CP(ln, u8, d.nL) CP(on, u8, d.nl) CP(en, u8, d.nl)
  CP(st, u8, d.nl) CP(rec, int, d.nl) CP(lcd, int, d.nl) CP(ncd, int, d.nS) CP(soft, int, d.nl)
  CP(done, u8, 1) CP(dead, u8, 1) CP(succ, u8, 1) CP(btype, u8, d.nrows) CP(flag, int, 1) CP(ill, int, 1)
  CP(depth, int, 1) CP(nsolve, int, 1) CP(niter, int, 1) CP(slot, int, 1) CP(row, int, 1) CP(nlc, int, 1)
  CP(npc, int, 1) CP(epoch, int, 1) CP(prow, int, 1) CP(lev, u8, d.nl) CP(src, int, 1) CP(draws, unsigned, 1)
#undef CP
  return rc;
}

static void default_reward(double* rw, double c) {   // parameters/default14/reward_signal.py:8-43 with `constant` = c
  rw[0] = -1.0; rw[1] = -0.02; rw[2] = -c / 5.0; rw[3] = -c / 10.0; rw[4] = -c;
  rw[5] = rw[6] = rw[7] = -c / 100.0; rw[8] = -c; rw[9] = -c; rw[10] = -5.0 * c; rw[11] = -0.2; rw[12] = -0.1;
}

extern "C" int ppn_set_reward(ppn_engine* e, const ppn_reward_params* p) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !p) return PPN_E_INVALID;
  const double* v = (const double*)p;
  for (int k = 0; k < 13; ++k) e->dc.R.rw[k] = v[k];
  return PPN_OK;
}

// The work-queue rollout kernel (K_POLICY_ROLLOUT): items (step, environment) handed to as many workgroups as the GPU holds, XCD-affine
// where there are several L2s.  ppn_rollout_policy plays a built-in policy through it, ppn_rollout(auto_reset = 1) its action matrices.
static int queue_rollout_launch(ppn_engine* e, KArgs& a, int n_steps) {
  a.restart_prio = e->restart_prio;
  a.n_envs = e->batch; a.n_work = e->batch * n_steps;
  a.work_counter = e->d_work; a.progress = e->d_progress;
#ifndef PPN_EMU
  if (e->n_xcd == 0) {      // once per engine: how many L2s do workgroups land on?  (PPN_XCD_AFFINE=0: the agent-scope hand-over everywhere)
    e->n_xcd = 1;
    const char* v = getenv("PPN_XCD_AFFINE");
    if (!(v && v[0] == '0')) {
      int h = 0;
      if (dev_zero(e->d_xwork, sizeof(int), e->stream)) return fail(e, PPN_E_HIP, "XCD probe failed: %s", dev_err());
      hipLaunchKernelGGL(ppn_xcc_probe_kernel, dim3(4096), dim3(64), 0, e->stream, e->d_xwork);
      if (hipStreamSynchronize(e->stream) != hipSuccess || dev_d2h(&h, e->d_xwork, sizeof(int), e->stream)) return fail(e, PPN_E_HIP, "XCD probe failed: %s", dev_err());
      e->n_xcd = (h >= 0 && h < 16) ? h + 1 : 1;
    }
  }
  if (e->n_xcd > 1 && e->batch >= 2 * e->n_xcd) {      // XCD-affine hand-out: see K_POLICY_ROLLOUT
    a.n_xcd = e->n_xcd; a.work_counter = e->d_xwork;
    if (dev_zero(e->d_xwork, sizeof(int) * 16 * 32, e->stream)) return fail(e, PPN_E_HIP, "rollout: clearing the work counters failed: %s", dev_err());
  }
#endif
  int nblocks = e->batch;
  if (dev_zero(e->d_progress, sizeof(int) * (size_t)e->batch, e->stream) || dev_zero(e->d_work, sizeof(int), e->stream))
    return fail(e, PPN_E_HIP, "rollout: clearing the progress counters failed: %s", dev_err());
#ifndef PPN_EMU
  {
    // environments are handed out heaviest first within every step, as the stepped form does (the key of the LAST step before
    // the rollout: it only shapes the start of the launch)
    if (e->order_launches && e->batch > 1024) {
      hipLaunchKernelGGL(ppn_order_kernel, dim3(1), dim3(1024), 0, e->stream, e->st.prio, e->d_perm, e->batch, (int*)nullptr, 0, (const int*)nullptr);
      a.perm = e->d_perm;
    }
    const int slots_ = resident_slots_of(e);
    nblocks = std::max(1, std::min(slots_ > 0 ? slots_ : e->batch, e->batch));
    { const char* v = getenv("PPN_ROLLOUT_WORKGROUPS"); if (v && atoi(v) > 0) nblocks = std::min(nblocks, atoi(v)); }      // (experiments: the rollout kernel at the step server's occupancy)
  }
#endif
  if (launch<K_POLICY_ROLLOUT>(e, a, nblocks, true)) return fail(e, PPN_E_HIP, "policy rollout launch failed: %s", dev_err());
  return PPN_OK;
}

static int obs_length(const DevCase& d, int layout);
struct ObsSpec { void* dst; int sections, stride, f32; };      // ppn_step_observe: where the step kernel writes the observation rows
static int step_launch(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t simulate, int32_t auto_reset,
                       int n_steps, int per_step_actions, const ObsSpec* ob = nullptr) {
  if (e->chronics_dirty) { int rc = sync_chronics(e); if (rc) return rc; }
  const size_t mat = (size_t)e->batch * e->dc.alen;
  const u8* dact = actions;
  if (!actions_on_device) {
    const size_t need = per_step_actions ? mat * (size_t)n_steps : mat;
    if (need > mat) {       // a whole action sequence from the host: staged in a buffer of its own
      if (need > e->d_rollout_cap) {
        void* p = nullptr;
        if (dev_malloc(&p, need)) return fail(e, PPN_E_HIP, "rollout action buffer allocation failed: %s", dev_err());
        e->allocs.push_back(p);      // (the outgrown buffer is released with the engine)
        e->d_rollout = (u8*)p; e->d_rollout_cap = need;
      }
      if (dev_h2d(e->d_rollout, actions, need, e->stream)) return fail(e, PPN_E_HIP, "action upload failed");
      dact = e->d_rollout;
    } else {
      if (dev_h2d(e->d_actions, actions, mat, e->stream)) return fail(e, PPN_E_HIP, "action upload failed");
      dact = e->d_actions;
    }
  }
  const int mode_asked = simulate ? 0 : (auto_reset == 2 ? 2 : (auto_reset ? 1 : 0));
  if (mode_asked != 2) { int rcs = settle_restarts(e); if (rcs) return rcs; }
  // (Restart memo under the FUSED restart -- ppn_step_observe, the rollouts, the step server: those kernels serve a restart from its
  //  snapshot when there is one and save the ones they had to compute themselves, body_episode<MEMO, APPLY>; nothing to do here.  The
  //  plain step kernel is compiled without either: ppn_step(auto_reset = 1) computes every restart.)
  const int mode = mode_asked;
  if (mode_asked == 2 && e->d_memo && e->pending_restart) {
    // restart memo, in front of the step kernel: (1) the owed restarts that have a snapshot are served by a light kernel; (2) a
    // LEARNING pass of the game-over kernel computes and saves the eligible ones that have none yet (body_game_over, only_owed = 2) --
    // run in the first steps after the memo was set up and every n-th step afterwards (positions that come up rarely): it is carved
    // for the solver and costs a launch of `batch` such workgroups even when it finds nothing to do.  Whatever neither takes is
    // restarted by the step kernel itself, as without the memo.
    KArgs am = make_args(e, false);
#ifdef PPN_EMU
    for (int env = 0; env < e->batch; ++env) (void)body_memo_apply(am.d, am.st, am.memo, env, 0);
#else
    hipLaunchKernelGGL(ppn_memo_apply_kernel, dim3(e->batch), dim3(64), 0, e->stream, am);
    if (hipGetLastError() != hipSuccess) return fail(e, PPN_E_HIP, "restart-memo launch failed: %s", dev_err());
#endif
    const bool learn = e->memo_learn_left > 0 || (e->memo_learn_every > 0 && (++e->memo_tick % e->memo_learn_every) == 0);
    if (e->memo_learn_left > 0) --e->memo_learn_left;
    if (learn) {
      am.sim = 2;
      if (launch<K_GAMEOVER>(e, am, e->batch)) return fail(e, PPN_E_HIP, "restart-memo pass launch failed: %s", dev_err());
    }
  }
  if (simulate) { if (copy_state(e, &e->sim, &e->st)) return fail(e, PPN_E_HIP, "state fork failed: %s", dev_err()); }
  KArgs a = make_args(e, simulate != 0);
  a.actions = dact; a.sim = simulate ? 1 : 0; a.auto_reset = mode;
  a.n_steps = n_steps; a.action_step_stride = per_step_actions ? mat : 0;
  a.restart_prio = e->restart_prio;
  if (ob) { a.obs = ob->dst; a.obs_sections = ob->sections; a.obs_stride = ob->stride; a.obs_f32 = ob->f32; }
  int nblocks = e->batch;
  // schedule pre-pass: only where node switches can change the schedule at all (busbars beyond one per substation: the four-word
  // kernels), for the step the launch below executes first
  const bool two_cap = e->two_cap && !simulate && n_steps == 1 && e->lds_sched <= 64 * 1024;
  if (two_cap) a.ecap_small = e->ecap_small;
  // A step that runs the schedule pre-pass (two-capacity stepping always does) is timed as ONE group -- pre-pass, launch order and
  // the step launch(es) between one pair of events: the pre-pass is per-step GPU work of such a step
  bool grouped = false, group_timed = false;
  if (e->sched_prepass && e->W == 4 && e->dc.NB > e->dc.nS && !simulate && e->lds_sched <= 64 * 1024) {
    bool run = e->sched_prepass == 2 || two_cap;      // (two-capacity stepping needs every environment's class before every step)
#ifndef PPN_EMU
    if (!run) { const int slots_ = resident_slots_of(e); run = slots_ > 0 && (long)e->sched_rounds * slots_ <= (long)e->batch; }
#else
    run = true;      // (the emulation build always runs it: the tests exercise the code)
#endif
#if !defined(PPN_ONLY_W1) && !defined(PPN_ONLY_W2)
    if (run) {
      grouped = true; group_timed = time_group_begin(e);
      if (launch_sched<4>(e, a, e->batch)) return fail(e, PPN_E_HIP, "schedule pre-pass launch failed: %s", dev_err());
    }
#endif
  }
  if (two_cap && !grouped) { grouped = true; group_timed = time_group_begin(e); }
  a.ecap_small = 0;
  const size_t lds_step = two_cap ? e->lds_small_cap : e->lds_bytes;
#ifndef PPN_EMU
  if (e->order_launches && !simulate && e->batch > 1024) {   // more workgroups than resident slots: hand out the long ones first
    const int slots_ = e->persistent ? resident_slots_of(e, lds_step) : 0;
    // (the throughput regime only: see K_STEP_PERSIST; not for the one-word kernels -- an IEEE-14 step is ~40 us, the trip to the
    //  position counter between two of them costs more than the workgroup launch it replaces: 37.9 vs 36.8 M at 16384)
    const bool pers = !ob && e->persistent && e->W >= 2 && n_steps == 1 && slots_ > 0 && (long)e->persistent_rounds * slots_ <= (long)e->batch;
    hipLaunchKernelGGL(ppn_order_kernel, dim3(1), dim3(1024), 0, e->stream, e->st.prio, e->d_perm, e->batch, pers ? e->d_work : (int*)nullptr, slots_, (const int*)nullptr);
    a.perm = e->d_perm;
    if (pers) { a.work_counter = e->d_work; a.n_work = e->batch; nblocks = e->resident_slots; }
  }
#endif
  const bool timed1 = !grouped;      // (a lone step launch carries its own event pair)
  int rc_step = 0;
  if (two_cap) {
    // the small-storage launch (class 0), then the large-storage one for whatever is left (class 1: plain form, one workgroup per
    // environment -- those of the other class return at once)
    KArgs as = a;
    as.d.ECAP = e->ecap_small; as.d.QCAP = e->ecap_small; as.d.LUCAP = 4 * e->ecap_small;
    as.cap_class = 0;
    e->lds_override = e->lds_small_cap;
    rc_step = ob ? launch<K_STEP_OBS>(e, as, nblocks, false) : (as.work_counter ? launch<K_STEP_PERSIST>(e, as, nblocks, false) : launch<K_STEP>(e, as, nblocks, false));
    e->lds_override = 0;
    if (!rc_step) {
      KArgs al = a;
      al.cap_class = 1; al.perm = nullptr; al.work_counter = nullptr;
      rc_step = ob ? launch<K_STEP_OBS>(e, al, e->batch, false) : launch<K_STEP>(e, al, e->batch, false);
    }
  } else if (ob) rc_step = launch<K_STEP_OBS>(e, a, nblocks, timed1);
  else if (n_steps > 1 && mode == 1 && (long long)n_steps * e->batch <= 0x7fffff00LL) {
    // OPEN-LOOP rollout with the fused restart: through the work-queue kernel (round 6) -- no environment is pinned to a workgroup for
    // all its steps, so the launch does not end with the environment whose steps add up to the longest chain
    KArgs q = a;
    q.policy.id = PPN_POLICY_REPLAY; q.perm = nullptr; q.work_counter = nullptr;
    q.policy_out = e->d_actions;      // (not written)
    const int rcq = queue_rollout_launch(e, q, n_steps);
    if (rcq) return rcq;
  }
  else rc_step = n_steps > 1 ? launch<K_ROLLOUT>(e, a, e->batch, timed1) : (a.work_counter ? launch<K_STEP_PERSIST>(e, a, nblocks, timed1) : launch<K_STEP>(e, a, nblocks, timed1));
  if (grouped) time_group_end(e, group_timed);
  if (rc_step) return fail(e, PPN_E_HIP, "step kernel launch failed: %s", dev_err());
  e->last_step_form = (ob ? 2 : (n_steps > 1 ? 3 : (a.work_counter ? 1 : 0))) + (two_cap ? 4 : 0);
  if (a.auto_reset && e->maybe_dead) {
    // environments that were already over when the step began (after ppn_reset or after steps without auto_reset) did
    // not step; they are restarted by this post-pass.  Environments that end DURING a step restart inside the step
    // kernel, so from the second auto-reset step on nothing is left to do here.
    if (launch<K_GAMEOVER>(e, a, e->batch)) return fail(e, PPN_E_HIP, "game-over kernel launch failed: %s", dev_err());
    e->maybe_dead = false;
    // (ppn_step_observe: the rows of the environments this post-pass restarted were written before it -- gathered once more; only
    //  the first auto-reset step after a ppn_reset or after steps without auto_reset comes through here)
    if (ob && launch<K_OBS>(e, a, e->batch)) return fail(e, PPN_E_HIP, "observation kernel launch failed: %s", dev_err());
  }
  if (!a.auto_reset && !simulate) e->maybe_dead = true;
  if (mode == 2) e->pending_restart = true;
  return PPN_OK;
}

extern "C" int ppn_step(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t simulate,
                        int32_t auto_reset) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions) return PPN_E_INVALID;
  return step_launch(e, actions, actions_on_device, simulate, auto_reset, 1, 0);
}

extern "C" int ppn_step_observe(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t auto_reset,
                                int32_t layout, int32_t as_f32, void* obs_device, size_t bytes) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions || !obs_device || layout < 0 || layout > 2) return PPN_E_INVALID;
  if (auto_reset != 0 && auto_reset != 1) return fail(e, PPN_E_INVALID, "ppn_step_observe: auto_reset must be 0 or 1 (a deferred restart would leave the observation of the ended episode in the rows)");
  const int len = obs_length(e->dc, layout);
  const size_t need = (size_t)e->batch * (size_t)len * (as_f32 ? sizeof(float) : sizeof(double));
  if (bytes < need) return fail(e, PPN_E_INVALID, "ppn_step_observe: buffer too small (%zu < %zu)", bytes, need);
  ObsSpec ob = { obs_device, layout == 0 ? 3 : layout, len, as_f32 ? 1 : 0 };
  return step_launch(e, actions, actions_on_device, 0, auto_reset, 1, 0, &ob);
}

extern "C" int ppn_rollout(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t n_steps,
                           int32_t per_step_actions, int32_t auto_reset) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions || n_steps <= 0) return PPN_E_INVALID;
  if (e->maybe_dead && auto_reset) {      // environments that are over right now are restarted first: every environment plays all its steps
    int rc = ppn_process_game_over(e, nullptr);
    if (rc) return rc;
    e->maybe_dead = false;
  }
  return step_launch(e, actions, actions_on_device, 0, auto_reset, n_steps, per_step_actions);
}

// ---- device-side policies -------------------------------------------------------------------------------------------------
static int fill_policy(ppn_engine* e, KArgs* a, int32_t policy, const double* params, int32_t n_params) {
  if (policy != PPN_POLICY_DO_NOTHING && policy != PPN_POLICY_LINE_RELIEF) return fail(e, PPN_E_INVALID, "unknown policy %d", policy);
  if (n_params < 0 || n_params > 4 || (n_params > 0 && !params)) return fail(e, PPN_E_INVALID, "a policy takes at most 4 parameters");
  a->policy.id = policy;
  for (int k = 0; k < 4; ++k) a->policy.p[k] = k < n_params ? params[k] : 0.0;
  if (policy == PPN_POLICY_LINE_RELIEF && n_params < 1) a->policy.p[0] = 1.0;      // (default threshold: the thermal limit itself)
  return PPN_OK;
}

extern "C" int ppn_policy_actions(ppn_engine* e, int32_t policy, const double* params, int32_t n_params, uint8_t* actions_out_device) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions_out_device) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }      // the policy looks at the state
  KArgs a = make_args(e, false);
  { int rc = fill_policy(e, &a, policy, params, n_params); if (rc) return rc; }
  a.policy_out = actions_out_device;
  if (launch<K_POLICY>(e, a, e->batch)) return fail(e, PPN_E_HIP, "policy kernel launch failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int ppn_rollout_policy(ppn_engine* e, int32_t policy, const double* params, int32_t n_params, int32_t n_steps) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || n_steps <= 0) return PPN_E_INVALID;
  if ((long long)n_steps * e->batch > 0x7fffff00LL) return fail(e, PPN_E_INVALID, "ppn_rollout_policy: batch x n_steps exceeds the work counter");
  if (e->chronics_dirty) { int rc = sync_chronics(e); if (rc) return rc; }
  if (e->maybe_dead) {      // environments that are over right now are restarted first: every environment plays all its steps
    int rc = ppn_process_game_over(e, nullptr);
    if (rc) return rc;
    e->maybe_dead = false;
  }
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  KArgs a = make_args(e, false);
  { int rc = fill_policy(e, &a, policy, params, n_params); if (rc) return rc; }
  a.policy_out = e->d_actions;
  a.auto_reset = 1;
  return queue_rollout_launch(e, a, n_steps);
}

// ---- topology-action search: K candidate actions evaluated from the current state of chosen environments --------------
// The fork of ppn_simulate_candidates copies ~40 per-environment rows per candidate: ONE launch over a table of (destination, source,
// bytes) instead of one launch per field (round 6: 42 launches of a few microseconds each were 8 % of a search round).
#define PPN_FORK_MAX_FIELDS 48
struct ForkTable { unsigned char* dst[PPN_FORK_MAX_FIELDS]; const unsigned char* src[PPN_FORK_MAX_FIELDS]; unsigned bytes[PPN_FORK_MAX_FIELDS]; int n_fields; };
#ifndef PPN_EMU
__global__ void __launch_bounds__(256) ppn_fork_rows(const ForkTable t, const int* idx) {
  const size_t c = blockIdx.x, e_ = (size_t)idx[blockIdx.x];
  for (int f = 0; f < t.n_fields; ++f) {
    const size_t rb = t.bytes[f];
    const unsigned char* s = t.src[f] + e_ * rb;
    unsigned char* d = t.dst[f] + c * rb;
    if (((rb | (size_t)s | (size_t)d) & 15) == 0) { for (size_t k = threadIdx.x; k < rb / 16; k += 256) ((uint4*)d)[k] = ((const uint4*)s)[k]; }
    else if (((rb | (size_t)s | (size_t)d) & 3) == 0) { for (size_t k = threadIdx.x; k < rb / 4; k += 256) ((unsigned*)d)[k] = ((const unsigned*)s)[k]; }
    else { for (size_t k = threadIdx.x; k < rb; k += 256) d[k] = s[k]; }
  }
}
#endif
static int fork_rows(ppn_engine* e, const ForkTable& t, const int* d_idx, const int* h_idx, int n) {
#ifdef PPN_EMU
  (void)e; (void)d_idx;
  for (int f = 0; f < t.n_fields; ++f)
    for (int c = 0; c < n; ++c) memcpy(t.dst[f] + (size_t)c * t.bytes[f], t.src[f] + (size_t)h_idx[c] * t.bytes[f], t.bytes[f]);
  return 0;
#else
  (void)h_idx;
  hipLaunchKernelGGL(ppn_fork_rows, dim3(n), dim3(256), 0, e->stream, t, d_idx);
  return hipGetLastError() == hipSuccess ? 0 : -1;
#endif
}

extern "C" int ppn_simulate_candidates(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device,
                                       const int32_t* env_ids, int32_t n) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions || !env_ids || n <= 0) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  for (int c = 0; c < n; ++c) if (env_ids[c] < 0 || env_ids[c] >= e->batch) return fail(e, PPN_E_INVALID, "ppn_simulate_candidates: environment %d out of range", env_ids[c]);
  if (e->chronics_dirty) { int rc = sync_chronics(e); if (rc) return rc; }
  const DevCase& d = e->dc;
  if (n > e->cand_cap) {      // (re)allocate the candidate slots; the outgrown ones are released first
    const int cap = std::max(n, std::max(e->batch, 2 * e->cand_cap));
#ifndef PPN_EMU
    (void)hipStreamSynchronize(e->stream);     // nothing in flight may still read the old slots
#endif
    for (void* p : e->cand_allocs) dev_free(p);
    e->cand_allocs.clear();
    e->cand_cap = 0; e->n_cand = 0;
    const size_t mark = e->allocs.size();
    alloc_state(e, &e->cand, (size_t)cap);
    e->d_cand_actions = dalloc<u8>(e, (size_t)cap * d.alen);
    e->d_cand_obs = dalloc<double>(e, (size_t)cap * d.obslen);
    e->d_cand_ids = dalloc<int>(e, (size_t)cap);
    e->d_cand_perm = dalloc<int>(e, (size_t)cap);
    e->cand_allocs.assign(e->allocs.begin() + mark, e->allocs.end());      // (dalloc files everything under e->allocs)
    e->allocs.resize(mark);
    if (e->mem_failed) {
      e->mem_failed = false;
      for (void* p : e->cand_allocs) dev_free(p);
      e->cand_allocs.clear();
      return fail(e, PPN_E_HIP, "candidate slots: device allocation failed: %s", dev_err());
    }
    e->cand_cap = cap;
  }
  e->n_cand = n;
  if (dev_h2d(e->d_cand_ids, env_ids, sizeof(int) * (size_t)n, e->stream)) return fail(e, PPN_E_HIP, "candidate ids upload failed");
  const u8* dact = actions;
  if (!actions_on_device) {
    if (dev_h2d(e->d_cand_actions, actions, (size_t)n * d.alen, e->stream)) return fail(e, PPN_E_HIP, "action upload failed");
    dact = e->d_cand_actions;
  }
  // fork: every per-environment array of the live state.  The schedule caches: engines whose busbars may split run the schedule
  // pre-pass over the candidate slots (body_sched, candidate mode: own cache / the environment's / build -- a slot KEEPS its schedule
  // across calls); the others copy the environment's cache and records into the slot as before (a candidate that does not move an
  // element to another busbar solves on the schedule its environment already has, anything else is built inside the solve)
  const bool cand_prepass = e->sched_prepass && e->W == 4 && e->dc.NB > e->dc.nS && e->lds_sched <= 64 * 1024 && e->cand_cache;
  int rc = 0;
  DevState* dst = &e->cand; const DevState* src = &e->st;
  ForkTable ft; ft.n_fields = 0;
#define GR(m, type, cnt) { const int f_ = ft.n_fields++; ft.dst[f_] = (unsigned char*)dst->m; ft.src[f_] = (const unsigned char*)src->m; ft.bytes[f_] = (unsigned)(sizeof(type) * (size_t)(cnt)); }
  GR(vm, double, d.nrows) GR(va, double, d.nrows) GR(pg, double, d.nP) GR(qg, double, d.nP) GR(vg, double, d.nP)
  GR(pd, double, d.nL) GR(qd, double, d.nL) GR(pf, double, d.nl) GR(qf, double, d.nl) GR(pt, double, d.nl)
  GR(qt, double, d.nl) GR(amps, double, d.nl) GR(pn, u8, d.nP) GR(ln, u8, d.nL) GR(on, u8, d.nl) GR(en, u8, d.nl)
  GR(st, u8, d.nl) GR(rec, int, d.nl) GR(lcd, int, d.nl) GR(ncd, int, d.nS) GR(soft, int, d.nl)
  GR(done, u8, 1) GR(dead, u8, 1) GR(succ, u8, 1) GR(btype, u8, d.nrows) GR(flag, int, 1) GR(ill, int, 1)
  GR(depth, int, 1) GR(nsolve, int, 1) GR(niter, int, 1) GR(slot, int, 1) GR(row, int, 1) GR(nlc, int, 1)
  GR(npc, int, 1) GR(epoch, int, 1) GR(prow, int, 1) GR(lev, u8, d.nl) GR(src, int, 1) GR(draws, unsigned, 1)
  if (!cand_prepass) { GR(ws_tri, u64, d.TCAP) GR(ws_pair, u64, d.MCAP) GR(ws_piv, unsigned, d.NB) GR(ws_cache, u8, d.cache_stride) }
#undef GR
  static_assert(PPN_FORK_MAX_FIELDS >= 44, "the fork table holds every row of the fork");
  rc |= fork_rows(e, ft, e->d_cand_ids, env_ids, n);
  if (rc) return fail(e, PPN_E_HIP, "state fork failed: %s", dev_err());
  KArgs a = make_args(e, false);
  a.st = e->cand;
  a.actions = dact; a.sim = 1; a.auto_reset = 0;
  const bool two_cap = e->two_cap && cand_prepass;
#if !defined(PPN_ONLY_W1) && !defined(PPN_ONLY_W2)
  if (cand_prepass) {
    a.ssrc.cache = e->st.ws_cache; a.ssrc.tri = e->st.ws_tri; a.ssrc.pair = e->st.ws_pair; a.ssrc.piv = e->st.ws_piv; a.ssrc.ids = e->d_cand_ids;
    a.ecap_small = two_cap ? e->ecap_small : 0;
    if (launch_sched<4>(e, a, n)) return fail(e, PPN_E_HIP, "schedule pre-pass launch failed: %s", dev_err());
    a.ecap_small = 0;
    memset(&a.ssrc, 0, sizeof a.ssrc);
  }
#endif
  int nb_first = n;      // workgroups of the first (or only) step launch
#ifndef PPN_EMU
  if (e->order_launches && n > 1024) {
    // more candidates than resident slots: hand out the long ones first, as the stepped launch does -- a candidate inherits the
    // loading its parent's last step left (round 6: 8192 candidates over 1024 slots in index order ended with their cascades) --
    // and, from persistent_rounds candidates per slot on, through the persistent form of the step kernel
    const int slots_ = e->persistent ? resident_slots_of(e, two_cap ? e->lds_small_cap : e->lds_bytes) : 0;
    const bool pers = e->persistent && e->W >= 2 && slots_ > 0 && (long)e->persistent_rounds * slots_ <= (long)n;
    hipLaunchKernelGGL(ppn_order_kernel, dim3(1), dim3(1024), 0, e->stream, e->st.prio, e->d_cand_perm, n, pers ? e->d_work : (int*)nullptr, slots_, (const int*)e->d_cand_ids);
    a.perm = e->d_cand_perm;
    if (pers) { a.work_counter = e->d_work; a.n_work = n; nb_first = slots_; }
  }
#endif
  if (two_cap) {
    // two-capacity stepping for the candidates as well (round 5): the pre-pass has classed every candidate's schedule; the
    // small-storage launch plays four candidates per CU, the large-storage one whatever does not fit
    KArgs as = a;
    as.d.ECAP = e->ecap_small; as.d.QCAP = e->ecap_small; as.d.LUCAP = 4 * e->ecap_small;
    as.cap_class = 0;
    e->lds_override = e->lds_small_cap;
    const int rc_s = as.work_counter ? launch<K_STEP_PERSIST>(e, as, nb_first) : launch<K_STEP>(e, as, n);
    e->lds_override = 0;
    if (rc_s) return fail(e, PPN_E_HIP, "step kernel launch failed: %s", dev_err());
    a.cap_class = 1; a.work_counter = nullptr; nb_first = n;      // (the large-storage launch: plain form, the other class's workgroups return at once)
  }
  if (a.work_counter ? launch<K_STEP_PERSIST>(e, a, nb_first) : launch<K_STEP>(e, a, n)) return fail(e, PPN_E_HIP, "step kernel launch failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int ppn_process_game_over(ppn_engine* e, const uint8_t* env_mask) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  if (e->chronics_dirty) { int rc = sync_chronics(e); if (rc) return rc; }
  KArgs a = make_args(e, false);
  if (env_mask) {
    if (dev_h2d(e->d_valid, env_mask, e->batch, e->stream)) return fail(e, PPN_E_HIP, "mask upload failed");
    a.valid = e->d_valid;
  }
  if (launch<K_GAMEOVER>(e, a, e->batch)) return fail(e, PPN_E_HIP, "game-over kernel launch failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int ppn_is_action_valid(ppn_engine* e, const uint8_t* actions, uint8_t* valid) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !actions || !valid) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  if (dev_h2d(e->d_actions, actions, (size_t)e->batch * e->dc.alen, e->stream)) return fail(e, PPN_E_HIP, "action upload failed");
  KArgs a = make_args(e, false);
  a.actions = e->d_actions; a.valid = e->d_valid;
  if (launch<K_VALID>(e, a, e->batch)) return fail(e, PPN_E_HIP, "kernel launch failed: %s", dev_err());
  if (dev_d2h(valid, e->d_valid, e->batch, e->stream)) return fail(e, PPN_E_HIP, "download failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int ppn_runpf_batch(ppn_engine* e) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
  KArgs a = make_args(e, false);
  if (launch<K_RUNPF>(e, a, e->batch, true)) return fail(e, PPN_E_HIP, "runpf kernel launch failed: %s", dev_err());
  return PPN_OK;
}

#include "ppn_mpc.inc"
#include "ppn_async.inc"

extern "C" int ppn_sync(ppn_engine* e) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
  { int rcs = settle_restarts(e); if (rcs) return rcs; }
#ifndef PPN_EMU
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(e, PPN_E_HIP, "stream sync failed: %s", dev_err());
#endif
  return PPN_OK;
}

extern "C" int ppn_wait(ppn_engine* e) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
#ifndef PPN_EMU
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(e, PPN_E_HIP, "stream sync failed: %s", dev_err());
#endif
  return PPN_OK;
}

extern "C" void* ppn_stream(ppn_engine* e) {
#ifdef PPN_EMU
  (void)e; return nullptr;
#else
  return e ? (void*)e->stream : nullptr;
#endif
}

extern "C" int ppn_kernel_time(ppn_engine* e, int32_t reset, double* total_ms, int64_t* launches) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e) return PPN_E_INVALID;
#ifndef PPN_EMU
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(e, PPN_E_HIP, "stream sync failed: %s", dev_err());
  for (size_t k = 0; k + 1 < e->ev_used; k += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[k], e->ev[k + 1]) == hipSuccess) e->time_ms += ms;
  }
  e->ev_used = 0;
#endif
  if (total_ms) *total_ms = e->time_ms;
  if (launches) *launches = e->launches;
  if (reset) { e->time_ms = 0.0; e->launches = 0; }
  return PPN_OK;
}

static int obs_length(const DevCase& d, int layout) {     // environment.py:406-531
  const int mini = 4 * d.nL + 4 * d.nP + 6 * d.nl + d.nS + d.nl + 6;
  const int ac = mini + 3 * d.nL + 3 * d.nP + 6 * d.nl;
  return layout == 1 ? mini : (layout == 2 ? ac : d.obslen);
}

extern "C" int ppn_read_observation(ppn_engine* e, int32_t layout, int32_t as_f32, void* dst, size_t bytes, int32_t to_host,
                                    int32_t from_simulation) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !dst || layout < 0 || layout > 2) return PPN_E_INVALID;
  if (from_simulation == 0) { int rcs = settle_restarts(e); if (rcs) return rcs; }
  if (from_simulation == 2 && e->n_cand <= 0) return fail(e, PPN_E_INVALID, "ppn_read_observation: no candidates have been simulated");
  const size_t B = (from_simulation == 2) ? (size_t)e->n_cand : (size_t)e->batch;
  const int len = obs_length(e->dc, layout);
  const size_t need = B * (size_t)len * (as_f32 ? sizeof(float) : sizeof(double));
  if (bytes < need) return fail(e, PPN_E_INVALID, "ppn_read_observation: buffer too small (%zu < %zu)", bytes, need);
  if (e->chronics_dirty) { int rc = sync_chronics(e); if (rc) return rc; }
  KArgs a = make_args(e, from_simulation != 0);
  a.st = (from_simulation == 2) ? e->cand : (from_simulation ? e->sim : e->st);
  void* stage = (from_simulation == 2) ? (void*)e->d_cand_obs : (void*)e->d_obs;     // sized for the full f64 layout
  a.obs = to_host ? stage : dst;
  a.obs_sections = layout == 0 ? 3 : layout; a.obs_stride = len; a.obs_f32 = as_f32 ? 1 : 0;
  if (launch<K_OBS>(e, a, (int)B)) return fail(e, PPN_E_HIP, "observation kernel launch failed: %s", dev_err());
  if (to_host && dev_d2h(dst, stage, need, e->stream)) return fail(e, PPN_E_HIP, "download failed: %s", dev_err());
  return PPN_OK;
}

extern "C" int32_t ppn_observation_length(const ppn_engine* e, int32_t layout) {
  return (!e || layout < 0 || layout > 2) ? -1 : obs_length(e->dc, layout);
}

extern "C" int ppn_read(ppn_engine* e, ppn_field f, void* dst, size_t bytes, int32_t to_host, int32_t from_simulation) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !dst) return PPN_E_INVALID;
  if (from_simulation == 2 && e->n_cand <= 0) return fail(e, PPN_E_INVALID, "ppn_read: no candidates have been simulated");
  const DevState& s = (from_simulation == 2) ? e->cand : (from_simulation ? e->sim : e->st);
  const size_t B = (from_simulation == 2) ? (size_t)e->n_cand : (size_t)e->batch;
  if (f == PPN_F_OBSERVATION) return ppn_read_observation(e, 0, 0, dst, bytes, to_host, from_simulation);
  {   // the report of the last step is what it is; everything else shows the restarted episode
    const bool report = f == PPN_F_DONE || f == PPN_F_FLAG || f == PPN_F_ILLEGAL || f == PPN_F_REWARD || f == PPN_F_ILLEGAL_COUNTS ||
                        f == PPN_F_ACTION_SWITCHES || f == PPN_F_CASCADE_DEPTH || f == PPN_F_LINE_EVENTS || f == PPN_F_SOLVE_OUTCOME ||
                        f == PPN_F_STEP_REPORT || f == PPN_F_N_STEPS || f == PPN_F_RETURN /* counters of EXECUTED steps: a restart does not touch them */ ||
                        (int)f == 100 /* phase counters of the profiling build */ || (int)f == 102 || (int)f == 103;
    if (!report && from_simulation == 0) { int rcs = settle_restarts(e); if (rcs) return rcs; }
  }
  FieldInfo fi; bool w;
  if (!field_info(e, f, &fi, &w)) return fail(e, PPN_E_INVALID, "ppn_read: unknown field %d", (int)f);
  const size_t need = fi.elem * fi.n * B;
  if (bytes < need) return fail(e, PPN_E_INVALID, "ppn_read: buffer too small (%zu < %zu)", bytes, need);
  void* src = *(void**)((char*)&s + fi.off);
  const int rc = to_host ? dev_d2h(dst, src, need, e->stream) : dev_d2d(dst, src, need, e->stream);
  return rc ? fail(e, PPN_E_HIP, "ppn_read copy failed: %s", dev_err()) : PPN_OK;
}

extern "C" int ppn_write(ppn_engine* e, ppn_field f, const void* src, size_t bytes) {
  enter(e);
  PPN_QUIESCE(e);
  if (!e || !src) return PPN_E_INVALID;
  if ((int)f != 100) { int rcs = settle_restarts(e); if (rcs) return rcs; }
  FieldInfo fi; bool w;
  if (!field_info(e, f, &fi, &w) || !w) return fail(e, PPN_E_INVALID, "ppn_write: field %d is not writable", (int)f);
  const size_t need = fi.elem * fi.n * (size_t)e->batch;
  if (bytes != need) return fail(e, PPN_E_INVALID, "ppn_write: expected %zu bytes, got %zu", need, bytes);
  void* dst = *(void**)((char*)&e->st + fi.off);
  return dev_h2d(dst, src, need, e->stream) ? fail(e, PPN_E_HIP, "ppn_write copy failed: %s", dev_err()) : PPN_OK;
}
